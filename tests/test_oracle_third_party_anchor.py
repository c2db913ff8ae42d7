"""The VQT / RMS oracle restates librosa 0.8.1 and resampy 0.4.2, which cannot be installed here (PARITY UNPINNED).
What CAN be checked against real third-party code in this image: the functions those libraries themselves delegate to.
  resampy.filters.sinc_window  ->  scipy.signal.kaiser (window), numpy.sinc, numpy.linspace
  librosa.filters.get_window   ->  scipy.signal.get_window(name, n, fftbins=True)
  librosa.stft / cqt response  ->  numpy / scipy FFT, numpy.pad(mode="reflect")
These are not a pin of the composed algorithm; they remove the leaves from the list of things that could be wrong."""
import numpy as np
import pytest

from oracle import stem_check_ref, vqt_ref


def test_kaiser_fast_window_from_scipys_kaiser():
    """resampy generated kaiser_fast with sinc_window(num_zeros=16, precision=9, window=partial(scipy.signal.kaiser,
    beta=8.555504641634386), rolloff=0.85): the same expression on scipy's window function."""
    scipy_signal = pytest.importorskip("scipy.signal")
    n = (2 ** 9) * 16
    sinc_win = 0.85 * np.sinc(0.85 * np.linspace(0, 16, num=n + 1, endpoint=True))
    taper = scipy_signal.windows.kaiser(2 * n + 1, 8.555504641634386)[n:]
    want = taper * sinc_win
    got = vqt_ref.kaiser_fast_half_window()
    assert got.shape == want.shape == (n + 1,)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-15)      # numpy's and scipy's Bessel I0 differ in the last bit
    left, right = vqt_ref.decimate2_taps()
    assert left.shape == (32,) and right.shape == (31,)
    np.testing.assert_allclose(left, 0.5 * want[np.arange(32) * 256], rtol=0, atol=1e-15)
    np.testing.assert_allclose(right, 0.5 * want[256 + np.arange(31) * 256], rtol=0, atol=1e-15)
    assert abs(left[0] - 0.425) < 1e-15                       # rolloff / 2 at the centre tap


def test_periodic_hann_is_scipys_fftbins_window():
    scipy_signal = pytest.importorskip("scipy.signal")
    for n in (2, 7, 16, 127, 128, 1000, 1001):
        np.testing.assert_allclose(vqt_ref.hann_periodic(n), scipy_signal.get_window("hann", n, fftbins=True), rtol=0, atol=1e-15)


def test_stft_with_a_ones_window_against_scipy_fft():
    """librosa.stft(window="ones", center=True, pad_mode="reflect"): frames of the reflect-padded signal through an rFFT."""
    scipy_fft = pytest.importorskip("scipy.fft")
    rng = np.random.default_rng(0)
    y = rng.standard_normal(1000).astype(np.float32)
    for n_fft, hop in ((128, 256), (128, 64), (16, 2)):
        got = vqt_ref.stft_ones(y, n_fft, hop)
        ypad = np.pad(y, n_fft // 2, mode="reflect")
        n_frames = 1 + (len(ypad) - n_fft) // hop
        want = np.stack([scipy_fft.rfft(ypad[t * hop:t * hop + n_fft].astype(np.float64)) for t in range(n_frames)], axis=1)
        assert got.shape == want.shape == (n_fft // 2 + 1, n_frames)
        np.testing.assert_allclose(got, want.astype(np.complex64), rtol=0, atol=1e-5)


def test_feature_rms_against_a_direct_numpy_framing():
    """librosa.feature.rms(y, frame_length=2048, hop_length=512, center=True, pad_mode="reflect") =
    sqrt(mean(|frame|^2)) over frames of numpy.pad(y, 1024, "reflect")."""
    rng = np.random.default_rng(1)
    y = (0.3 * rng.standard_normal(20000)).astype(np.float32)
    got = stem_check_ref.feature_rms(y)
    ypad = np.pad(y, 1024, mode="reflect")
    n_frames = 1 + (len(ypad) - 2048) // 512
    want = np.array([np.sqrt(np.mean(np.abs(ypad[t * 512:t * 512 + 2048].astype(np.float64)) ** 2)) for t in range(n_frames)])
    assert got.shape == (1, n_frames)
    np.testing.assert_allclose(got[0], want, rtol=2e-6)


@pytest.mark.parametrize("n", [4097, 1000, 999, 131, 70, 64, 5, 1])
def test_decimation_against_scipys_polyphase_resamplers(n):
    """The restated 2:1 decimation (resampy.resample(x, 2, 1, 'kaiser_fast') as librosa.resample calls it) against two
    independent implementations of "filter with these 63 taps, keep every second sample": scipy.signal.upfirdn and
    scipy.signal.resample_poly.  This pins what a restatement gets wrong first -- the PHASE (output t is centred on input
    2t), the EDGES (zeros outside the signal, no reflection) and the LENGTH rules (resampy computes int(n / 2) outputs,
    librosa pads to ceil(n / 2) with zeros) -- to third-party code that is in this image.  The taps themselves are pinned
    by test_kaiser_fast_window_from_scipys_kaiser; the float32 per-tap rounding is resampy's own and stays unpinned."""
    scipy_signal = pytest.importorskip("scipy.signal")
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32)
    got = vqt_ref.resample2(x, accumulate="float64")
    n_out, n_fix = n // 2, (n + 1) // 2
    assert got.shape == (n_fix,) and got.dtype == np.float32
    left, right = vqt_ref.decimate2_taps()
    h = np.concatenate([right[::-1], left])                 # by input offset +31 .. +1, 0, -1 .. -31: h[k] multiplies x[2t + 31 - k]
    np.testing.assert_array_equal(h, h[::-1])                # a symmetric, odd-length (zero-phase) filter
    scale = np.float32(np.sqrt(0.5))
    x64 = x.astype(np.float64)
    # upfirdn: full convolution sampled at even indices; a leading zero tap puts output t + 16 on input 2t
    full = scipy_signal.upfirdn(np.concatenate([[0.0], h]), x64, up=1, down=2)
    want = np.zeros(n_fix, dtype=np.float32)
    want[:n_out] = full[16:16 + n_out].astype(np.float32)
    want /= scale
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-7 * max(1.0, float(np.abs(want).max())))
    # resample_poly centres an odd-length window itself and zero-pads the edges (padtype="constant")
    poly = scipy_signal.resample_poly(x64, 1, 2, window=h, padtype="constant")
    assert poly.shape == (n_fix,)
    want2 = np.zeros(n_fix, dtype=np.float32)
    want2[:n_out] = poly[:n_out].astype(np.float32)          # for odd n scipy computes one more real output; librosa pads a zero
    want2 /= scale
    np.testing.assert_allclose(got, want2, rtol=0, atol=2e-7 * max(1.0, float(np.abs(want2).max())))
    if n % 2:
        assert got[-1] == 0.0
    # and the shipped arithmetics are roundings of the same sums
    for acc in ("resampy", "fma32"):
        np.testing.assert_allclose(vqt_ref.resample2(x, accumulate=acc), got, rtol=0, atol=4e-6 * max(1.0, float(np.abs(got).max())))
