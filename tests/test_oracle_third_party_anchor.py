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
