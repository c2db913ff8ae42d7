"""numpy restatement of generate_XQT (TEST INFRASTRUCTURE; see oracle/__init__.py).

*** PARITY UNPINNED ***  The arithmetic of /root/reference/zeroNoteSamba/processing/input_rep.py:11-57
lives in third-party code that is NOT in this container and not vendored by the reference:
    librosa 0.8.1   (poetry.lock:1183-1184; pyproject.toml:34 `librosa = "0.8.*"`)
    resampy 0.4.2   (poetry.lock:2188-2189), numba JIT
and the reference has no tests or golden vectors for this function.  This file restates the published
algorithm of those versions from memory; what CAN be pinned without them is checked in tests/test_vqt.py
(output shape/dtype, sinusoid known-answer magnitude and bin, linearity, silence floor, CQT==VQT(gamma=0)).

Call chain restated (librosa 0.8.1 unless noted):
  input_rep.generate_XQT(signal, 16000, mode)                          input_rep.py:18-54
    -> librosa.vqt(y, sr=16000, hop_length=256, fmin=C0, n_bins=96, bins_per_octave=12)   [mode "vqt"]
       librosa.cqt(...) == vqt(..., gamma=0)                                               [mode "cqt"]
       core/constantq.py::vqt        octave recursion, early-downsample count = 0 here, res_type kaiser_fast
       core/constantq.py::__cqt_filter_fft / filters.py::constant_q, constant_q_lengths,
                                      window_bandwidth("hann") = 1.50018310546875, util.sparsify_rows(0.01)
       core/constantq.py::__cqt_response = sparse_basis.dot(stft(y, n_fft, hop, window="ones", center, reflect))
       core/audio.py::resample(y, 2, 1, "kaiser_fast", fix=True, scale=True) -> resampy 0.4.2 resample_f
       core/constantq.py::__trim_stack, final V /= sqrt(lengths)
    -> np.log(np.abs(V) + 1e-9)                                          input_rep.py:51-52 (eps = 10e-10)
"""
from functools import lru_cache
from typing import Dict, List, Tuple

import numpy as np

SR = 16000
HOP = 256
N_BINS = 96
BPO = 12
N_OCT = 8
EPS = 10e-10                       # input_rep.py:22
FMIN = 440.0 * 2.0 ** ((12 - 69) / 12.0)   # librosa.note_to_hz("C0") = midi 12 -> 16.351597831287414 Hz
HANN_BW = 1.50018310546875         # filters.WINDOW_BANDWIDTHS["hann"]
# resampy 0.4.2 data/kaiser_fast.npz: sinc_window(num_zeros=16, precision=9, rolloff=0.85,
#                                                  window=kaiser(beta=8.555504641634386))
KF_ZEROS, KF_PREC, KF_ROLLOFF, KF_BETA = 16, 9, 0.85, 8.555504641634386


def alpha() -> float:
    return 2.0 ** (1.0 / BPO) - 1.0


def gamma_for(mode: str) -> float:
    if mode == "vqt":
        return 24.7 * alpha() / 0.108     # constantq.py::vqt default gamma
    if mode == "cqt":
        return 0.0
    raise Exception("Mode can only be vqt or cqt!")   # input_rep.py:56-57


# ------------------------------------------------------------------ resampy kaiser_fast, 2:1
@lru_cache(maxsize=None)
def kaiser_fast_half_window() -> np.ndarray:
    """resampy/filters.py::sinc_window: right wing of the windowed sinc, length zeros*2^prec + 1."""
    num_bits = 2 ** KF_PREC
    n = num_bits * KF_ZEROS
    sinc_win = KF_ROLLOFF * np.sinc(KF_ROLLOFF * np.linspace(0, KF_ZEROS, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, KF_BETA)[n:]
    return taper * sinc_win


@lru_cache(maxsize=None)
def decimate2_taps() -> Tuple[np.ndarray, np.ndarray]:
    """Taps resampy's interpolation loop touches for sr 2 -> 1 (t_out = 2t exactly, so frac = 0, eta = 0):
       left wing  weight_i = win[i*step]       * x[2t - i],      i = 0 .. nwin//step - 1      (32 taps)
       right wing weight_k = win[(k+1)*step]   * x[2t + k + 1],  k = 0 .. (nwin-step)//step-1 (31 taps)
    with win = 0.5 * half_window (interp_win *= sample_ratio) and step = int(0.5 * 2^prec)."""
    win = kaiser_fast_half_window() * 0.5
    step = int(0.5 * (2 ** KF_PREC))
    nwin = win.shape[0]
    left = win[np.arange(nwin // step) * step]
    right = win[step + np.arange((nwin - step) // step) * step]
    return left.astype(np.float64), right.astype(np.float64)


def resample2(y: np.ndarray, accumulate: str = "resampy") -> np.ndarray:
    """librosa.resample(y, 2, 1, res_type='kaiser_fast', fix=True, scale=True) for float32 y.

    resampy accumulates `y[t] += weight * x[...]` into a float32 output array tap by tap (left wing
    first, then right wing): every partial sum is rounded to float32.  Restated exactly (accumulate="resampy").

    Two other arithmetics of the SAME taps in the SAME order, used by the tests to state how far apart equally valid
    roundings of this sum lie (they are not restatements of the reference):
      "fma32"   -- taps rounded once to float32, one float32 fused multiply-add per tap (the HIP default path)
      "float64" -- float64 accumulation, rounded to float32 once at the end (MORE accurate than resampy's own)"""
    y = np.asarray(y, dtype=np.float32)
    n_in = y.shape[0]
    n_out = int(n_in * 0.5)                     # resampy: int(shape * sample_ratio)
    left, right = decimate2_taps()
    if accumulate == "fma32":                   # a product of two float32 is exact in float64, so this IS the fused rounding
        left, right = left.astype(np.float32).astype(np.float64), right.astype(np.float32).astype(np.float64)
    elif accumulate not in ("resampy", "float64"):
        raise ValueError(accumulate)
    keep64 = accumulate == "float64"
    t2 = 2 * np.arange(n_out)
    acc = np.zeros(n_out, dtype=np.float64 if keep64 else np.float32)
    x64 = y.astype(np.float64)
    for i, w in enumerate(left):                # i_max = min(n + 1, len(left))
        idx = t2 - i
        ok = idx >= 0
        acc[ok] = (acc[ok].astype(np.float64) + w * x64[idx[ok]]).astype(acc.dtype)
    for k, w in enumerate(right):               # k_max = min(n_orig - n - 1, len(right))
        idx = t2 + k + 1
        ok = idx < n_in
        acc[ok] = (acc[ok].astype(np.float64) + w * x64[idx[ok]]).astype(acc.dtype)
    acc = acc.astype(np.float32)
    n_fix = int(np.ceil(n_in * 0.5))            # librosa: fix_length to ceil(n * ratio)
    if n_fix > n_out:
        acc = np.concatenate([acc, np.zeros(n_fix - n_out, dtype=np.float32)])
    acc /= np.float32(np.sqrt(0.5))             # scale=True: y_hat /= sqrt(ratio), float32 in place
    return acc.astype(np.float32)


def resample_general(x: np.ndarray, sr_orig: float, sr_new: float) -> np.ndarray:
    """resampy 0.4.2 `core.resample(x, sr_orig, sr_new, filter="kaiser_fast")` for 1-D float32 x, restated in its
    GENERAL form (any ratio): core.py computes
        sample_ratio = sr_new / sr_orig;  n_out = int(len(x) * sample_ratio);  y = zeros(n_out, dtype=x.dtype)
        interp_win = half window (x sample_ratio when downsampling);  interp_delta = diff(interp_win, append=last)
        scale = min(1, sample_ratio);  t_out = arange(n_out) / sample_ratio
    and interpn.py::_resample_loop walks, per output sample, the left wing (x[n], x[n-1], ...) and then the right wing
    (x[n+1], ...) of the windowed sinc with linear interpolation between table entries
        weight = interp_win[offset + i*index_step] + eta * interp_delta[offset + i*index_step],   y[t] += weight * x[...]
    where y is float32: every partial sum is rounded to float32.  Scalar loops, literal order: slow, small inputs only.
    `resample2` above is this loop specialised by hand to the exact 2:1 ratio (frac = 0, eta = 0); tests/test_vqt.py
    asserts the two agree bit for bit, so the specialisation is not taken on trust."""
    x = np.asarray(x, dtype=np.float32)
    sample_ratio = float(sr_new) / float(sr_orig)
    n_out = int(x.shape[0] * sample_ratio)
    y = np.zeros(n_out, dtype=np.float32)
    interp_win = kaiser_fast_half_window().astype(np.float64).copy()
    num_table = 2 ** KF_PREC
    if sample_ratio < 1:
        interp_win *= sample_ratio
    interp_delta = np.diff(interp_win, append=interp_win[-1])
    scale = min(1.0, sample_ratio)
    t_out = np.arange(n_out) * (1.0 / sample_ratio)
    index_step = int(scale * num_table)
    nwin = interp_win.shape[0]
    n_orig = x.shape[0]
    for t in range(n_out):
        time_register = t_out[t]
        n = int(time_register)
        frac = scale * (time_register - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(n + 1, (nwin - offset) // index_step)
        for i in range(i_max):
            weight = interp_win[offset + i * index_step] + eta * interp_delta[offset + i * index_step]
            y[t] = np.float32(np.float64(y[t]) + weight * np.float64(x[n - i]))
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_orig - n - 1, (nwin - offset) // index_step)
        for k in range(k_max):
            weight = interp_win[offset + k * index_step] + eta * interp_delta[offset + k * index_step]
            y[t] = np.float32(np.float64(y[t]) + weight * np.float64(x[n + k + 1]))
    return y


def librosa_resample_general(y: np.ndarray, sr_orig: float, sr_new: float) -> np.ndarray:
    """librosa 0.8.1 core/audio.py::resample(y, orig_sr, target_sr, res_type="kaiser_fast", fix=True, scale=True):
    resampy, then util.fix_length to ceil(n * ratio) (zero padded), then `y_hat /= np.sqrt(ratio)` in float32."""
    ratio = float(sr_new) / float(sr_orig)
    out = resample_general(y, sr_orig, sr_new)
    n_fix = int(np.ceil(len(y) * ratio))
    if n_fix > len(out):
        out = np.concatenate([out, np.zeros(n_fix - len(out), dtype=np.float32)])
    out = out[:n_fix].copy()
    out /= np.float32(np.sqrt(ratio))
    return out.astype(np.float32)


# ------------------------------------------------------------------ filter bank
def cq_lengths(sr: float, fmin: float, n_bins: int, gamma: float) -> np.ndarray:
    """filters.constant_q_lengths."""
    a = alpha()
    Q = 1.0 / a
    freq = fmin * (2.0 ** (np.arange(n_bins, dtype=float) / BPO))
    if freq[-1] * (1 + 0.5 * HANN_BW / Q) > sr / 2.0:
        raise ValueError("Filter pass-band lies beyond Nyquist")
    return Q * sr / (freq + gamma / a)


def hann_periodic(n: int) -> np.ndarray:
    """scipy.signal.get_window('hann', n, fftbins=True)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def octave_filters(sr: float, fmin: float, gamma: float):
    """filters.constant_q(sr, fmin, n_bins=12, pad_fft=True, norm=1, dtype=complex64) -> (basis[12,n_fft], lengths)."""
    lengths = cq_lengths(sr, fmin, BPO, gamma)
    freqs = fmin * (2.0 ** (np.arange(BPO, dtype=float) / BPO))
    filt = []
    for ilen, freq in zip(lengths, freqs):
        sig = np.exp(np.arange(-ilen // 2, ilen // 2, dtype=float) * 1j * 2 * np.pi * freq / sr)
        n = len(sig)
        win = hann_periodic(n)                  # __float_window('hann')(len(sig)) with integer length
        sig = sig * win
        sig = sig / np.sum(np.abs(sig))         # util.normalize(norm=1)
        filt.append(sig)
    max_len = int(2.0 ** (np.ceil(np.log2(max(lengths)))))
    basis = np.zeros((BPO, max_len), dtype=np.complex64)
    for k, sig in enumerate(filt):
        lpad = int((max_len - len(sig)) // 2)   # util.pad_center
        basis[k, lpad:lpad + len(sig)] = sig.astype(np.complex64)
    return basis, lengths


def sparsify_rows(x: np.ndarray, quantile: float = 0.01) -> np.ndarray:
    """util.sparsify_rows: zero the smallest-magnitude entries holding < quantile of each row's L1 mass."""
    mags = np.abs(x)
    norms = np.sum(mags, axis=1, keepdims=True)
    mag_sort = np.sort(mags, axis=1)
    cumulative_mag = np.cumsum(mag_sort / norms, axis=1)
    threshold_idx = np.argmin(cumulative_mag < quantile, axis=1)
    out = np.zeros_like(x)
    for i, j in enumerate(threshold_idx):
        idx = np.where(mags[i] >= mag_sort[i, j])
        out[i, idx] = x[i, idx]
    return out


@lru_cache(maxsize=None)
def filter_bank(mode: str):
    """Per octave i (0 = top): dict(n_fft, hop, fft_basis complex64 [12, n_fft/2+1] incl. sqrt(2^i), lengths)."""
    g = gamma_for(mode)
    freqs_top = FMIN * (2.0 ** (np.arange(N_BINS, dtype=float) / BPO))[-BPO:]
    fmin_t = float(np.min(freqs_top))
    bank = []
    for i in range(N_OCT):
        sr_i = SR / (2.0 ** i)
        basis, lengths = octave_filters(sr_i, fmin_t * 2.0 ** -i, g)
        n_fft = basis.shape[1]
        basis = (basis * (lengths[:, np.newaxis] / float(n_fft))).astype(np.complex64)   # in-place on complex64
        fft_basis = np.fft.fft(basis.astype(np.complex128), n=n_fft, axis=1)[:, : (n_fft // 2) + 1]
        fft_basis = sparsify_rows(fft_basis, 0.01).astype(np.complex64)
        fft_basis = (fft_basis.astype(np.complex128) * np.sqrt(2 ** i)).astype(np.complex64)   # fft_basis[:] *= sqrt(2**i)
        bank.append(dict(n_fft=n_fft, hop=HOP // (2 ** i), fft_basis=fft_basis, lengths=lengths))
    full_lengths = cq_lengths(SR, FMIN, N_BINS, g)
    return bank, full_lengths


def time_domain_bank(mode: str):
    """Equivalent strided FIR bank the HIP kernel applies: V[k,t] = sum_n g[k,n] * ypad[t*hop + n],
    g[k,n] = sum_f B[k,f] exp(-2 pi j f n / n_fft)  (rfft then sparse product, folded), including the
    final 1/sqrt(length) scale.  float64 complex [12, n_fft] per octave."""
    bank, full_lengths = filter_bank(mode)
    out = []
    for i, o in enumerate(bank):
        n_fft = o["n_fft"]
        f = np.arange(n_fft // 2 + 1)
        n = np.arange(n_fft)
        E = np.exp(-2j * np.pi * np.outer(f, n) / n_fft)
        gk = o["fft_basis"].astype(np.complex128) @ E
        k0 = N_BINS - BPO * (i + 1)
        gk = gk / np.sqrt(full_lengths[k0:k0 + BPO])[:, None]
        out.append(gk)
    return out


def stft_ones(y: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    """core/spectrum.py::stft(window='ones', center=True, pad_mode='reflect', dtype=complex64)."""
    ypad = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(ypad) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    frames = ypad[idx]
    return np.fft.rfft(frames.astype(np.float64), axis=0).astype(np.complex64)


def xqt_complex(y: np.ndarray, mode: str = "vqt", accumulate: str = "resampy") -> np.ndarray:
    """librosa.vqt / librosa.cqt as called by generate_XQT: complex64 [96, 1 + len(y)//256].
    accumulate: arithmetic of the decimation sums (resample2); anything but "resampy" is a test aid, not the reference."""
    y = np.asarray(y, dtype=np.float32)
    bank, full_lengths = filter_bank(mode)
    resp: List[np.ndarray] = []
    my_y = y
    for i, o in enumerate(bank):
        if i > 0:
            if len(my_y) < 2:
                raise ValueError("Input signal length is too short for 8-octave CQT/VQT")
            my_y = resample2(my_y, accumulate)
        if len(my_y) <= o["n_fft"] // 2:
            raise ValueError("Input signal length is too short for reflect padding")
        D = stft_ones(my_y, o["n_fft"], o["hop"])
        resp.append((o["fft_basis"] @ D).astype(np.complex64))
    max_col = min(r.shape[1] for r in resp)
    V = np.empty((N_BINS, max_col), dtype=np.complex64)
    end = N_BINS
    for r in resp:                               # __trim_stack: octaves from the top down
        V[end - BPO:end] = r[:, :max_col]
        end -= BPO
    # numpy's in-place `complex64 /= float64 array` computes the quotient in complex128 and casts back
    V[...] = (V.astype(np.complex128) / np.sqrt(full_lengths[:, np.newaxis])).astype(np.complex64)
    return V


def generate_XQT(signal: np.ndarray, sample_rate: int, mode: str, accumulate: str = "resampy") -> np.ndarray:
    """input_rep.generate_XQT restated: float32 [96, T]."""
    if mode not in ("vqt", "cqt"):
        raise Exception("Mode can only be vqt or cqt!")
    V = xqt_complex(signal, mode, accumulate)
    return np.log(np.abs(V) + EPS).astype(np.float32)


def xqt_time_domain(y: np.ndarray, mode: str = "vqt") -> np.ndarray:
    """Same transform through the folded FIR bank, float64 (what the HIP kernel computes in fp32)."""
    y = np.asarray(y, dtype=np.float32)
    tb = time_domain_bank(mode)
    bank, _ = filter_bank(mode)
    n_frames = 1 + len(y) // HOP
    V = np.zeros((N_BINS, n_frames), dtype=np.complex128)
    my_y = y
    for i, (o, gk) in enumerate(zip(bank, tb)):
        if i > 0:
            my_y = resample2(my_y)
        n_fft, hop = o["n_fft"], o["hop"]
        ypad = np.pad(my_y.astype(np.float64), n_fft // 2, mode="reflect")
        nf = min(n_frames, 1 + (len(ypad) - n_fft) // hop)
        idx = np.arange(n_fft)[:, None] + hop * np.arange(nf)[None, :]
        k0 = N_BINS - BPO * (i + 1)
        V[k0:k0 + BPO, :nf] = gk @ ypad[idx]
    return V


# ------------------------------------------------------------------ independent derivation: brute-force correlator
def brute_force_response(y: np.ndarray, mode: str = "vqt", octaves=range(N_OCT)) -> Dict[int, np.ndarray]:
    """A second, FFT-free derivation of each octave's response straight from the `filters.constant_q` definition.

    librosa forms  R[k,t] = sum_{f=0}^{N/2} FFT(b_k)[f] * D[f,t],  D[f,t] = sum_n ypad[t*hop+n] e^{-2 pi i f n / N}
    (the basis spectrum is NOT conjugated).  Summed over ALL N frequencies this is, by orthogonality of the DFT
    exponentials, the plain time-domain sum
        R_full[k,t] = N * sum_n b_k[(-n) mod N] * ypad[t*hop + n]
    which needs no FFT, no sparsification and no folding.  What librosa computes differs from it only by (1) the
    negative-frequency half of FFT(b_k) it never forms (b_k is a windowed e^{+i w n}: nearly analytic) and (2) the
    entries sparsify_rows(1 %) zeroes; both are bounded per (k,t) by `brute_force_bound`.  Returns {octave: complex128
    [12, frames]} INCLUDING the sqrt(2^i) rescale and the final 1/sqrt(length), i.e. comparable to rows of xqt_complex."""
    y = np.asarray(y, dtype=np.float32)
    g = gamma_for(mode)
    freqs_top = FMIN * (2.0 ** (np.arange(N_BINS, dtype=float) / BPO))[-BPO:]
    fmin_t = float(np.min(freqs_top))
    full_lengths = cq_lengths(SR, FMIN, N_BINS, g)
    out: Dict[int, np.ndarray] = {}
    my_y = y
    n_frames = 1 + len(y) // HOP
    for i in range(N_OCT):
        if i > 0:
            my_y = resample2(my_y)
        if i not in octaves:
            continue
        basis, lengths = octave_filters(SR / 2.0 ** i, fmin_t * 2.0 ** -i, g)
        N = basis.shape[1]
        b = (basis * (lengths[:, None] / float(N))).astype(np.complex64).astype(np.complex128)
        hop = HOP // (2 ** i)
        ypad = np.pad(my_y.astype(np.float64), N // 2, mode="reflect")
        nf = min(n_frames, 1 + (len(ypad) - N) // hop)
        rev = b[:, (-np.arange(N)) % N]                       # b_k[(-n) mod N]
        R = np.zeros((BPO, nf), dtype=np.complex128)
        for t in range(nf):                                   # plain correlation, frame by frame
            R[:, t] = N * (rev @ ypad[t * hop:t * hop + N])
        k0 = N_BINS - BPO * (i + 1)
        out[i] = R * np.sqrt(2.0 ** i) / np.sqrt(full_lengths[k0:k0 + BPO])[:, None]
    return out


def brute_force_bound(y: np.ndarray, mode: str, i: int) -> np.ndarray:
    """Per (k,t) bound on |librosa response - brute_force_response| for octave i:
         ( L1 mass of the dropped positive-frequency entries  +  L1 mass of the negative-frequency half ) * max_f |D[f,t]|
    scaled like the responses.  [12, frames]."""
    y = np.asarray(y, dtype=np.float32)
    g = gamma_for(mode)
    bank, full_lengths = filter_bank(mode)
    freqs_top = FMIN * (2.0 ** (np.arange(N_BINS, dtype=float) / BPO))[-BPO:]
    fmin_t = float(np.min(freqs_top))
    my_y = y
    for _ in range(i):
        my_y = resample2(my_y)
    basis, lengths = octave_filters(SR / 2.0 ** i, fmin_t * 2.0 ** -i, g)
    N = basis.shape[1]
    b = (basis * (lengths[:, None] / float(N))).astype(np.complex64).astype(np.complex128)
    F = np.fft.fft(b, axis=1)
    kept = bank[i]["fft_basis"].astype(np.complex128) / np.sqrt(2.0 ** i)
    dropped = np.abs(F[:, :N // 2 + 1] - kept).sum(axis=1)
    negative = np.abs(F[:, N // 2 + 1:]).sum(axis=1)
    hop = HOP // (2 ** i)
    ypad = np.pad(my_y.astype(np.float64), N // 2, mode="reflect")
    n_frames = 1 + len(y) // HOP
    nf = min(n_frames, 1 + (len(ypad) - N) // hop)
    idx = np.arange(N)[:, None] + hop * np.arange(nf)[None, :]
    Dmax = np.abs(np.fft.fft(ypad[idx], axis=0)).max(axis=0)
    k0 = N_BINS - BPO * (i + 1)
    scale = np.sqrt(2.0 ** i) / np.sqrt(full_lengths[k0:k0 + BPO])
    return (dropped + negative)[:, None] * Dmax[None, :] * scale[:, None]
