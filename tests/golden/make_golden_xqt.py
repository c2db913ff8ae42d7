#!/usr/bin/env python3
"""The pin kit for generate_XQT / CQT / librosa.feature.rms ("parity unpinned" until this has run).

The arithmetic of processing/input_rep.py:27-49 and processing/stem_check.py:14 lives in librosa 0.8.1 + resampy 0.4.2
(poetry.lock:1183-1184, 2188-2189), which the build image does not have and cannot install.  Anyone who HAS them pins the
oracle (oracle/vqt_ref.py, oracle/stem_check_ref.py) and the device in one command, from the repo root:

    PYTHONPATH=/path/to/zeroNoteSamba/parent:. python tests/golden/make_golden_xqt.py
    python -m pytest tests/test_xqt_pin.py            # CPU: oracle vs fixtures;  -m gpu: device vs fixtures

It imports the reference's own `zeroNoteSamba.processing.input_rep.generate_XQT` and
`zeroNoteSamba.processing.stem_check.compute_rms` when the reference package is importable (source "reference"), else
calls librosa with exactly the arguments those two functions pass (source "librosa-direct"; same third-party code path),
runs them on the seeded signals the parity tests already use, and stores only DATA next to this script:

    xqt_pin.npz   for n in {160000, 80000, 33333, 4097} x mode in {vqt, cqt}: the (96, 1 + n // 256) float32 output, or
                  the exception's type name where librosa refuses the input (cqt on rows too short for 8 octaves);
                  librosa.feature.rms(frame_length=2048, hop_length=512) of three signals; library versions; source.

Signals are regenerated from seeds on both sides (numpy Generator streams are stable across numpy versions for
standard_normal); the first 8 samples of each are stored as a guard.  The reference never travels.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

LENGTHS = (160000, 80000, 33333, 4097)
MODES = ("vqt", "cqt")


def noise(seed, n, sigma=0.1):
    """tests/test_vqt.py::noise -- white noise + a chirp, float32."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    chirp = 0.05 * np.sin(2 * np.pi * (200.0 * t + 0.5 * 300.0 * t * t))
    return (sigma * rng.standard_normal(n) + chirp).astype(np.float32)


def xqt_signal(n):
    return noise(10 + n % 7, n)


def rms_signals():
    """tests/test_oracle_online.py: the gated pair's first 10 s (anchor, positive) and the 5000-sample edge-frame case."""
    rng = np.random.default_rng(0)
    n = 16000 * 25
    pos = (0.1 * rng.standard_normal(n)).astype(np.float32)
    t = np.arange(n) % (5 * 16000)
    gain = np.where(t < 2 * 16000, 0.6, 0.1).astype(np.float32)
    anchor = (pos * gain).astype(np.float32)
    x = np.random.default_rng(1).standard_normal(5000).astype(np.float32)
    return {"anchor": anchor[:160000], "positive": pos[:160000], "short": x}


def load_reference():
    """(generate_XQT, feature_rms, source)."""
    try:
        from zeroNoteSamba.processing.input_rep import generate_XQT          # input_rep.py:11
        from zeroNoteSamba.processing.stem_check import compute_rms          # stem_check.py:8
        return generate_XQT, (lambda y: compute_rms(y)[0]), "reference"
    except Exception as e:                                                   # reference absent, or matplotlib etc. missing
        print("reference package not importable (%s: %s): calling librosa with its arguments" % (type(e).__name__, e))
    import librosa

    def generate_XQT(signal, sample_rate, mode):                             # the calls of input_rep.py:27-34, 42-49
        fmin = librosa.note_to_hz("C0")
        fn = {"cqt": librosa.cqt, "vqt": librosa.vqt}[mode]
        X = fn(y=signal, sr=sample_rate, hop_length=256, fmin=fmin, n_bins=96, bins_per_octave=12)
        return np.log(np.abs(X) + 10e-10)

    def feature_rms(y):                                                      # stem_check.py:14
        return librosa.feature.rms(y=y, frame_length=2048, hop_length=512)

    return generate_XQT, feature_rms, "librosa-direct"


def main(out_dir=HERE):
    gen, rms, source = load_reference()
    import librosa
    import resampy
    out = {"source": np.array(source), "librosa": np.array(librosa.__version__), "resampy": np.array(resampy.__version__),
           "numpy": np.array(np.__version__)}
    if not librosa.__version__.startswith("0.8.") or resampy.__version__ != "0.4.2":
        print("WARNING: the reference pins librosa 0.8.1 / resampy 0.4.2, found %s / %s" % (librosa.__version__, resampy.__version__))
    for n in LENGTHS:
        y = xqt_signal(n)
        out["head_%d" % n] = y[:8].copy()
        for mode in MODES:
            key = "%s_%d" % (mode, n)
            try:
                X = np.asarray(gen(y, 16000, mode))
                assert X.shape == (96, 1 + n // 256), X.shape
                out[key] = np.ascontiguousarray(X, dtype=np.float32)
                out[key + "_dtype"] = np.array(str(X.dtype))
                print(key, X.shape, X.dtype, "min %.4f max %.4f" % (X.min(), X.max()))
            except Exception as e:                                           # librosa.ParameterError for too-short rows
                out[key + "_raises"] = np.array(type(e).__name__)
                print(key, "raises", type(e).__name__, e)
    for name, y in rms_signals().items():
        r = np.asarray(rms(y))
        out["rms_" + name] = r.astype(np.float32)
        out["rms_" + name + "_dtype"] = np.array(str(r.dtype))
        out["rms_head_" + name] = y[:8].copy()
        print("rms", name, r.shape, r.dtype)
    path = os.path.join(out_dir, "xqt_pin.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else HERE)
