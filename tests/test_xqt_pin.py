"""The generate_XQT / CQT / feature.rms pin.  tests/golden/xqt_pin.npz is written by
tests/golden/make_golden_xqt.py on a machine that HAS librosa 0.8.1 + resampy 0.4.2 (this image does not and cannot install
them); until that file is committed these tests SKIP with the reason spelled out, and the oracle stays "parity unpinned".

  CPU  : oracle/vqt_ref.generate_XQT and oracle/stem_check_ref.feature_rms against the fixtures
  GPU  : the device (default, per-stage, exact pyramid) against the fixtures, through the reference's own signature
  here : the kit's plumbing, with the oracle standing in for librosa (NOT a pin: it proves the script and these tests agree
         on signals, keys, shapes and the refusal cases, so the one command works the first time someone runs it)."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import stem_check_ref, vqt_ref

HERE = os.path.dirname(os.path.abspath(__file__))
PIN = os.path.join(HERE, "golden", "xqt_pin.npz")
ABSENT = ("fixtures absent: parity unpinned -- tests/golden/xqt_pin.npz has to be generated where librosa 0.8.1 + resampy "
          "0.4.2 are installed: python tests/golden/make_golden_xqt.py")


def kit():
    spec = importlib.util.spec_from_file_location("make_golden_xqt", os.path.join(HERE, "golden", "make_golden_xqt.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def lin_err(got_log, ref_log):
    g, r = np.exp(got_log.astype(np.float64)), np.exp(ref_log.astype(np.float64))
    return np.abs(g - r).max() / r.max()


def check_oracle_against(z, K=None):
    """The restatement against a fixture file: every bin of the log output to 1e-4 (north_star), 1e-6 of the loudest bin in
    the linear domain; refusals where librosa refuses; RMS to float32 rounding."""
    K = K or kit()
    worst = {}
    for n in K.LENGTHS:
        y = K.xqt_signal(n)
        np.testing.assert_array_equal(y[:8], z["head_%d" % n])          # same signal on both sides
        for mode in K.MODES:
            key = "%s_%d" % (mode, n)
            if key + "_raises" in z:
                with pytest.raises(Exception):
                    vqt_ref.generate_XQT(y, 16000, mode)
                continue
            ref = z[key]
            got = vqt_ref.generate_XQT(y, 16000, mode)
            assert got.shape == ref.shape and got.dtype == np.float32
            worst[key] = (lin_err(got, ref), float(np.abs(got.astype(np.float64) - ref).max()))
            assert worst[key][0] < 1e-6 and worst[key][1] < 1e-4, (key, worst[key])
    for name, y in K.rms_signals().items():
        np.testing.assert_array_equal(y[:8], z["rms_head_" + name])
        got = stem_check_ref.feature_rms(y)
        assert got.shape == z["rms_" + name].shape
        np.testing.assert_allclose(got, z["rms_" + name], rtol=2e-6, atol=0)
    return worst


def test_pin_kit_plumbing_with_the_oracle_standing_in_for_librosa(tmp_path, monkeypatch):
    """NOT a pin.  make_golden_xqt.main() with librosa / resampy faked by the oracle itself: the file it writes carries every
    key the consumers read, the cqt refusal on short rows is recorded as `_raises`, and check_oracle_against() passes on it."""
    import sys
    import types
    K = kit()
    fake_l = types.ModuleType("librosa")
    fake_l.__version__ = "0.8.1"
    fake_r = types.ModuleType("resampy")
    fake_r.__version__ = "0.4.2"
    monkeypatch.setitem(sys.modules, "librosa", fake_l)
    monkeypatch.setitem(sys.modules, "resampy", fake_r)
    monkeypatch.setattr(K, "load_reference", lambda: (vqt_ref.generate_XQT, stem_check_ref.feature_rms, "oracle-stand-in"))
    monkeypatch.setattr(K, "LENGTHS", (33333, 4097))                     # seconds, not minutes
    K.main(str(tmp_path))
    z = np.load(tmp_path / "xqt_pin.npz")
    assert str(z["source"]) == "oracle-stand-in" and str(z["librosa"]) == "0.8.1"
    assert "vqt_33333" in z and "cqt_33333" in z and "vqt_4097" in z and "cqt_4097_raises" in z
    assert z["vqt_33333"].shape == (96, 131) and z["rms_short"].shape == (1, 10)
    worst = check_oracle_against(z, K)                                   # the consumer side, same (reduced) set of lengths
    assert set(worst) == {"vqt_33333", "cqt_33333", "vqt_4097"} and all(v == (0.0, 0.0) for v in worst.values())


def test_oracle_matches_the_librosa_fixtures():
    if not os.path.exists(PIN):
        pytest.skip(ABSENT)
    z = np.load(PIN)
    assert str(z["source"]) in ("reference", "librosa-direct")
    worst = check_oracle_against(z)
    print("oracle vs librosa %s / resampy %s (%s):" % (z["librosa"], z["resampy"], z["source"]))
    for k, (le, ge) in sorted(worst.items()):
        print("  %-12s linear %.2e of the loudest bin, log %.2e" % (k, le, ge))


@pytest.mark.gpu
@pytest.mark.parametrize("impl", ["default", "separate", "exact"])
def test_device_matches_the_librosa_fixtures(impl):
    if not os.path.exists(PIN):
        pytest.skip(ABSENT)
    import torch
    from test_vqt import _flags, assert_matches_oracle
    from zeronotesamba_amd.processing.input_rep import generate_XQT, generate_XQT_batch
    K = kit()
    z = np.load(PIN)
    for n in K.LENGTHS:
        y = K.xqt_signal(n)
        for mode in K.MODES:
            key = "%s_%d" % (mode, n)
            if key + "_raises" in z:
                with pytest.raises(RuntimeError):
                    generate_XQT(y, 16000, mode)
                continue
            if impl == "default":
                got = generate_XQT(y, 16000, mode)
            else:
                got = generate_XQT_batch(torch.from_numpy(y).cuda().reshape(1, -1), 16000, mode,
                                         stages=3 | _flags(impl))[0].cpu().numpy()
            assert_matches_oracle(got, z[key], impl)


@pytest.mark.gpu
def test_device_rms_matches_the_librosa_fixtures():
    if not os.path.exists(PIN):
        pytest.skip(ABSENT)
    from zeronotesamba_amd.processing.stem_check import compute_rms
    K = kit()
    z = np.load(PIN)
    for name, y in K.rms_signals().items():
        got = np.asarray(compute_rms(y)[0])
        np.testing.assert_allclose(got.reshape(z["rms_" + name].shape), z["rms_" + name], rtol=1e-5, atol=0)
