"""CPU: the oracle's restatement of the online clip selection (stem_check.py:8-51, pretext.py:30-86).

librosa is not installed, so `feature_rms` is pinned by known answers instead of by the library (PARITY UNPINNED
is stated in oracle/stem_check_ref.py); the selection loop is plain Python and is checked for the reference's
draw-by-draw behaviour (randint consumption, lower_p halving after the 10th try)."""
import random

import numpy as np
import pytest

from oracle import stem_check_ref as sc


def gated_pair(n=16000 * 25, seed=0):
    """positive = white noise; anchor = positive * gain, gain 0.6 for 2 s then 0.1 for 3 s, repeating: any 10 s
    crop has ~40 % of its frames inside the (0.5, 4) ratio window."""
    rng = np.random.default_rng(seed)
    pos = (0.1 * rng.standard_normal(n)).astype(np.float32)
    t = np.arange(n) % (5 * 16000)
    gain = np.where(t < 2 * 16000, 0.6, 0.1).astype(np.float32)
    return (pos * gain).astype(np.float32), pos


def test_feature_rms_known_answers():
    c = np.full(16000, -0.3, np.float32)
    r = sc.feature_rms(c)
    assert r.shape == (1, 1 + 16000 // 512) and r.dtype == np.float32
    np.testing.assert_allclose(r, 0.3, rtol=1e-6)
    t = np.arange(160000)
    s = (0.5 * np.sin(2 * np.pi * 64 * t / 2048)).astype(np.float32)       # 64 whole periods per frame
    r = sc.feature_rms(s)
    assert r.shape == (1, 313)
    np.testing.assert_allclose(r[0, 4:-4], 0.5 / np.sqrt(2), rtol=2e-6)
    np.testing.assert_allclose(sc.feature_rms(3 * s), 3 * r, rtol=1e-6)
    # edge frames by brute force: frame f covers padded samples [512 f, 512 f + 2048), padded = reflect(1024)
    x = np.random.default_rng(1).standard_normal(5000).astype(np.float32)
    got = sc.feature_rms(x)[0]
    for f in (0, 1, len(got) - 1):
        idx = np.arange(512 * f, 512 * f + 2048) - 1024
        idx = np.abs(idx)
        idx = np.where(idx >= 5000, 2 * 4999 - idx, idx)
        want = np.sqrt(np.mean(x[idx].astype(np.float64) ** 2))
        assert abs(got[f] - want) < 1e-6 * want
    rms, m, sd = sc.compute_rms(x)
    assert abs(m - rms.mean()) < 1e-7 and abs(sd - rms.std()) < 1e-7


def test_gate_thresholds():
    rng = np.random.default_rng(2)
    p = rng.standard_normal(160000).astype(np.float32)
    assert sc.gate_fraction(0.6 * p, p) == 1.0 and sc.gate_fraction(0.4 * p, p) == 0.0
    assert sc.gate_fraction(3.9 * p, p) == 1.0 and sc.gate_fraction(4.1 * p, p) == 0.0
    assert sc.check_CL_clips(0.6 * p, p, 0.5, 1.0) and not sc.check_CL_clips(0.6 * p, p, 0.5, 0.9)
    assert not sc.check_CL_clips(0.4 * p, p, 0.0, 1.0)       # lower bound is strict: 0 < 0 is false
    a, pos = gated_pair()
    frac = sc.gate_fraction(a[:160000], pos[:160000])
    assert 0.35 < frac < 0.42


def test_selection_loop_draws_and_halving():
    a, pos = gated_pair()
    stems = {"bass": a * 0.5, "drums": pos, "other": a * 0.25, "vocals": a * 0.25}   # anchor mix == a exactly
    calls = []

    def fake_xqt(sig, sr, mode):
        calls.append((len(sig), sr, mode))
        return np.zeros((96, 1 + len(sig) // 256), np.float32)

    yml = {"clip_len": 10, "input_mode": "vqt", "lower_p": 2.0, "upper_p": 1.0}
    r = random.Random(123)
    anchor, drums, xa, xp, start = sc.drum_anchor_positive(stems, yml, fake_xqt, rng=r)
    np.testing.assert_array_equal(anchor, a)
    np.testing.assert_array_equal(drums, pos)
    assert xa.shape == (96, 626) and calls == [(160000, 16000, "vqt")] * 2
    # lower_p 2.0 can only pass after three halvings (tries 10, 11, 12) -> the 13th draw is accepted
    chk = random.Random(123)
    draws = [chk.randint(0, len(a) - 160000 - 1) for _ in range(13)]
    assert start == draws[-1] and r.getstate() == chk.getstate()
    # a passing first draw consumes exactly one
    yml["lower_p"] = 0.2
    r = random.Random(5)
    *_, start = sc.drum_anchor_positive(stems, yml, fake_xqt, rng=r)
    chk = random.Random(5)
    assert start == chk.randint(0, len(a) - 160000 - 1) and r.getstate() == chk.getstate()
    with pytest.raises(Exception, match="Anchor is still None"):
        sc.drum_anchor_positive({"drums": pos}, yml, fake_xqt)
