"""GPU: the online pretext path through the C ABI against oracle/stem_check_ref.py:
stem mix, RMS gate, the RMS-gated random crop (same randint draws as the reference's loop), the XQT pair, and
the device-resident bank feeding ShiftBank + PretextTrainer."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import stem_check_ref as sc  # noqa: E402
from oracle import vqt_ref  # noqa: E402
from tests.test_oracle_online import gated_pair  # noqa: E402

DEV = "cuda:0"


def test_rms_gate_kernel_matches_oracle():
    from zeronotesamba_amd.processing import stem_check
    rng = np.random.default_rng(0)
    for n in (5000, 160000, 33333):
        p = (0.2 * rng.standard_normal(n)).astype(np.float32)
        env = (0.3 + 0.7 * (rng.random(n // 997 + 1).repeat(997)[:n] > 0.5)).astype(np.float32)
        a = (p * env).astype(np.float32)
        ad, pd = torch.from_numpy(a).to(DEV), torch.from_numpy(p).to(DEV)
        counts, n_frames, ra, rp = stem_check.gate_counts(ad, pd, [0], n, want_rms=True)
        assert n_frames == 1 + n // 512
        np.testing.assert_allclose(ra.cpu().numpy(), sc.feature_rms(a), rtol=2e-6)
        np.testing.assert_allclose(rp.cpu().numpy(), sc.feature_rms(p), rtol=2e-6)
        assert counts.item() == round(sc.gate_fraction(a, p) * n_frames)
        rms, m, sd = stem_check.compute_rms(a)
        ref, rm, rsd = sc.compute_rms(a)
        assert rms.shape == ref.shape and abs(m - rm) < 1e-6 and abs(sd - rsd) < 1e-6
        for lo, hi in ((0.1, 0.9), (0.0, 0.3), (0.6, 1.0)):
            assert stem_check.check_CL_clips(a, p, lo, hi) == sc.check_CL_clips(a, p, lo, hi)
    # several candidate crops in one launch == one launch per crop
    starts = [0, 17, 4001, n - 20000]
    many, _, _, _ = stem_check.gate_counts(ad, pd, starts, 20000)
    for s, c in zip(starts, many.tolist()):
        assert c == round(sc.gate_fraction(a[s:s + 20000], p[s:s + 20000]) * (1 + 20000 // 512))
    with pytest.raises(RuntimeError, match="reflect"):
        stem_check.gate_counts(ad, pd, [0], 1024)
    with pytest.raises(IndexError):
        stem_check.gate_counts(ad, pd, [n - 10], 20000)


def test_mix_and_selection_follow_the_reference_loop():
    from zeronotesamba_amd import pretext
    a, pos = gated_pair()
    rng = np.random.default_rng(3)
    stems = {"bass": (a * rng.random(len(a))).astype(np.float32), "drums": pos,
             "other": (0.3 * rng.standard_normal(len(a))).astype(np.float32), "vocals": a}
    ref_anchor, ref_pos = sc.mix_stems(stems)
    anchor, p = pretext.mix_stems(stems)
    np.testing.assert_array_equal(anchor.cpu().numpy(), ref_anchor)      # float32 adds in dict order: bitwise
    np.testing.assert_array_equal(p.cpu().numpy(), ref_pos)
    with pytest.raises(Exception, match="Anchor is still None"):
        pretext.mix_stems({"drums": pos})
    # the halving schedule and the generator state, with speculation crossing a batch boundary
    ad, pd = torch.from_numpy(a).to(DEV), torch.from_numpy(pos).to(DEV)
    for seed, lower_p, n_draws in ((123, 2.0, 13), (5, 0.2, 1)):
        r = random.Random(seed)
        start = pretext.select_clip(ad, pd, 10, lower_p, 1.0, rng=r, speculate=8)
        chk = random.Random(seed)
        draws = [chk.randint(0, len(a) - 160000 - 1) for _ in range(n_draws)]
        assert start == draws[-1] and r.getstate() == chk.getstate()
    with pytest.raises(ValueError, match="shorter"):
        pretext.select_clip(ad[:1000], pd[:1000], 10, 0.1, 1.0)


def test_drum_anchor_positive_and_resident_bank():
    from zeronotesamba_amd import pretext
    from zeronotesamba_amd.models.models import Pretext_CNN
    a, pos = gated_pair(16000 * 12, seed=7)
    stems = {"bass": a, "drums": pos}
    yml = {"clip_len": 10, "input_mode": "vqt", "lower_p": 0.2, "upper_p": 1.0}
    r1, r2 = random.Random(11), random.Random(11)
    anchor, drums, xa, xp = pretext.drum_anchor_positive(stems, yml, rng=r1)
    ra, rd, rxa, rxp, start = sc.drum_anchor_positive(stems, yml, vqt_ref.generate_XQT, rng=r2)
    assert r1.getstate() == r2.getstate()
    np.testing.assert_array_equal(anchor, ra)
    np.testing.assert_array_equal(drums, rd)
    assert xa.shape == rxa.shape == (96, 626) and xa.dtype == np.float32
    assert np.abs(xa - rxa).max() < 1e-4 and np.abs(xp - rxp).max() < 1e-4
    with pytest.raises(Exception, match="Mode can only be vqt or cqt!"):
        pretext.drum_anchor_positive(stems, dict(yml, input_mode="mel"), rng=random.Random(1))
    # bank fill: same draws as calling drum_anchor_positive per file, rows land in fps order
    tracks = {f"t{i}": {"bass": np.roll(a, 1000 * i), "drums": np.roll(pos, 1000 * i)} for i in range(3)}
    fps = ["t2", "t0", "t1"]
    bank = pretext.fill_memory_bank(tracks, fps, yml, rng=random.Random(21), chunk=2)
    assert tuple(bank.shape) == (3, 2, 96, 626) and bank.is_cuda
    rr = random.Random(21)
    for row, fp in enumerate(fps):
        _, _, ea, ep = pretext.drum_anchor_positive(tracks[fp], yml, rng=rr)
        np.testing.assert_array_equal(bank[row, 0].cpu().numpy(), ea)
        np.testing.assert_array_equal(bank[row, 1].cpu().numpy(), ep)
    # ... and feeds the shift batching + fused trainer without leaving the device
    sb = pretext.ShiftBank(bank, batch_len=4, crop=64, device=DEV)
    tr = pretext.PretextTrainer(Pretext_CNN().to(DEV), batch_len=4, lr=1e-6, seed=1)
    out = tr.step(sb.batch(1, rng=random.Random(2))).tolist()
    assert all(np.isfinite(out))


def test_fill_memory_bank_group_speculation_consumes_the_reference_draws():
    """fill_memory_bank gates several files per host synchronisation (first-draw guess, rewind on a miss).  Whatever
    the pass rate, the crop starts and the generator state must equal the per-file reference loop's (pretext.py:66-84
    run file after file on ONE random stream): a mix of files that pass at once, files that need several draws and
    files that only pass after lower_p has been halved."""
    from zeronotesamba_amd import pretext
    from zeronotesamba_amd.processing.input_rep import generate_XQT_batch
    n = 16000 * 12
    tracks = {}
    for i in range(9):
        a, pos = gated_pair(n, seed=30 + i)
        if i % 3 == 1:                         # in-window gain for the first 6 s only: the share of passing frames
            t = np.arange(n)                   # (0.6 .. 0.4) now depends on the crop start -> at lower_p = 0.55 a
            a = (pos * np.where(t < 6 * 16000, 0.6, 0.1)).astype(np.float32)   # quarter of the draws pass
        tracks[f"f{i}"] = {"bass": a, "drums": pos}
    fps = [f"f{i}\n" for i in (4, 0, 7, 1, 2, 8, 3, 5, 6)]                 # file lists carry the newline (pretext.py:146)
    for lower_p in (0.2, 0.55, 2.0):
        yml = {"clip_len": 10, "input_mode": "vqt", "lower_p": lower_p, "upper_p": 1.0}
        # reference order of events on one stream (the oracle's loop; its own XQT is not needed here)
        rr = random.Random(77)
        want = [sc.drum_anchor_positive(tracks[fp.strip()], yml, lambda y, sr, m: None, rng=rr)[4] for fp in fps]
        for fpsync in (1, 4, 32):
            r = random.Random(77)
            bank = pretext.fill_memory_bank(tracks, fps, yml, rng=r, chunk=4, files_per_sync=fpsync)
            assert r.getstate() == rr.getstate(), (lower_p, fpsync)
            assert tuple(bank.shape) == (9, 2, 96, 626)
            for row in (0, 3, 8):
                tr = tracks[fps[row].strip()]
                s0 = want[row]
                crops = torch.from_numpy(np.stack([tr["bass"][s0:s0 + 160000], tr["drums"][s0:s0 + 160000]])).to(DEV)
                exp = generate_XQT_batch(crops, 16000, "vqt")
                assert torch.equal(bank[row], exp), (lower_p, fpsync, row)
