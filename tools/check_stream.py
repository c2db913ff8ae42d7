#!/usr/bin/env python3
"""Quick check of the fused generate_XQT kernel against the per-stage launches (same arithmetic: must be bitwise equal)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zeronotesamba_amd import _lib
from zeronotesamba_amd.processing.input_rep import generate_XQT_batch

cases = [(160000, 3), (80000, 5), (33333, 4), (4097, 2), (131072, 2), (20000, 600), (8192, 1), (160000, 300), (12345, 7), (4096, 1), (1_600_000, 2)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
bad = 0
for n, clips in cases:
    g = torch.Generator().manual_seed(n + clips)
    y = (0.1 * torch.randn(clips, n, generator=g)).cuda()
    sep = generate_XQT_batch(y, 16000, "vqt", stages=3 | _lib.ZNS_XQT_SEPARATE)
    fus = generate_XQT_batch(y, 16000, "vqt", stages=3 | _lib.ZNS_XQT_FUSED)
    torch.cuda.synchronize()
    eq = torch.equal(sep, fus)
    d = (sep - fus).abs()
    nbad = int((d > 0).sum().item())
    msg = ""
    if not eq:
        bad += 1
        idx = torch.nonzero(d > 0)
        bins = sorted(set(idx[:, 1].tolist()))
        frames = sorted(set(idx[:, 2].tolist()))
        msg = f" maxdiff {d.max().item():.3e} nbad {nbad} bins {bins[:6]}..{bins[-3:]} frames {frames[:8]}..{frames[-5:]} clips {sorted(set(idx[:,0].tolist()))[:5]}"
        if not torch.isfinite(fus).all():
            msg += " NONFINITE"
    print(f"n={n} clips={clips}: {'OK bitwise' if eq else 'MISMATCH'}{msg}", flush=True)
sys.exit(1 if bad else 0)
