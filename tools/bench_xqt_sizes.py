#!/usr/bin/env python3
"""generate_XQT: fused kernel vs per-stage launches over batch sizes (10 s clips resident in HBM); ms per call."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zeronotesamba_amd import _lib
from zeronotesamba_amd.processing.input_rep import generate_XQT_batch

def timed(y, res, stages, iters):
    for _ in range(3):
        generate_XQT_batch(y, 16000, "vqt", out=res, stages=stages)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        generate_XQT_batch(y, 16000, "vqt", out=res, stages=stages)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

for n in (1, 8, 32, 64, 128, 256, 512, 1024, 4096):
    g = torch.Generator().manual_seed(n)
    y = (0.1 * torch.randn(n, 160000, generator=g)).cuda()
    res = torch.empty((n, 96, 626), dtype=torch.float32, device="cuda")
    it = 50 if n <= 256 else 10
    row = {"n_clips": n, "fused_ms": timed(y, res, 3 | _lib.ZNS_XQT_FUSED, it), "separate_ms": timed(y, res, 3 | _lib.ZNS_XQT_SEPARATE, it),
           "exact_ms": timed(y, res, 3 | _lib.ZNS_XQT_EXACT, it)}
    row["fused_frac_of_8TBps"] = n * (160000 * 4 + 96 * 626 * 4) / (row["fused_ms"] * 1e-3) / 8e12
    print(json.dumps(row), flush=True)
