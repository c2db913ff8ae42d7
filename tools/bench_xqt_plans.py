#!/usr/bin/env python3
"""generate_XQT, marching kernel against the per-stage launches, over batch sizes (10 s clips resident in HBM)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zeronotesamba_amd import _lib
from zeronotesamba_amd.processing.input_rep import generate_XQT_batch


def t(y, out, stages, iters):
    for _ in range(2):
        generate_XQT_batch(y, 16000, "vqt", out=out, stages=stages)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        generate_XQT_batch(y, 16000, "vqt", out=out, stages=stages)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [64, 128, 192, 256, 384, 512, 1024, 2048, 4096]
    g = torch.Generator().manual_seed(1234)
    base = (0.1 * torch.randn(64, 160000, generator=g)).cuda()
    for n in sizes:
        y = base.repeat((n + 63) // 64, 1)[:n].contiguous()
        out = torch.empty((n, 96, 626), dtype=torch.float32, device="cuda")
        it = max(3, min(50, 4096 // n))
        sep = t(y, out, 3 | _lib.ZNS_XQT_SEPARATE, it)
        mar = t(y, out, 3 | _lib.ZNS_XQT_MARCH, it)
        print(json.dumps({"n_clips": n, "separate_ms": round(sep, 4), "march_ms": round(mar, 4), "march_vs_separate": round(mar / sep, 3),
                          "march_clips_per_s": round(n / mar * 1e3)}), flush=True)
