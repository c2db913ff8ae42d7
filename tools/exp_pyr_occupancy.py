#!/usr/bin/env python3
"""Pyramid alone (per-level launches) at 4096 clips; ZNS_XQT_LDS_PAD (KB) caps the chain kernel's occupancy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zeronotesamba_amd.processing.input_rep import generate_XQT_batch
n = 2048
y = (0.1 * torch.randn(64, 160000)).cuda().repeat(n // 64, 1).contiguous()
out = torch.empty((n, 96, 626), dtype=torch.float32, device="cuda")
for _ in range(2):
    generate_XQT_batch(y, 16000, "vqt", out=out, stages=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    generate_XQT_batch(y, 16000, "vqt", out=out, stages=1)
torch.cuda.synchronize()
print("LDS_PAD", os.environ.get("ZNS_XQT_LDS_PAD"), "pyramid ms per 2048 clips", (time.perf_counter() - t0) / 5 * 1e3)
