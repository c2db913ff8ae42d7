"""Pretext step time at other shapes than the headline one (e.g. the reference's 313-frame crops, pretext.py:285)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from zeronotesamba_amd.pretext import PretextTrainer

dev = torch.device("cuda", 0)
for B, T in [(16, 313), (16, 512), (16, 626), (8, 313), (32, 313)]:
    tr = PretextTrainer(bench.build_model(dev), batch_len=B, seed=1)
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = (torch.randn((B, 2, 96, T), generator=g) * 3 - 6).clamp_(-20.72, 3.0)
    x = tr.stack_inputs(x)
    for _ in range(3):
        tr.step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        tr.step(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(json.dumps({"B": B, "T": T, "ms_per_step": dt * 1e3, "us_per_frame_pair": dt * 1e6 / (B * T)}))
    del tr
    torch.cuda.empty_cache()
