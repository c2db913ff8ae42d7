#!/usr/bin/env python3
"""Where fill_memory_bank's host time goes (cProfile over the bench.py workload: 256 files x 4 stems resident in HBM)."""
import cProfile, os, pstats, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zeronotesamba_amd.pretext import fill_memory_bank
dev = torch.device("cuda", 0)
n_files, n = 256, 12 * 16000
g = torch.Generator().manual_seed(0)
drums = (0.1 * torch.randn((n_files, n), generator=g)).to(dev)
t = torch.arange(n, device=dev) % (5 * 16000)
gain = torch.where(t < 2 * 16000, 0.6, 0.1).float()
bass = (drums[0] * gain * 0.5).contiguous()
stems = {f"f{i}": {"bass": bass, "drums": drums[i], "other": bass, "vocals": 0.01 * drums[(i + 1) % n_files]} for i in range(n_files)}
fps = list(stems)
yml = {"clip_len": 10, "input_mode": "vqt", "lower_p": 0.1, "upper_p": 1.0}
fill_memory_bank(stems, fps[:64], yml, rng=random.Random(1))
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    fill_memory_bank(stems, fps, yml, rng=random.Random(2))
    torch.cuda.synchronize()
    print("fill_memory_bank: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
fill_memory_bank(stems, fps, yml, rng=random.Random(2))
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
