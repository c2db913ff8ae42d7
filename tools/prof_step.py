#!/usr/bin/env python3
"""The fused pretext step at a given crop length (default: the reference's own 313 frames, pretext.py:285), for
rocprofv3:  rocprofv3 --kernel-trace --stats ... -- python3 tools/prof_step.py --frames 313 --steps 4"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from zeronotesamba_amd.pretext import PretextTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=313)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--win128", type=int, default=0, help="cv3 / cv4 window length: 0 library rule, 1 force 128-point, -1 force 64-point")
ap.add_argument("--ipw", type=int, default=0, help="ZNS_TUNE_IPW (experiments)")
ap.add_argument("--ksplit", type=int, default=-1, help="cv7 / cv8 input-channel split: -1 rule, 0 off, 2 / 4 / 8")
a = ap.parse_args()
if a.ipw:
    from zeronotesamba_amd import _lib
    _lib.check(_lib.lib().zns_test_set_tuning(_lib.ZNS_TUNE_IPW, a.ipw), "tuning")
if a.ksplit != -1:
    from zeronotesamba_amd import _lib
    _lib.check(_lib.lib().zns_test_set_tuning(_lib.ZNS_TUNE_KSPLIT, a.ksplit), "tuning")
if a.win128:
    from zeronotesamba_amd import _lib
    _lib.check(_lib.lib().zns_test_set_tuning(_lib.ZNS_TUNE_WIN128, a.win128), "tuning")
dev = torch.device("cuda", 0)
tr = PretextTrainer(bench.build_model(dev), batch_len=a.batch, seed=1)
g = torch.Generator().manual_seed(a.batch * 1000 + a.frames)
x = tr.stack_inputs((torch.randn((a.batch, 2, 96, a.frames), generator=g) * 3 - 6).clamp_(-20.72, 3.0))
for _ in range(a.warmup):
    tr.step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    tr.step(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"B": a.batch, "T": a.frames, "ms_per_step": dt * 1e3, "us_per_frame_pair": dt * 1e6 / (a.batch * a.frames),
                  "executed_mfma_tflop_per_step": bench.exec_flop(a.frames, 2 * a.batch) / 1e12,
                  "executed_frac_of_fp32_mfma_peak": bench.exec_flop(a.frames, 2 * a.batch) / dt / (bench.FP32_MFMA_PEAK_TFLOPS * 1e12)}))
