#!/usr/bin/env python3
"""The full-size pretext step (B = 16 pairs) three ways with the device's decisions replayed: the device, the float32 CPU
oracle (numerically the reference) and the same oracle in float64.  Prints, per parameter gradient, the relative L2 distance
device-oracle32 (what tests/test_gpu_fullsize.py bounds by 1e-4), device-float64 and oracle32-float64: how much of the first
is the reference's OWN float32 rounding of an ill-conditioned sum (NT-Xent on near-collinear embeddings).
    python tools/diag_fullsize_f64.py [T] [win128: -1 / 0 / 1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import step_ref
from replay_util import device_decisions, rel_l2
from test_gpu_fullsize import build
from zeronotesamba_amd import _lib
from zeronotesamba_amd.pretext import PretextTrainer

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
win = int(sys.argv[2]) if len(sys.argv) > 2 else 0
_lib.lib().zns_test_set_tuning(_lib.ZNS_TUNE_WIN128, win)
B, tau = 16, 0.25
torch.set_num_threads(len(os.sched_getaffinity(0)))
model = build(2)
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
g = torch.Generator().manual_seed(9)
batch = (torch.randn((B, 2, 96, T), generator=g) * 3.0 - 6.0).clamp_(-20.72, 3.0)
tr = PretextTrainer(model, batch_len=B, temperature=tau, lr=1e-6, train=False)
x = tr.stack_inputs(batch)
out3 = tr.step(x, update=False).tolist()
replay, emb = device_decisions(x, tr.P, B)
op32 = step_ref.OracleParams(sd, 1e-6)
l32, _, _, a32, p32 = step_ref.pretext_step(op32, batch, B, tau, drop_seed=None, step_opt=False, replay=replay)
op64 = step_ref.OracleParams({k: v.double() for k, v in sd.items()}, 1e-6)
l64, _, _, a64, p64 = step_ref.pretext_step(op64, batch.double(), B, tau, drop_seed=None, step_opt=False, replay=replay)
g32, g64 = op32.grads(), op64.grads()
e = emb.cpu().double()
r32, r64 = torch.cat([a32, p32]).double(), torch.cat([a64, p64])
print(f"T={T} win128={win}: loss device {out3[0]:.7f} oracle32 {l32:.7f} float64 {l64:.9f}")
print("embeddings max|diff|: device-oracle32 %.2e  device-float64 %.2e  oracle32-float64 %.2e" %
      ((e - r32).abs().max(), (e - r64).abs().max(), (r32 - r64).abs().max()))
w = [0, 0, 0]
for name, p in model.named_parameters():
    d = [rel_l2(p.grad, g32[name]), rel_l2(p.grad, g64[name]), rel_l2(g32[name], g64[name])]
    w = [max(a, b) for a, b in zip(w, d)]
    print(f"  {name:34s} dev-o32 {d[0]:.2e}  dev-f64 {d[1]:.2e}  o32-f64 {d[2]:.2e}")
print("worst: dev-o32 %.2e  dev-f64 %.2e  o32-f64 %.2e" % tuple(w))
