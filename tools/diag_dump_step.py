#!/usr/bin/env python3
"""GPU half of the float64 diagnosis (tools/diag_compare_f64.py is the CPU half): the full-size step of
tests/test_gpu_fullsize.py::test_full_size_step_against_the_cpu_oracle on the device; dumps embeddings, some layers'
activations and the top layers' gradients to gpurun_out/diag_step_T<T>_w<win>.pt.
    python tools/diag_dump_step.py [T] [win128: -1 / 0 / 1] [ksplit: -1 / 0 / 8]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from replay_util import device_decisions
from test_gpu_fullsize import build
from zeronotesamba_amd import _lib, _ops
from zeronotesamba_amd.pretext import PretextTrainer

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
win = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ks = int(sys.argv[3]) if len(sys.argv) > 3 else -1
_lib.lib().zns_test_set_tuning(_lib.ZNS_TUNE_WIN128, win)
_lib.lib().zns_test_set_tuning(_lib.ZNS_TUNE_KSPLIT, ks)
B, tau = 16, 0.25
model = build(2)
g = torch.Generator().manual_seed(9)
batch = (torch.randn((B, 2, 96, T), generator=g) * 3.0 - 6.0).clamp_(-20.72, 3.0)
tr = PretextTrainer(model, batch_len=B, temperature=tau, lr=1e-6, train=False)
x = tr.stack_inputs(batch)
out3 = tr.step(x, update=False).tolist()
replay, emb = device_decisions(x, tr.P, B)
with torch.no_grad():
    _, acts, _ = _ops.stack_forward(x, tr.P, B, 0.0, 0, save=True)
keep = ("fc1", "cv8", "cv7")
rec = {"T": T, "win": win, "ks": ks, "out3": out3, "emb": emb.cpu(), "act7": acts[6].cpu(), "act8": acts[7].cpu(),
       "act6_s0": acts[5][0].cpu(), "act6_s16": acts[5][16].cpu(), "act4_s0": acts[3][0].cpu(), "act3_s0": acts[2][0, :32].cpu(),
       "grads": {n: p.grad.detach().cpu() for n, p in model.named_parameters() if any(k in n for k in keep)}}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
torch.save(rec, os.path.join(ROOT, "gpurun_out", f"diag_step_T{T}_w{win}_k{ks}.pt"))
print("saved", T, win, ks, out3)
