#!/usr/bin/env python3
"""Which ATen operators (and from which Python line) does one module-path step call?  python tools/diag_aten_ops.py [finetune]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from zeronotesamba_amd.epochs import bce_loss
from zeronotesamba_amd.models.loss_functions import NTXent
from zeronotesamba_amd.models.models import Down_CNN, Pretext_CNN
from zeronotesamba_amd.optim import Adam
DEV = "cuda:0"
ft = len(sys.argv) > 1 and sys.argv[1] == "finetune"
B, T = (1, 96) if ft else (4, 64)
if len(sys.argv) > 2:
    T = int(sys.argv[2])
g = torch.Generator().manual_seed(3)
both = (torch.randn((2, B, 1, 96, T), generator=g) * 3 - 6).clamp_(-20.72, 3.0).to(DEV)
m = (Down_CNN() if ft else Pretext_CNN()).to(DEV)
crit = NTXent(B, 0.25)
msk = torch.zeros((1, T), device=DEV)
opt = Adam(m.parameters(), lr=1e-6)
def step():
    opt.zero_grad()
    if ft:
        loss = bce_loss(m(both[0], both[1]), msk)
    else:
        a, p = m(both[0], both[1])
        loss, _, _ = crit(a, p)
    loss.backward()
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
for e in prof.events():
    if e.name.startswith("aten::") and any(k in e.name for k in ("fill", "zero", "ones", "copy", "mul", "add", "cat", "clone", "contiguous", "to")):
        st = [s for s in (e.stack or []) if "zeronotesamba" in s or "diag_aten" in s or "torch/" in s][:4]
        print(e.name, "|", getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0)), "|", st)
