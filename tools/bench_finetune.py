"""Fine-tune step timing (BASELINE config 4: batch of one 30 s clip, T=1876, Down_CNN + BCE + Adam)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from zeronotesamba_amd.epochs import bce_loss
from zeronotesamba_amd.models.models import Down_CNN
from zeronotesamba_amd.optim import Adam

dev = torch.device("cuda", 0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1876
torch.manual_seed(0)
m = Down_CNN().to(dev)
opt = Adam(m.parameters(), lr=5e-7)
g = torch.Generator().manual_seed(1)
x = (torch.randn((2, 1, 1, 96, T), generator=g) * 3 - 6).clamp_(-20.72, 3.0).to(dev)
msk = torch.zeros((1, T), device=dev)
msk[0, ::31] = 1.0


def step():
    opt.zero_grad()
    out = m(x[0], x[1])
    loss = bce_loss(out, msk)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(json.dumps({"T": T, "ms_per_step": dt * 1e3, "steps_per_s": 1 / dt, "loss": float(l)}))
