"""Diagnostic: per-parameter gradient error of the HIP paths (naive, mfma) and of the fp32 CPU oracle,
all measured against a float64 CPU evaluation of the same step.  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cnn_ref, ntxent_ref
from zeronotesamba_amd import _lib
from zeronotesamba_amd.models.models import Pretext_CNN
from zeronotesamba_amd.models.loss_functions import NTXent

torch.set_num_threads(os.cpu_count())
B, T, tau = 4, 64, 0.25
sd = cnn_ref.synth_state_dict(5, 2.0)
batch = cnn_ref.synth_vqt(200, (B, 2, 96, T))

def cpu_grads(dtype):
    p = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    a, q = cnn_ref.pretext_forward(p, batch[:, 0:1].to(dtype), batch[:, 1:2].to(dtype))
    loss, _ = ntxent_ref.ntxent_forward(a, q, B, tau)
    loss.backward()
    return {k: v.grad.double() for k, v in p.items()}, a.detach().double()

g64, a64 = cpu_grads(torch.float64)
g32, a32 = cpu_grads(torch.float32)

def gpu_grads(impl):
    _lib.lib().zns_set_impl(impl)
    m = Pretext_CNN(); m.load_state_dict(sd); m.to("cuda:0").eval()
    a, q = m(batch[:, 0:1].to("cuda:0"), batch[:, 1:2].to("cuda:0"))
    loss, _, _ = NTXent(B, tau)(a, q)
    loss.backward()
    _lib.lib().zns_set_impl(0)
    return {k: v.grad.cpu().double() for k, v in m.named_parameters()}, a.detach().cpu().double()

gn, an = gpu_grads(1)
gm, am = gpu_grads(0)
print("emb err vs f64: cpu32 %.2e naive %.2e mfma %.2e" % ((a32-a64).abs().max(), (an-a64).abs().max(), (am-a64).abs().max()))
print("%-34s %10s %10s %10s %10s" % ("param", "cpu32", "naive", "mfma", "mfma-naive"))
for k in g64:
    n = g64[k].norm().item() + 1e-30
    print("%-34s %10.2e %10.2e %10.2e %10.2e" % (k, (g32[k]-g64[k]).norm().item()/n, (gn[k]-g64[k]).norm().item()/n,
          (gm[k]-g64[k]).norm().item()/n, (gm[k]-gn[k]).norm().item()/n))
