"""Where does the anchor-branch cv8 gradient differ from the CPU oracle at full size?  Compares layer activations
(ReLU masks) and gradients of AUTO / DIRECT device paths against the oracle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import cnn_ref, step_ref
from tests.test_gpu_fullsize import build, run
from zeronotesamba_amd import _lib, _ops
from zeronotesamba_amd.pretext import PretextTrainer

B, T, tau = 16, 512, 0.25
torch.set_num_threads(16)
model = build(2)
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
g = torch.Generator().manual_seed(9)
batch = (torch.randn((B, 2, 96, T), generator=g) * 3.0 - 6.0).clamp_(-20.72, 3.0)
taps_a, taps_p = [], []
with torch.no_grad():
    cnn_ref.ds_cnn_forward(sd, "anchor.", batch[:, 0:1], taps=taps_a)
    cnn_ref.ds_cnn_forward(sd, "postve.", batch[:, 1:2], taps=taps_p)
x = batch.transpose(0, 1).reshape(2 * B, 1, 96, T).contiguous().to("cuda:0")
for impl, nm in ((_lib.ZNS_IMPL_AUTO, "auto"), (_lib.ZNS_IMPL_DIRECT, "direct")):
    e, acts, _ = run(impl, model, x, B, backward=False)
    for li in range(8):
        ref = torch.cat([taps_a[li], taps_p[li]])
        got = acts[li].cpu()
        d = (got - ref).abs()
        flips = ((got > 0) != (ref > 0))
        fa, fp = flips[:B].sum().item(), flips[B:].sum().item()
        big = (flips & ((got.abs() + ref.abs()) > 1e-4))
        print(f"{nm} layer {li + 1}: max|diff| {d.max().item():.2e} (scale {ref.abs().max().item():.2e})  "
              f"mask flips anchor {fa} postve {fp} of {flips[:B].numel()}  flips with |v|>1e-4: {big.sum().item()}")
