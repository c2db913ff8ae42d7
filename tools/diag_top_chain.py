#!/usr/bin/env python3
"""Which kernel of the top of the backward chain loses accuracy?  Feeds the device NT-Xent kernels and head_bwd with
the embeddings / cv8 activations of a dump written by tools/diag_dump_step.py (argv[1]; copy it into the repo tree first:
gpurun_out/ does not travel) and compares each with float64 autograd on the same inputs.  Round 5 found the NT-Xent tables
this way: float32 scalars of the gradient chain err coherently along an embedding (csrc/zns_small.hip, ntx_dots_kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import ntxent_ref
from test_gpu_fullsize import build
from zeronotesamba_amd import _ops
r = torch.load(sys.argv[1])
dev = torch.device("cuda:0")
B, tau = 16, 0.25
E = r["emb"]
T = E.shape[1]
e = E.double().clone().requires_grad_(True)
loss, _, _ = ntxent_ref.ntxent(e[:B], e[B:], B, tau)
loss.backward()
d64 = e.grad
a, p = E[:B].to(dev).contiguous(), E[B:].to(dev).contiguous()
out3 = torch.empty(3, device=dev)
ga, gp = torch.empty_like(a), torch.empty_like(p)
_ops.ntxent(a, p, B, T, B, tau, out3, ga, gp)
torch.cuda.synchronize()
dd = torch.cat([ga, gp]).cpu().double()
print("NT-Xent gradient kernel vs float64 autograd on the same embeddings: rel %.2e  (loss %.7f / %.7f)" %
      ((dd - d64).norm() / d64.norm(), out3[0].item(), loss.item()))
# head_bwd with the float64 demb rounded to float32 as input
model = build(2)
P = _ops.StackParams.from_flat(model.anchor.stack_params() + model.postve.stack_params(), 2)
h = r["act8"].to(dev)
emb = E.to(dev)
for label, demb in (("exact demb", d64.float().to(dev)), ("device demb", dd.float().to(dev))):
    dw = [torch.zeros_like(t) for t in P.fc_w]
    db = [torch.zeros_like(t) for t in P.fc_b]
    ws = _ops.workspace(_ops.conv_workspace_bytes(2, B, T), dev, 0)
    dz = _ops.head_bwd(2, B, T, h, emb, demb.contiguous(), P.fc_w, 1.0, dw, db, False, ws)
    torch.cuda.synchronize()
    ed = E.double()
    dpre = d64 * ed * (1 - ed)
    hh = r["act8"].double().reshape(2 * B, 128, -1)
    for br in range(2):
        sl = slice(br * B, (br + 1) * B)
        gw = torch.einsum("nt,nct->c", dpre[sl], hh[sl])
        print(f"  head_bwd ({label}) branch {br}: fc1.weight grad rel %.2e   dumped step's: %.2e" %
              ((dw[br].cpu().double().reshape(-1) - gw).norm() / gw.norm(),
               (r["grads"][("anchor", "postve")[br] + ".fc1.weight"].double().reshape(-1) - gw).norm() / gw.norm()))
