#!/usr/bin/env python3
"""CPU half of the float64 diagnosis: the dumps of tools/diag_dump_step.py against the float32 oracle and the same oracle in
float64 (forward only, own decisions: a flipped near-tie moves an activation by its rounding error).  Per dumped layer: rms /
max distance to float64, device and oracle32; then the NT-Xent gradient with respect to the embeddings evaluated in float64 at
the three sets of embeddings: its sensitivity to embedding errors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import cnn_ref, ntxent_ref
from test_gpu_fullsize import build

torch.set_num_threads(len(os.sched_getaffinity(0)))
files = sys.argv[1:]
recs = [torch.load(f) for f in files]
T = recs[0]["T"]
B, tau = 16, 0.25
import zeronotesamba_amd.models.models as mm
torch.manual_seed(2)
m = mm.Pretext_CNN()
with torch.no_grad():
    for name, p in m.named_parameters():
        if name.endswith("weight"):
            p.mul_(2.0)
sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
g = torch.Generator().manual_seed(9)
batch = (torch.randn((B, 2, 96, T), generator=g) * 3.0 - 6.0).clamp_(-20.72, 3.0)
out = {}
for dt, nm in ((torch.float32, "o32"), (torch.float64, "f64")):
    s = {k: v.to(dt) for k, v in sd.items()}
    ta, tp = [], []
    with torch.no_grad():
        a = cnn_ref.ds_cnn_forward(s, "anchor.", batch[:, 0:1].to(dt), taps=ta)
        p = cnn_ref.ds_cnn_forward(s, "postve.", batch[:, 1:2].to(dt), taps=tp)
    out[nm] = {"emb": torch.cat([a, p]), "act7": torch.cat([ta[6], tp[6]]), "act8": torch.cat([ta[7], tp[7]]),
               "act6_s0": ta[5][0], "act6_s16": tp[5][0], "act4_s0": ta[3][0], "act3_s0": ta[2][0, :32]}
    del ta, tp
    print("oracle", nm, "done", flush=True)
f64 = out["f64"]

def dist(x, ref):
    d = (x.double() - ref.double())
    r = ref.double().pow(2).mean().sqrt()
    return "rms %.2e max %.2e" % ((d.pow(2).mean().sqrt() / r).item(), (d.abs().max() / r).item())
for key in ("act3_s0", "act4_s0", "act6_s0", "act6_s16", "act7", "act8", "emb"):
    line = f"{key:9s} o32: {dist(out['o32'][key], f64[key])}"
    for r in recs:
        line += f" | dev(win {r['win']:2d}, ks {r['ks']:2d}): {dist(r[key], f64[key])}"
    print(line)

def ntx_grad(emb):
    e = emb.double().clone().requires_grad_(True)
    loss, _, _ = ntxent_ref.ntxent(e[:B], e[B:], B, tau)
    loss.backward()
    return e.grad
g64 = ntx_grad(f64["emb"])
print("NT-Xent d loss / d emb in float64 at each set of embeddings, relative L2 distance to the one at the float64 embeddings:")
print("  oracle32 embeddings: %.2e" % ((ntx_grad(out['o32']['emb']) - g64).norm() / g64.norm()).item())
for r in recs:
    print("  device (win %d, ks %d): %.2e" % (r['win'], r['ks'], ((ntx_grad(r['emb']) - g64).norm() / g64.norm()).item()))
