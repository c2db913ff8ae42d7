"""Diagnostic: do naive / MFMA forward passes pick different max-pool rows than the fp32 CPU oracle,
and are those windows near-ties?  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle import cnn_ref
from zeronotesamba_amd import _lib, _ops
from zeronotesamba_amd.models.models import Pretext_CNN

torch.set_num_threads(os.cpu_count())
B, T = 4, 64
sd = cnn_ref.synth_state_dict(5, 2.0)
batch = cnn_ref.synth_vqt(200, (B, 2, 96, T))
x = torch.cat([batch[:, 0:1], batch[:, 1:2]])

# CPU: raw conv outputs per pooled layer, per branch
def cpu_raw(dtype):
    out = {}
    for br, pre in enumerate(("anchor.", "postve.")):
        cur = x[br*B:(br+1)*B].to(dtype)
        for li, (name, cin, cout, kh, kw, pool) in enumerate(cnn_ref.LAYERS):
            z = F.conv2d(cur, sd[f"{pre}pretrained.{name}.weight"].to(dtype), sd[f"{pre}pretrained.{name}.bias"].to(dtype), padding=(kh//2, kw//2))
            if pool > 1:
                out[(br, li)] = z
                z = F.max_pool2d(z, (pool, 1))
            cur = F.relu(z)
    return out
raw32 = cpu_raw(torch.float32)
raw64 = cpu_raw(torch.float64)

m = Pretext_CNN(); m.load_state_dict(sd); m.to("cuda:0").eval()
P = _ops.StackParams.from_flat(m.anchor.stack_params() + m.postve.stack_params(), 2)
for impl, nm in ((1, "naive"), (0, "mfma")):
    _lib.lib().zns_set_impl(impl)
    with torch.no_grad():
        emb, acts, argm = _ops.stack_forward(x.to("cuda:0"), P, B, 0.0, 0, save=True)
    _lib.lib().zns_set_impl(0)
    for li in (1, 3, 5):
        pool = cnn_ref.LAYERS[li][5]
        am = argm[li].cpu().long()
        for br in range(2):
            for tag, raw in (("cpu32", raw32), ("cpu64", raw64)):
                z = raw[(br, li)]
                n, c, h, t = z.shape
                zc = z.reshape(n, c, h // pool, pool, t)
                ref_am = zc.argmax(dim=3)
                mine = am[br*B:(br+1)*B]
                diff = (ref_am != mine)
                top2 = zc.topk(2, dim=3).values
                gap = (top2[:, :, :, 0] - top2[:, :, :, 1]).abs()
                active = zc.max(dim=3).values > 0
                nd = int((diff & active).sum())
                gaps = gap[diff & active]
                print(f"{nm:5s} layer {li} br {br} vs {tag}: {nd} active flips of {int(active.sum())} active windows; "
                      f"max gap at flips {gaps.max().item() if nd else 0:.2e}; windows with gap<1e-5: {int(((gap<1e-5)&active).sum())}")
