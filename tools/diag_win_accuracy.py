#!/usr/bin/env python3
"""How far each kernel family is from a float64 convolution, cv3 / cv4 at T = 512 (forward, backward-data, backward-weights):
64-point windows, 128-point windows, the direct MFMA kernels and torch's float32 CPU conv2d (what the oracle runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from zeronotesamba_amd import _lib, _ops

dev = torch.device("cuda:0")
L = _lib.lib()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.set_num_threads(16)
for li in (2, 3):
    name, cin, cout, kh, kw, pool = _ops.LAYERS[li]
    H = _ops.H_IN[li]
    g = torch.Generator().manual_seed(li)
    nbr, B = 1, 2
    x = (torch.randn((B, cin, H, T), generator=g) * 0.5 + 0.3).clamp_(min=0)      # post-ReLU-like: non-negative, many zeros
    w = torch.randn((cout, cin, kh, kw), generator=g) / (cin * kh * kw) ** 0.5
    b = torch.zeros(cout)
    dz = torch.randn((B, cout, H, T), generator=g)
    z64 = F.conv2d(x.double(), w.double(), None, padding=(kh // 2, kw // 2))
    z32 = F.conv2d(x, w, None, padding=(kh // 2, kw // 2))
    dx64 = torch.nn.grad.conv2d_input(x.shape, w.double(), dz.double(), padding=(kh // 2, kw // 2))
    dx32 = torch.nn.grad.conv2d_input(x.shape, w, dz, padding=(kh // 2, kw // 2))
    dw64 = torch.nn.grad.conv2d_weight(x.double(), w.shape, dz.double(), padding=(kh // 2, kw // 2))
    dw32 = torch.nn.grad.conv2d_weight(x, w.shape, dz, padding=(kh // 2, kw // 2))

    def err(got, ref):
        d = (got.double().cpu() - ref)
        return "rms %.2e max %.2e (of rms %.2e)" % ((d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(),
                                                     (d.abs().max() / ref.pow(2).mean().sqrt()).item(), ref.pow(2).mean().sqrt().item())
    print(f"{name} T={T}: torch f32 CPU  fwd {err(z32, z64)} | dgrad {err(dx32, dx64)} | wgrad {err(dw32, dw64)}")
    xd, wd, bd, dzd = x.to(dev), [w.to(dev)], [b.to(dev)], dz.to(dev)
    for label, impl, win in (("64-point", _lib.ZNS_IMPL_AUTO, -1), ("128-point", _lib.ZNS_IMPL_AUTO, 1), ("direct", _lib.ZNS_IMPL_DIRECT, 0)):
        L.zns_test_set_tuning(_lib.ZNS_TUNE_WIN128, win)
        prev = L.zns_set_impl(impl)
        try:
            ws = _ops.workspace(_ops.conv_workspace_bytes(nbr, B, T), dev, 0)
            # forward without pooling / ReLU is not exposed: compare the pooled, ReLU'd output through the same ops in float64
            y, am, spec = _ops.conv_fwd(li, nbr, B, T, xd, wd, bd, 0.0, 0, True, ws)
            yp = F.max_pool2d(z64, (pool, 1)) if pool > 1 else z64
            yref = F.relu(yp)
            mask = (yref > 1e-3)                           # away from the ReLU edge
            dxg = _ops.conv_dgrad(li, nbr, B, T, dzd, wd, None, 1.0, None, 1, ws)
            dw = [torch.zeros_like(wd[0])]
            db = [torch.zeros_like(bd[0])]
            _ops.conv_wgrad(li, nbr, B, T, dzd, xd, dw, db, False, ws)
            torch.cuda.synchronize()
            e = ((y.double().cpu() - yref)[mask])
            print(f"   {label:10s} fwd rms {(e.pow(2).mean().sqrt() / yref[mask].pow(2).mean().sqrt()).item():.2e} max {(e.abs().max() / yref[mask].pow(2).mean().sqrt()).item():.2e}"
                  f" | dgrad {err(dxg, dx64)} | wgrad {err(dw[0], dw64)}")
        finally:
            L.zns_set_impl(prev)
            L.zns_test_set_tuning(_lib.ZNS_TUNE_WIN128, 0)
