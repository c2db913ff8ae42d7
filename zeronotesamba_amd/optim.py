"""torch.optim.Adam stand-in whose update runs in the fused HIP kernels zns_adam_multi_f32 / zns_adam_step_f32.

Same defaults as the optimizer the reference builds (pretext.py:202 `Adam(params, lr=1e-6)`,
loader.py:43 `Adam(params, lr=..., betas=(0.9, 0.999))`): eps 1e-8, no weight decay, no amsgrad.
One launch per parameter GROUP and device (a device-resident table of the tensors' pointers, rebuilt only when a
parameter, its gradient buffer or its state was replaced); PretextTrainer (pretext.py) uses the flat single-buffer form.
"""
from typing import Iterable, Tuple

import torch

from . import _lib, _ops


class Adam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.0, amsgrad: bool = False):
        if weight_decay != 0.0 or amsgrad:
            raise ValueError("zeronotesamba_amd.optim.Adam: weight_decay/amsgrad are not used by the reference "
                             "and are not implemented")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._tables = {}      # (group index, device, step) bucket -> (pointer signature, device table)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            buckets = {}       # (device, step after this update) -> [(p, g, exp_avg, exp_avg_sq)]
            for p in group["params"]:
                if p.grad is None:
                    continue
                _ops.require_device(p, "Adam parameter")
                if not p.is_contiguous():
                    raise RuntimeError("zeronotesamba_amd.optim.Adam: parameters must be contiguous")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                _ops.same_device(p, (p.grad,), "Adam")
                buckets.setdefault((p.device, st["step"]), []).append((p, p.grad, st["exp_avg"], st["exp_avg_sq"]))
            for bi, ((dev, step), items) in enumerate(buckets.items()):
                sig = tuple(t.data_ptr() for it in items for t in it)
                key = (gi, dev, bi)
                hit = self._tables.get(key)
                if hit is None or hit[0] != sig:
                    rows = [[p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()] for p, g, m, v in items]
                    hit = (sig, torch.tensor(rows, dtype=torch.int64).to(dev))
                    self._tables[key] = hit
                with _ops.on_device(items[0][0]):
                    rc = L.zns_adam_multi_f32(hit[1].data_ptr(), len(items), max(it[0].numel() for it in items),
                                              group["lr"], b1, b2, group["eps"], step, 1.0, _ops.stream_ptr(dev))
                _lib.check(rc, "zns_adam_multi_f32")
                for p, _, _, _ in items:
                    _ops.bump_version(p)       # written through the raw pointer: invalidate cached packed operands
        return loss
