"""torch.optim.Adam stand-in whose update runs in the fused HIP kernel zns_adam_step_f32.

Same defaults as the optimizer the reference builds (pretext.py:202 `Adam(params, lr=1e-6)`,
loader.py:43 `Adam(params, lr=..., betas=(0.9, 0.999))`): eps 1e-8, no weight decay, no amsgrad.
One launch per parameter tensor; PretextTrainer (pretext.py) uses the flat single-launch form.
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

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                _ops.require_device(p, "Adam parameter")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.is_contiguous():
                    raise RuntimeError("zeronotesamba_amd.optim.Adam: parameters must be contiguous")
                _ops.same_device(p, (g,), "Adam")
                with _ops.on_device(p):
                    rc = L.zns_adam_step_f32(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                             st["exp_avg_sq"].data_ptr(), p.numel(), group["lr"], b1, b2,
                                             group["eps"], st["step"], 1.0, _ops.stream_ptr(p.device))
                _lib.check(rc, "zns_adam_step_f32")
                _ops.bump_version(p)       # written through the raw pointer: invalidate cached packed operands
        return loss
