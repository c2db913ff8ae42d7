"""One optimisation step on the CPU oracle (TEST INFRASTRUCTURE; see oracle/__init__.py).

pretext_step  restates /root/reference/zeroNoteSamba/pretext.py:475-492
              (slice batch[:,0:1] / batch[:,1:2], zero_grad, forward, NT-Xent,
              backward, torch.optim.Adam.step with lr=1e-6, :202).
clmr_step     restates pretext.py:494-511 (pt_task "clmr": ONE shared DS_CNN applied to both
              views, Adam lr=1e-5, :205-208); sd keys are a lone DS_CNN's (`pretrained.cv1.weight` ...).
finetune_step restates /root/reference/zeroNoteSamba/epochs.py:48-63
              (Down_CNN forward, BCELoss (loader.py:16), backward, Adam).
adam_numpy    restates torch.optim.Adam (betas .9/.999, eps 1e-8, no weight decay,
              no amsgrad: the defaults pretext.py:202 relies on) in float64-free
              float32 numpy, to check the fused HIP Adam kernel element-wise.
"""
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import cnn_ref, ntxent_ref


class OracleParams:
    """Leaf tensors + a stock torch.optim.Adam over them, keyed like the reference state dict."""

    def __init__(self, sd: Dict[str, torch.Tensor], lr: float):
        self.sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        self.opt = torch.optim.Adam(list(self.sd.values()), lr=lr)

    def grads(self):
        return {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in self.sd.items()}

    def values(self):
        return {k: v.detach().clone() for k, v in self.sd.items()}


def pretext_step(op: OracleParams, batch: torch.Tensor, batch_len: int, tau: float,
                 drop_seed: Optional[int] = None, step_opt: bool = True, replay=None, replay_scale: float = 1.0,
                 flips=None):
    """batch: [B,2,96,T].  Returns (loss, pos, neg, anc_emb, pos_emb).

    replay / replay_scale / flips: mask-replay mode, see cnn_ref.cnn_forward -- the max-pool and ReLU/dropout
    decisions are taken from a record of another implementation's run (replay_scale = 1/(1-p) when that run
    had dropout on, the record's masks then already contain the keep decisions)."""
    anchors = batch[:, 0:1, :, :]
    postves = batch[:, 1:2, :, :]
    op.opt.zero_grad()
    a, p = cnn_ref.pretext_forward(op.sd, anchors, postves, drop_seed=drop_seed, replay=replay,
                                   replay_scale=replay_scale, flips=flips)
    loss, cpos, cneg = ntxent_ref.ntxent(a, p, batch_len, tau)
    loss.backward()
    if step_opt:
        op.opt.step()
    return float(loss.item()), cpos, cneg, a.detach(), p.detach()


def clmr_step(op: OracleParams, batch: torch.Tensor, batch_len: int, tau: float,
              drop_seed: Optional[int] = None, step_opt: bool = True, replay=None, replay_scale: float = 1.0,
              flips=None):
    """batch: [B,2,96,T].  anc_emb = model(anchors); pos_emb = model(postves) with the SAME weights (pretext.py:501-502).
    Returns (loss, pos, neg, anc_emb, pos_emb).  replay: {"anchor": {...}, "postve": {...}} decisions of the two passes."""
    op.opt.zero_grad()
    fa = fp = None
    if flips is not None:
        fa, fp = flips.setdefault("anchor", {}), flips.setdefault("postve", {})
    a = cnn_ref.ds_cnn_forward(op.sd, "", batch[:, 0:1, :, :], drop_seed=drop_seed, branch=0,
                               replay=None if replay is None else replay["anchor"], replay_scale=replay_scale, flips=fa)
    p = cnn_ref.ds_cnn_forward(op.sd, "", batch[:, 1:2, :, :], drop_seed=drop_seed, branch=0,
                               replay=None if replay is None else replay["postve"], replay_scale=replay_scale, flips=fp)
    loss, cpos, cneg = ntxent_ref.ntxent(a, p, batch_len, tau)
    loss.backward()
    if step_opt:
        op.opt.step()
    return float(loss.item()), cpos, cneg, a.detach(), p.detach()


def finetune_step(op: OracleParams, vqt: torch.Tensor, msk: torch.Tensor, reduction: str = "max",
                  drop_seed: Optional[int] = None, step_opt: bool = True, replay=None, replay_scale: float = 1.0,
                  flips=None):
    """vqt: [2,96,T], msk: [T] in {0,0.5,1} (ballroom.py:219-221).  sd keys prefixed 'pretext.'."""
    v1 = vqt[0].reshape(1, 1, vqt.shape[1], vqt.shape[2])
    v2 = vqt[1].reshape(1, 1, vqt.shape[1], vqt.shape[2])
    m = msk.reshape(1, -1)
    op.opt.zero_grad()
    out = cnn_ref.down_forward(op.sd, v1, v2, reduction=reduction, drop_seed=drop_seed, replay=replay,
                               replay_scale=replay_scale, flips=flips)
    loss = F.binary_cross_entropy(out, m)
    loss.backward()
    if step_opt:
        op.opt.step()
    return float(loss.item()), out.detach()


def adam_numpy(p, g, m, v, step: int, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    """One Adam update, float32 arithmetic in torch's (non-fused, non-capturable) op order."""
    p = p.astype(np.float32).copy()
    g = g.astype(np.float32)
    m = (m * np.float32(b1) + g * np.float32(1 - b1)).astype(np.float32)
    v = (v * np.float32(b2) + g * g * np.float32(1 - b2)).astype(np.float32)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    step_size = lr / bc1
    denom = (np.sqrt(v) / np.float32(np.sqrt(bc2)) + np.float32(eps)).astype(np.float32)
    p = (p - np.float32(step_size) * (m / denom)).astype(np.float32)
    return p, m, v
