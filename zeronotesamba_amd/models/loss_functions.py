"""Drop-in for /root/reference/zeroNoteSamba/models/loss_functions.py (NTXent) on MI355X.

Same constructor and return triple `(loss tensor, mean anchor/positive cosine, mean anchor/negative
cosine)`.  The reference loops over rows in Python with three `.item()` syncs per row
(loss_functions.py:35-49); here one fused kernel computes the cosine matrix, the row-wise
log-softmax, the two statistics AND the gradient w.r.t. both embeddings (zns_ntxent_f32), and the
two floats are read back with a single device->host copy.
"""
from typing import Tuple

import torch
import torch.nn as nn

from .. import _lib, _ops


class _NTXentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchors, poss, batch_len, temperature, out3=None):
        """out3: where the (loss, pos cos, neg cos) triple goes (a row of an epoch's statistics table), or None."""
        a = anchors.contiguous()
        p = poss.contiguous()
        rows, D = a.shape
        if out3 is None:
            out3 = torch.empty(3, dtype=torch.float32, device=a.device)
        loss = torch.empty((), dtype=torch.float32, device=a.device)     # written by the kernel beside out3[0]: no copy
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        ga = torch.empty_like(a) if need else None
        gp = torch.empty_like(p) if need else None
        _ops.ntxent(a, p, rows, D, batch_len, temperature, out3, ga, gp, loss_out=loss)
        if need:
            ctx.save_for_backward(ga, gp)
        ctx.mark_non_differentiable(out3)
        ctx.set_materialize_grads(False)      # no zero-filled "gradient" of the statistics triple handed to backward
        return loss, out3

    @staticmethod
    def backward(ctx, dloss, _dstats):
        if dloss is None:
            return None, None, None, None, None
        ga, gp = ctx.saved_tensors
        da, dp = _ops.scale_by(dloss, ga, gp)      # d(loss) may be any device scalar: applied on the device, one launch
        return da, dp, None, None, None


class NTXent(nn.Module):
    """
    Compute NT-Xent loss for contrastive learning.
    """

    def __init__(self, batch_len: int, temperature: float = 0.25):
        """
        Arguments:
        -- batch_len: batch size
        -- temperature: parameter
        """
        super(NTXent, self).__init__()

        self.batch_len = batch_len
        self.temperature = temperature
        self.CS = nn.CosineSimilarity(dim=1, eps=1e-08)  # kept for attribute parity; not used by the kernel

    def forward(self, anchors: torch.Tensor, poss: torch.Tensor) -> Tuple[torch.Tensor, float, float]:
        """
        Arguments:
        -- anchors: tensor of shape (batch_len, embedding_size)
        -- poss: tensor of shape (batch_len, embedding_size)
        """
        _ops.require_device(anchors, "NTXent anchors")
        _ops.require_device(poss, "NTXent positives")
        if anchors.dim() != 2 or anchors.shape != poss.shape:
            raise RuntimeError(f"NTXent: expected two [batch, D] tensors, got {tuple(anchors.shape)} and "
                               f"{tuple(poss.shape)}")
        if anchors.shape[0] > self.batch_len:
            # the reference would index past its torch.zeros(batch_len) buffer (loss_functions.py:30,49)
            raise IndexError(f"NTXent: {anchors.shape[0]} rows > batch_len={self.batch_len}")
        loss, stats = _NTXentFn.apply(anchors, poss, self.batch_len, self.temperature)
        s = stats.tolist()  # one D2H copy for both floats (the reference does 3 .item() per row)
        return loss, s[1], s[2]
