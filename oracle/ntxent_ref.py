"""NT-Xent restatement (TEST INFRASTRUCTURE; see oracle/__init__.py).

Follows /root/reference/zeroNoteSamba/models/loss_functions.py:24-55:
  for each anchor row i:  s_ij = cos(a_i, p_j)  (nn.CosineSimilarity(dim=1, eps=1e-8), :22)
                          L_i  = -log( exp(s_ii/tau) / sum_j exp(s_ij/tau) )        (:38-49)
  returns (sum_i L_i / batch_len, sum_i s_ii / batch_len,
           sum_i ((sum_j s_ij - s_ii)/(batch_len-1)) / batch_len)                    (:51-55)
Note the divisions by ``batch_len`` (constructor argument), not by the number of
rows actually present (:30 allocates torch.zeros(batch_len)).

``ntxent_numpy`` is an independent float64 closed form with analytic gradients,
used to check the HIP backward kernel without autograd.
"""
from typing import Tuple

import numpy as np
import torch

EPS = 1e-8


def cosine_matrix(a: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """C[i,j] = cos(a_i, p_j) with torch>=1.12 semantics: x/max(|x|,eps) . y/max(|y|,eps)."""
    an = a / a.norm(dim=1, keepdim=True).clamp_min(EPS)
    pn = p / p.norm(dim=1, keepdim=True).clamp_min(EPS)
    return an @ pn.t()


def ntxent_forward(a: torch.Tensor, p: torch.Tensor, batch_len: int, tau: float):
    """Differentiable torch version; returns (loss, C)."""
    c = cosine_matrix(a, p)
    s = c / tau
    li = torch.logsumexp(s, dim=1) - torch.diagonal(s)
    return li.sum() / batch_len, c


def ntxent(a: torch.Tensor, p: torch.Tensor, batch_len: int, tau: float) -> Tuple[torch.Tensor, float, float]:
    loss, c = ntxent_forward(a, p, batch_len, tau)
    cd = c.detach().double()
    pos = float(torch.diagonal(cd).sum() / batch_len)
    neg = float(((cd.sum(dim=1) - torch.diagonal(cd)) / (batch_len - 1)).sum() / batch_len)
    return loss, pos, neg


def ntxent_numpy(a: np.ndarray, p: np.ndarray, batch_len: int, tau: float):
    """float64 closed form + analytic gradients (dL/da, dL/dp)."""
    a = a.astype(np.float64)
    p = p.astype(np.float64)
    na = np.maximum(np.linalg.norm(a, axis=1, keepdims=True), EPS)
    npn = np.maximum(np.linalg.norm(p, axis=1, keepdims=True), EPS)
    ah, ph = a / na, p / npn
    c = ah @ ph.T
    s = c / tau
    m = s.max(axis=1, keepdims=True)
    e = np.exp(s - m)
    z = e.sum(axis=1, keepdims=True)
    li = (m + np.log(z))[:, 0] - np.diag(s)
    loss = li.sum() / batch_len
    n = a.shape[0]
    # dL/dC
    g = (e / z - np.eye(n)) / (tau * batch_len)
    # C = ah ph^T ; d ah_i/d a_i = (I - ah ah^T)/|a|
    gah = g @ ph
    gph = g.T @ ah
    ga = (gah - ah * (gah * ah).sum(axis=1, keepdims=True)) / na
    gp = (gph - ph * (gph * ph).sum(axis=1, keepdims=True)) / npn
    pos = np.diag(c).sum() / batch_len
    neg = ((c.sum(axis=1) - np.diag(c)) / (batch_len - 1)).sum() / batch_len
    return loss, pos, neg, ga, gp
