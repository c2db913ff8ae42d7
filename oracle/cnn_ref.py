"""CPU restatement of the reference CNNs (TEST INFRASTRUCTURE; see oracle/__init__.py).

Follows /root/reference/zeroNoteSamba/models/models.py:
  _CNN.forward      :32-74   8 x (conv2d -> [maxpool over H] -> relu -> dropout), squeeze(dim=2)
  DS_CNN.forward    :91-103  _CNN -> Conv1d(128->1,k=1) -> sigmoid -> reshape(B, T)
  Pretext_CNN       :117-124 two independent DS_CNN (anchor, postve)
  Down_CNN          :138-150 max (default) or mean of the two embeddings

Everything is stock ``torch.nn.functional`` on CPU tensors, driven by a plain
state dict with the reference's keys (``anchor.pretrained.cv1.weight`` ...), so
it needs no reference code at run time.  Dropout takes an explicit keep mask
(``oracle.dropout_ref``) instead of torch's RNG.
"""
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import dropout_ref

# (name, cin, cout, kh, kw, pool)  -- models.py:16-28 ; padding is (kh//2, kw//2) for every layer
LAYERS: List[Tuple[str, int, int, int, int, int]] = [
    ("cv1", 1, 64, 3, 11, 1),
    ("cv2", 64, 64, 7, 13, 3),
    ("cv3", 64, 128, 5, 15, 1),
    ("cv4", 128, 128, 9, 17, 4),
    ("cv5", 128, 256, 3, 19, 1),
    ("cv6", 256, 256, 5, 21, 8),
    ("cv7", 256, 128, 1, 23, 1),
    ("cv8", 128, 128, 1, 25, 1),
]
DROP_P = 0.1  # models.py:30


def param_shapes(prefix: str = "") -> Dict[str, Tuple[int, ...]]:
    """State-dict manifest of one DS_CNN (keys/shapes as the reference module reports them)."""
    out: Dict[str, Tuple[int, ...]] = {}
    for name, cin, cout, kh, kw, _ in LAYERS:
        out[f"{prefix}pretrained.{name}.weight"] = (cout, cin, kh, kw)
        out[f"{prefix}pretrained.{name}.bias"] = (cout,)
    out[f"{prefix}fc1.weight"] = (1, 128, 1)
    out[f"{prefix}fc1.bias"] = (1,)
    return out


def pretext_param_shapes() -> Dict[str, Tuple[int, ...]]:
    out = param_shapes("anchor.")
    out.update(param_shapes("postve."))
    return out


def synth_state_dict(seed: int, gain: float = 2.0) -> Dict[str, torch.Tensor]:
    """Closed-form, exactly reproducible Pretext_CNN state dict.

    U(-b, b) with b = gain/sqrt(fan_in) (gain>1 keeps the sigmoid outputs from
    collapsing to a constant, SURVEY.md section 7 'degenerate default init').
    """
    sd: Dict[str, torch.Tensor] = {}
    for k, (name, shape) in enumerate(pretext_param_shapes().items()):
        n = int(np.prod(shape))
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        u = dropout_ref.uniform_pm1(seed * 1000 + k, n)
        if name.endswith("bias"):
            b = 0.05
        else:
            b = gain / np.sqrt(fan_in)
        sd[name] = torch.from_numpy((u * np.float32(b)).reshape(shape).copy())
    return sd


def synth_vqt(seed: int, shape) -> torch.Tensor:
    """VQT-like input: smooth-ish values in [-20.72, 3] (log-magnitudes)."""
    n = int(np.prod(shape))
    u = dropout_ref.uniform_pm1(seed, n).reshape(shape)
    v = dropout_ref.uniform_pm1(seed + 7919, n).reshape(shape)
    x = -6.0 + 4.0 * u + 3.0 * u * v
    return torch.from_numpy(np.clip(x, -20.7233, 3.0).astype(np.float32))


def cnn_forward(
    sd: Dict[str, torch.Tensor],
    prefix: str,
    x: torch.Tensor,
    drop_seed: Optional[int] = None,
    branch: int = 0,
    n_branch_batch_offset: int = 0,
    keep: Optional[List[torch.Tensor]] = None,
    taps: Optional[list] = None,
    replay: Optional[Dict[int, Tuple[torch.Tensor, Optional[torch.Tensor]]]] = None,
    replay_scale: float = 1.0,
    flips: Optional[Dict[int, Tuple[int, int, int]]] = None,
) -> torch.Tensor:
    """_CNN.forward (models.py:32-74).  x: [B,1,96,T] -> [B,128,T].

    drop_seed=None -> eval mode (dropout is the identity, models.py:30 + nn.Module.eval()).
    Otherwise each layer's keep mask is dropout_ref.keep_mask(layer_seed(seed, layer, branch), ...)
    over the layer OUTPUT tensor [B,C,Hp,T] (contiguous linear index).

    replay (mask-replay mode): {layer index: (nonzero mask of the layer output [B,C,Hp,T] bool,
    arg-max row 0..pool-1 per pooled output uint8 or None)} recorded from ANOTHER implementation's run.
    The discontinuous decisions of the stack -- which row a max-pool window picks, which elements
    ReLU/dropout zero -- are then taken from the record instead of being re-derived from this run's
    values: out = gather(conv, argmax) * mask * replay_scale.  Everything continuous (conv2d and the
    gradients autograd derives from it) is still stock torch.  Two fp32 implementations resolve
    near-ties (gap below summation noise) differently; replaying the decisions removes exactly that
    difference, so per-parameter gradients can be compared at 1e-4 instead of a flip-tolerant 1e-2.
    flips[li] = (pool windows whose own arg-max differs, mask elements whose own ReLU decision
    differs, elements) counts how many decisions were actually overridden.
    """
    out = x
    for li, (name, cin, cout, kh, kw, pool) in enumerate(LAYERS):
        w = sd[f"{prefix}pretrained.{name}.weight"]
        b = sd[f"{prefix}pretrained.{name}.bias"]
        out = F.conv2d(out, w, b, padding=(kh // 2, kw // 2))
        if replay is not None:
            mask, am = replay[li]
            n_am = 0
            if pool > 1:
                Bn, C, Hh, Tt = out.shape
                z5 = out.reshape(Bn, C, Hh // pool, pool, Tt)
                own = z5.detach().argmax(dim=3)   # first maximum wins, as max_pool2d
                n_am = int((own != am.long()).sum())
                out = torch.gather(z5, 3, am.long().unsqueeze(3)).squeeze(3)
            # with dropout in the record only one direction is visible: kept by the record, zeroed by this run's ReLU
            n_mask = int(((out.detach() > 0) != mask).sum()) if replay_scale == 1.0 else int((mask & (out.detach() <= 0)).sum())
            if flips is not None:
                flips[li] = (n_am, n_mask, mask.numel())
            out = out * mask.to(out.dtype) * replay_scale
            if taps is not None:
                taps.append(out)
            continue
        if pool > 1:
            out = F.max_pool2d(out, (pool, 1))
        out = F.relu(out)
        if keep is not None:
            out = out * keep[li] * (1.0 / (1.0 - DROP_P))
        elif drop_seed is not None:
            m = dropout_ref.keep_mask(dropout_ref.layer_seed(drop_seed, li, branch), tuple(out.shape), DROP_P)
            out = out * torch.from_numpy(m.astype(np.float32)) * np.float32(1.0 / (1.0 - DROP_P))
        if taps is not None:
            taps.append(out)
    return torch.squeeze(out, dim=2)


def ds_cnn_forward(sd, prefix, x, drop_seed=None, branch=0, taps=None, replay=None, replay_scale=1.0,
                   flips=None) -> torch.Tensor:
    """DS_CNN.forward (models.py:91-103). -> [B, T]"""
    h = cnn_forward(sd, prefix, x, drop_seed=drop_seed, branch=branch, taps=taps, replay=replay,
                    replay_scale=replay_scale, flips=flips)
    y = F.conv1d(h, sd[f"{prefix}fc1.weight"], sd[f"{prefix}fc1.bias"])
    y = torch.sigmoid(y)
    return y.reshape(y.size(0), y.size(1) * y.size(2))


def pretext_forward(sd, anc, pos, drop_seed=None, prefix="", replay=None, replay_scale=1.0, flips=None):
    """Pretext_CNN.forward (models.py:117-124).  replay: {"anchor": {...}, "postve": {...}} (see cnn_forward)."""
    fa = fp = None
    if flips is not None:
        fa, fp = flips.setdefault("anchor", {}), flips.setdefault("postve", {})
    a = ds_cnn_forward(sd, f"{prefix}anchor.", anc, drop_seed=drop_seed, branch=0,
                       replay=None if replay is None else replay["anchor"], replay_scale=replay_scale, flips=fa)
    p = ds_cnn_forward(sd, f"{prefix}postve.", pos, drop_seed=drop_seed, branch=1,
                       replay=None if replay is None else replay["postve"], replay_scale=replay_scale, flips=fp)
    return a, p


def down_forward(sd, anc, pos, reduction="max", drop_seed=None, replay=None, replay_scale=1.0, flips=None):
    """Down_CNN.forward (models.py:138-150); sd keys carry the 'pretext.' prefix."""
    a, p = pretext_forward(sd, anc, pos, drop_seed=drop_seed, prefix="pretext.", replay=replay,
                           replay_scale=replay_scale, flips=flips)
    if reduction == "mean":
        return torch.div(a + p, 2)
    return torch.maximum(a, p)
