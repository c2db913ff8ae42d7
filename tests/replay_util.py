"""Mask replay between the device and the CPU oracle (test helper).

Two fp32 implementations of the conv stack agree on every activation to ~1e-6 of its scale, but a max-pool
window whose top two rows are closer than that, or a ReLU input within that distance of zero, can be decided
differently -- and ONE flipped decision reroutes a whole gradient column (measured: up to 1e-2 of a weight
gradient's L2 norm).  `device_decisions` records the decisions the HIP kernels took
(arg-max maps and the nonzero pattern of every layer output, which encodes ReLU AND dropout); the oracle
(oracle.cnn_ref.cnn_forward(replay=...)) then takes those decisions instead of its own, so that everything
continuous can be compared at the north_star tolerance of 1e-4.
"""
from typing import Dict, List

import torch


def device_decisions(x: torch.Tensor, P, B: int, drop_p: float = 0.0, seed: int = 0, names=("anchor", "postve"),
                     stack_B=None):
    """x: [2B,1,96,T] (anchor rows, then positive rows) on the device, P: _ops.StackParams.  Returns (replay dict
    keyed by view name for oracle.cnn_ref.pretext_forward / step_ref.clmr_step, embeddings [2B, T] on the device).
    stack_B: rows per weight set of the launch (B for the twin Pretext_CNN launch, 2B for the shared clmr DS_CNN)."""
    from zeronotesamba_amd import _ops
    with torch.no_grad():
        emb, acts, argm = _ops.stack_forward(x, P, stack_B or B, drop_p, seed, save=True)
    torch.cuda.synchronize()
    out: Dict[str, Dict[int, tuple]] = {}
    for br, name in enumerate(names):
        rec = {}
        for li in range(8):
            mask = (acts[li][br * B:(br + 1) * B] != 0).cpu()
            am = argm[li][br * B:(br + 1) * B].cpu() if li in argm else None
            rec[li] = (mask, am)
        out[name] = rec
    return out, emb


def rel_l2(got: torch.Tensor, ref: torch.Tensor) -> float:
    got, ref = got.detach().cpu().double(), ref.detach().double()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


def grad_err(name: str, got: torch.Tensor, refs) -> float:
    """rel_l2 of one parameter gradient against refs[name].  The single-element fc1.bias gradient is a sum of signed
    terms (sum of d_emb * emb * (1 - emb)) that can cancel to far below its terms (near-collinear embeddings: 1e-6 from
    terms of 1e-3), so its error is measured on the scale of the same sum weighted by the activations -- the largest
    fc1.weight gradient entry -- whenever that is larger than the bias gradient itself."""
    ref = refs[name].detach().double()
    err = float((got.detach().cpu().double() - ref).norm())
    scale = float(ref.norm())
    if name.endswith("fc1.bias"):
        scale = max(scale, float(refs[name[:-len("bias")] + "weight"].detach().abs().max()))
    return err / max(scale, 1e-30)


def flip_summary(flips) -> str:
    """'anchor: cv2 3 pool / 1 relu of 1.0e+06; ...' for the layers where a decision was overridden."""
    parts: List[str] = []
    for brname, per in flips.items():
        for li, (n_am, n_mask, n) in sorted(per.items()):
            if n_am or n_mask > 0:
                parts.append(f"{brname}.cv{li + 1}: {n_am} pool + {n_mask} relu of {n:.1e}")
    return "; ".join(parts) if parts else "no decision differs"


def total_flips(flips) -> int:
    return sum(n_am + max(n_mask, 0) for per in flips.values() for (n_am, n_mask, _n) in per.values())
