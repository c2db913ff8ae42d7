"""The model/loss/step part of /root/reference/zeroNoteSamba/epochs.py on MI355X.

`train_epoch` / `val_epoch` keep the reference's structure for the "pretrained" branch
(epochs.py:48-63, 143-153): per file, batch of one, Down_CNN forward -> BCELoss -> backward -> Adam.
The beat-picking evaluation the reference runs afterwards (madmom DBN + mir_eval, epochs.py:83-91)
is CPU post-processing outside this package's scope: pass `evaluate=callable` to plug it back in.
"""
import contextlib
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch

from . import _lib, _ops


class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, target):
        o = out.contiguous()
        t = target.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=o.device)
        need = ctx.needs_input_grad[0]
        dout = torch.empty_like(o) if need else None
        _ops.same_device(o, (t,), "BCELoss")
        with _ops.on_device(o):
            rc = _lib.lib().zns_bce_f32(o.data_ptr(), t.data_ptr(), o.numel(), loss.data_ptr(),
                                        dout.data_ptr() if need else None, 1.0, _ops.stream_ptr(o.device))
        _lib.check(rc, "zns_bce_f32")
        if need:
            ctx.save_for_backward(dout)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dout,) = ctx.saved_tensors
        return _ops.scale_by(dloss, dout), None


def bce_loss(output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """torch.nn.BCELoss() (mean reduction, loader.py:16) as one fused forward+backward kernel."""
    _ops.require_device(output, "BCE output")
    _ops.require_device(target, "BCE target")
    if output.shape != target.shape:
        raise ValueError(f"Using a target size ({tuple(target.shape)}) that is different to the input size "
                         f"({tuple(output.shape)}) is deprecated. Please ensure they have the same size.")
    return _BCEFn.apply(output, target)


class BCELoss(torch.nn.Module):
    """Stand-in for `torch.nn.BCELoss().cuda()` of loader.py:16."""

    def forward(self, output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return bce_loss(output, target)


def _to_dev(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32)


def train_epoch(model: torch.nn.Module, criterion: Callable, optimizer: torch.optim.Optimizer, _status: str,
                indices: List[str], real_times: Dict[str, Any], inputs: Dict[str, Any], masks: Dict[str, Any],
                threshold: bool = False, librosa: bool = False,
                evaluate: Optional[Callable] = None) -> Tuple[Any, ...]:
    """epochs.py:8-101.  Returns (model, optimizer, mean loss, f_measure, cmlc, cmlt, amlc, amlt, info_gain);
    the six metrics are 0.0 unless `evaluate(cpu_output, times, threshold=..., librosa=...)` is supplied."""
    device = next(model.parameters()).device
    model.train()
    # frozen backbone (loader.py:29-34): the optimizer only ever touches fc1, whose weights no kernel packs -- the conv
    # operands (forward and backward-data) are then built once for the whole epoch
    convs = [p for n, p in model.named_parameters() if ".pretrained." in n or n.startswith("pretrained.")]
    scope = _ops.frozen_weights() if convs and not any(p.requires_grad for p in convs) else contextlib.nullcontext()
    with scope:
        return _train_loop(model, criterion, optimizer, _status, indices, real_times, inputs, masks, threshold, librosa,
                           evaluate, device)


def _train_loop(model, criterion, optimizer, _status, indices, real_times, inputs, masks, threshold, librosa, evaluate,
                device):
    full_loss = 0.0
    metrics = [0.0] * 6
    batch_idx = -1
    for batch_idx, wav in enumerate(indices):
        times = real_times[wav]
        if _status == "pretrained":
            vqt = inputs[wav]
            vqt1 = _to_dev(torch.reshape(vqt[0, :, :], (1, 1, vqt.shape[1], vqt.shape[2])), device)
            vqt2 = _to_dev(torch.reshape(vqt[1, :, :], (1, 1, vqt.shape[1], vqt.shape[2])), device)
            msk = masks[wav]
            msk = _to_dev(torch.reshape(msk, (1, msk.shape[0])), device)
            optimizer.zero_grad()
            output = model(vqt1, vqt2)
        else:
            vqt = inputs[wav]
            vqt = _to_dev(torch.reshape(vqt[:, :], (1, 1, vqt.shape[0], vqt.shape[1])), device)
            msk = masks[wav]
            msk = _to_dev(torch.reshape(msk, (1, msk.shape[0])), device)
            optimizer.zero_grad()
            output = model(vqt)
        loss = criterion(output, msk)
        loss.backward()
        optimizer.step()
        full_loss += loss.item()
        if evaluate is not None:
            res = evaluate(output.squeeze(0).cpu().detach().numpy(), times, threshold=threshold, librosa=librosa)
            for i in range(6):
                metrics[i] += res[i]
    n = batch_idx + 1
    return (model, optimizer, full_loss / n, *[m / n for m in metrics])


def val_epoch(model: torch.nn.Module, criterion: Callable, _status: str, indices: List[str],
              real_times: Dict[str, Any], inputs: Dict[str, Any], masks: Dict[str, Any],
              threshold: bool = False, librosa: bool = False,
              evaluate: Optional[Callable] = None) -> Tuple[Any, ...]:
    """epochs.py:104-187 (model.eval(), no_grad).  Returns (mean loss, f_measure, cmlc, cmlt, amlc, amlt, info_gain)."""
    device = next(model.parameters()).device
    full_loss = 0.0
    metrics = [0.0] * 6
    model.eval()
    batch_idx = -1
    with torch.no_grad(), _ops.frozen_weights():     # no weight changes inside this loop: operands packed once per epoch
        for batch_idx, wav in enumerate(indices):
            times = real_times[wav]
            if _status == "pretrained":
                vqt = inputs[wav]
                vqt1 = _to_dev(torch.reshape(vqt[0, :, :], (1, 1, vqt.shape[1], vqt.shape[2])), device)
                vqt2 = _to_dev(torch.reshape(vqt[1, :, :], (1, 1, vqt.shape[1], vqt.shape[2])), device)
                output = model(vqt1, vqt2)
            else:
                vqt = inputs[wav]
                vqt = _to_dev(torch.reshape(vqt[:, :], (1, 1, vqt.shape[0], vqt.shape[1])), device)
                output = model(vqt)
            msk = masks[wav]
            msk = _to_dev(torch.reshape(msk, (1, msk.shape[0])), device)
            loss = criterion(output, msk)
            full_loss += loss.item()
            if evaluate is not None:
                res = evaluate(output.squeeze(0).cpu().numpy(), times, threshold=threshold, librosa=librosa)
                for i in range(6):
                    metrics[i] += res[i]
    n = batch_idx + 1
    return (full_loss / n, *[m / n for m in metrics])
