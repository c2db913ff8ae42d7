"""Drop-in for /root/reference/zeroNoteSamba/models/models.py on MI355X.

Same classes, constructor arguments, attribute names and state-dict keys
(`{anchor,postve}.pretrained.cv{1..8}.{weight,bias}`, `{anchor,postve}.fc1.{weight,bias}`,
`pretext.` prefix under Down_CNN), so `load_state_dict` of a reference checkpoint, `.parameters()`
into any torch optimizer, `requires_grad=False` freezing (loader.py:30-34) and `torch.save` keep working.

What differs is where the arithmetic happens: `forward` hands device pointers to the HIP kernels of
libzns_hip.so (fp32-MFMA implicit-GEMM convs with fused bias/pool/ReLU/dropout, fused fc1+sigmoid head),
and the autograd.Function's backward calls the matching dgrad/wgrad kernels.  The submodules
(`nn.Conv2d` ...) only own the parameters; their own forward is never used.  CPU tensors raise.
"""
from typing import Any, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import _ops


def _fresh_seed() -> int:
    # torch's CPU generator, so torch.manual_seed() makes train-mode runs repeatable
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


def _apply_stack(x, x1, nbr, B, drop_p, seed, with_head, *params):
    """_StackFn.apply with the caller's grad mode passed in (autograd switches it off inside forward(), and
    ctx.needs_input_grad only reflects requires_grad): under torch.no_grad() -- val_epoch, inference -- nothing is saved
    for a backward pass."""
    return _StackFn.apply(x, x1, nbr, B, drop_p, seed, with_head, torch.is_grad_enabled(), *params)


class _StackFn(torch.autograd.Function):
    """nbr DS_CNN stacks (conv stack [+ head]) as one autograd node.

    forward(x [nbr*B,1,96,T], x1, nbr, B, drop_p, seed, with_head, grad_on, *params) -> emb [nbr*B,T] (or h [nbr*B,128,T])
    x1: None, or -- Pretext_CNN, whose caller hands two tensors (models.py:119-122) -- the positives [B,1,96,T] with x the
        anchors: the first layer's kernels take both pointers (no torch.cat), the node then RETURNS the two embeddings as two
        tensors (halves of one buffer) and its backward takes their two gradients as they come (no slice-backward fills, no
        cat): the path from the caller's tensors to the parameter gradients launches no ATen kernel.
    drop_p: dropout probability of this call (the module's `dp.p` in train mode, 0 in eval mode).
    grad_on: torch.is_grad_enabled() at the call site (see _apply_stack).
    """

    @staticmethod
    def forward(ctx, x, x1, nbr, B, drop_p, seed, with_head, grad_on, *params):
        P = _ops.StackParams.from_flat(params, nbr)
        pneeds = ctx.needs_input_grad[8:]
        need_grad = any(pneeds) and bool(grad_on)
        drop_p = float(drop_p)
        emb, acts, argm = _ops.stack_forward(x, P, B, drop_p, seed, save=need_grad, with_head=with_head, x1=x1)
        ctx.nbr, ctx.B, ctx.with_head = nbr, B, with_head
        ctx.pair = x1 is not None
        ctx.inv_keep = 1.0 / (1.0 - drop_p)
        ctx.conv_needs = any(n for i, n in enumerate(pneeds) if (i % 18) < 16)
        ctx.pneeds = pneeds
        if need_grad:
            ctx.n_acts = len(acts)
            ctx.argm_keys = sorted(argm.keys())
            ctx.save_for_backward(x, x1, emb if with_head else acts[-1], *acts, *[argm[k] for k in ctx.argm_keys],
                                  *params)
        if with_head:
            if ctx.pair:
                return emb[:B], emb[B:]
            return emb
        return acts[-1].reshape(acts[-1].shape[0], 128, -1)

    @staticmethod
    def backward(ctx, dout, dout1=None):
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            raise NotImplementedError("zeronotesamba_amd: gradient w.r.t. the input spectrogram is not provided "
                                      "(the reference never differentiates through the VQT)")
        saved = ctx.saved_tensors
        x, x1, out = saved[0], saved[1], saved[2]
        acts = list(saved[3:3 + ctx.n_acts])
        ams = saved[3 + ctx.n_acts:3 + ctx.n_acts + len(ctx.argm_keys)]
        params = saved[3 + ctx.n_acts + len(ctx.argm_keys):]
        argm = dict(zip(ctx.argm_keys, ams))
        P = _ops.StackParams.from_flat(params, ctx.nbr)
        grads = [torch.empty_like(p) for p in params]
        G = _ops.StackParams.from_flat(grads, ctx.nbr)
        dout = dout.contiguous()
        if ctx.with_head:
            _ops.stack_backward(x, P, ctx.B, acts, argm, out, dout, None, ctx.inv_keep, G,
                                conv_grads=ctx.conv_needs, x1=x1,
                                demb1=dout1.contiguous() if ctx.pair else None)
        else:
            # standalone _CNN: ReLU/dropout backward of cv8 on the incoming gradient (rare path, host glue)
            h = acts[7]
            dz = (dout.reshape(h.shape) * (h != 0).to(dout.dtype) * ctx.inv_keep).contiguous()
            _ops.stack_backward(x, P, ctx.B, acts, argm, None, None, dz, ctx.inv_keep, G,
                                conv_grads=ctx.conv_needs)
        out_grads: List[Optional[torch.Tensor]] = []
        for i, p in enumerate(params):
            is_conv = (i % 18) < 16
            if not ctx.pneeds[i] or (is_conv and not ctx.conv_needs) or (not is_conv and not ctx.with_head):
                out_grads.append(None)
            else:
                out_grads.append(grads[i])
        return (None, None, None, None, None, None, None, None, *out_grads)


def _prep_input(x: torch.Tensor, what: str) -> torch.Tensor:
    _ops.require_device(x, what)
    if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != _ops.N_BINS:
        raise RuntimeError(f"zeronotesamba_amd: {what} must be [B, 1, 96, T], got {tuple(x.shape)}")
    if x.shape[0] == 0 or x.shape[3] == 0:
        raise RuntimeError(f"zeronotesamba_amd: {what} has an empty batch or time axis: {tuple(x.shape)}")
    return x.contiguous()


class _CNN(nn.Module):
    """
    Convolutional layers (reference models.py:7-74).
    """

    def __init__(self) -> None:
        super(_CNN, self).__init__()

        self.cv1 = nn.Conv2d(in_channels=1, out_channels=64, kernel_size=(3, 11), padding=(1, 5))
        self.cv2 = nn.Conv2d(in_channels=64, out_channels=64, kernel_size=(7, 13), padding=(3, 6))
        self.cv3 = nn.Conv2d(in_channels=64, out_channels=128, kernel_size=(5, 15), padding=(2, 7))
        self.cv4 = nn.Conv2d(in_channels=128, out_channels=128, kernel_size=(9, 17), padding=(4, 8))
        self.cv5 = nn.Conv2d(in_channels=128, out_channels=256, kernel_size=(3, 19), padding=(1, 9))
        self.cv6 = nn.Conv2d(in_channels=256, out_channels=256, kernel_size=(5, 21), padding=(2, 10))
        self.cv7 = nn.Conv2d(in_channels=256, out_channels=128, kernel_size=(1, 23), padding=(0, 11))
        self.cv8 = nn.Conv2d(in_channels=128, out_channels=128, kernel_size=(1, 25), padding=(0, 12))

        self.relu = nn.ReLU(inplace=True)
        self.maxpl1 = nn.MaxPool2d((3, 1), padding=(0, 0))
        self.maxpl2 = nn.MaxPool2d((4, 1), padding=(0, 0))
        self.maxpl3 = nn.MaxPool2d((8, 1), padding=(0, 0))

        self.dp = nn.Dropout(p=0.1)

    def drop_p(self) -> float:
        """Dropout probability in effect: `self.dp.p` (models.py:30) while training, 0 in eval mode."""
        p = float(self.dp.p) if (self.training and self.dp.training) else 0.0
        if not 0.0 <= p < 1.0:
            raise ValueError(f"dropout probability has to be in [0, 1), but got {p}")
        return p

    def conv_params(self) -> List[torch.Tensor]:
        out: List[torch.Tensor] = []
        for name, *_ in _ops.LAYERS:
            cv = getattr(self, name)
            out += [cv.weight, cv.bias]
        return out

    def forward(self, x: torch.Tensor) -> Any:
        """
        Pass input through convolutional layers.
        -- x: input (vqt) [B, 1, 96, T] -> [B, 128, T]
        """
        x = _prep_input(x, "_CNN input")
        params = self.conv_params()
        dummy = [params[0].new_zeros(1), params[0].new_zeros(1)]  # head slots, unused when with_head=False
        drop_p = self.drop_p()
        seed = _fresh_seed() if drop_p > 0.0 else 0
        return _apply_stack(x, None, 1, x.shape[0], drop_p, seed, False, *params, *dummy)


class DS_CNN(nn.Module):
    """
    Fully-convolutional architecture for beat tracking (reference models.py:77-103).
    """

    def __init__(self) -> None:
        super(DS_CNN, self).__init__()

        self.pretrained = _CNN()

        # Output
        self.fc1 = nn.Conv1d(in_channels=128, out_channels=1, kernel_size=1, padding=0)
        self.sig = nn.Sigmoid()

    def stack_params(self) -> List[torch.Tensor]:
        return self.pretrained.conv_params() + [self.fc1.weight, self.fc1.bias]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """
        -- x: input (vqt) [B, 1, 96, T] -> [B, T]
        """
        x = _prep_input(x, "DS_CNN input")
        drop_p = self.pretrained.drop_p()
        seed = _fresh_seed() if drop_p > 0.0 else 0
        return _apply_stack(x, None, 1, x.shape[0], drop_p, seed, True, *self.stack_params())


class Pretext_CNN(nn.Module):
    """
    DS_CNN tailored for percussive and non-percussive stems (reference models.py:106-124).
    Both branches run in ONE launch per layer (branch is a grid dimension).
    """

    def __init__(self) -> None:
        super(Pretext_CNN, self).__init__()

        self.anchor = DS_CNN()
        self.postve = DS_CNN()

    def forward(self, anc: torch.Tensor, pos: torch.Tensor) -> Tuple[Any, Any]:
        """
        Pass vqts through each model.
        """
        anc = _prep_input(anc, "anchor input")
        pos = _prep_input(pos, "positive input")
        same_dev = anc.device == pos.device == self.anchor.fc1.weight.device == self.postve.fc1.weight.device
        drop_p = self.anchor.pretrained.drop_p()
        if not same_dev or anc.shape != pos.shape or drop_p != self.postve.pretrained.drop_p():
            # the reference's branch-per-device placement (pretext.py:199-200): run the branches separately, each on
            # the device its input and weights live on (every C-ABI call switches to its tensors' device, _ops.on_device)
            return self.anchor(anc), self.postve(pos)
        B = anc.shape[0]
        seed = _fresh_seed() if drop_p > 0.0 else 0
        return _apply_stack(anc, pos, 2, B, drop_p, seed, True, *self.anchor.stack_params(),
                            *self.postve.stack_params())


class _DownFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, p, reduction):
        from .. import _lib
        a = a.contiguous()
        p = p.contiguous()
        out = torch.empty_like(a)
        _ops.same_device(a, (p,), "Down_CNN combine")
        with _ops.on_device(a):
            rc = _lib.lib().zns_down_fwd_f32(a.data_ptr(), p.data_ptr(), a.numel(), reduction, out.data_ptr(),
                                             _ops.stream_ptr(a.device))
        _lib.check(rc, "zns_down_fwd_f32")
        ctx.reduction = reduction
        ctx.save_for_backward(a, p)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .. import _lib
        a, p = ctx.saved_tensors
        dout = dout.contiguous()
        da, dp = torch.empty_like(a), torch.empty_like(p)
        with _ops.on_device(a):
            rc = _lib.lib().zns_down_bwd_f32(a.data_ptr(), p.data_ptr(), dout.data_ptr(), a.numel(), ctx.reduction,
                                             da.data_ptr(), dp.data_ptr(), _ops.stream_ptr(a.device))
        _lib.check(rc, "zns_down_bwd_f32")
        return da, dp, None


class Down_CNN(nn.Module):
    """
    Use of Pretext_CNN for downstream tasks (reference models.py:127-150).
    """

    def __init__(self, reduction: str = "max") -> None:
        super(Down_CNN, self).__init__()

        self.pretext = Pretext_CNN()
        self.reduction = reduction

    def forward(self, anc: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
        """
        Pass each input through each model. Add and output.
        """
        from .. import _lib
        anc_emb, pos_emb = self.pretext(anc, pos)
        if anc_emb.device != pos_emb.device:      # branch-per-device placement: combine where the positives live
            anc_emb = anc_emb.to(pos_emb.device)
        red = _lib.ZNS_REDUCE_MEAN if self.reduction == "mean" else _lib.ZNS_REDUCE_MAX
        return _DownFn.apply(anc_emb, pos_emb, red)
