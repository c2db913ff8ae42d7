"""Host glue between torch tensors (device memory + streams only) and the C ABI (include/zns.h).

Nothing here computes: every function checks shapes, allocates outputs with torch's caching
allocator, and enqueues HIP kernels on torch's current stream through ctypes.
"""
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib

# (name, cin, cout, kh, kw, pool) -- /root/reference/zeroNoteSamba/models/models.py:16-28
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
H_IN = [96, 96, 32, 32, 8, 8, 1, 1]   # rows seen by each conv (n_bins = 96, models.py pools 3, 4, 8)
DROP_P = 0.1                           # models.py:30
N_BINS = 96

_workspaces: Dict[Tuple[int, int, int], torch.Tensor] = {}


def require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"zeronotesamba_amd: {what} is on '{t.device}'. This package has no CPU path: move the module and "
            "its inputs to a ROCm device (.to('cuda')).")
    if t.dtype != torch.float32:
        raise RuntimeError(f"zeronotesamba_amd: {what} must be float32, got {t.dtype}")


def stream_ptr(device=None) -> int:
    """hipStream_t of torch's current stream ON `device` (default: the current device)."""
    return torch.cuda.current_stream(device).cuda_stream


class on_device:
    """Make `t`'s device the current HIP device for the enclosed C-ABI calls (kernels launch on the CURRENT device; the
    stream, the workspace and every pointer must belong to it).  Switches only when needed; restores on exit."""

    __slots__ = ("idx", "prev")

    def __init__(self, t: torch.Tensor):
        self.idx = t.device.index if t.device.index is not None else torch.cuda.current_device()

    def __enter__(self):
        self.prev = torch.cuda.current_device()
        if self.prev != self.idx:
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev != self.idx:
            torch.cuda.set_device(self.prev)
        return False


def same_device(ref: torch.Tensor, tensors, what: str) -> None:
    """Every tensor of one call must live on ref's device: the library never copies between GPUs."""
    for t in tensors:
        if t is not None and t.device != ref.device:
            raise RuntimeError(f"zeronotesamba_amd: {what}: tensors on different devices ({ref.device} and {t.device}); "
                               "move the module and its inputs to ONE device (or call the branches separately)")


def workspace(nbytes: int, device: torch.device, slot: int = 0) -> torch.Tensor:
    """Grow-only scratch buffer per (device, CURRENT STREAM of that device, slot): kernels of one stream run in order and
    may share scratch, two streams never do (a forward on a side stream gets its own buffer instead of silently
    overwriting the default stream's split-K slabs and spectra).  slot separates users inside one stream."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream_ptr(device), slot)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def release_workspaces(device=None) -> None:
    """Drop the scratch buffers (of `device`, or of every device): they are grow-only and one set exists per stream that
    ever launched a kernel, so a process that cycles through many streams or shrinks its batch size can give the memory
    back.  Safe at any time: a buffer still referenced by enqueued kernels stays alive in torch's stream-ordered allocator."""
    idx = None if device is None else torch.device(device).index
    for key in [k for k in _workspaces if idx is None or k[0] == idx]:
        del _workspaces[key]


def conv_workspace_bytes(nbr: int, B: int, T: int) -> int:
    """Largest per-layer workspace of one stack pass (split-K factors follow the CURRENT device's CU count)."""
    L = _lib.lib()
    need = 0
    for (name, cin, cout, kh, kw, pool), H in zip(LAYERS, H_IN):
        need = max(need, L.zns_conv2d_workspace_bytes(nbr, B, T, cin, cout, kh, kw, H))
    return max(need, (nbr * B * 129 + 64) * 4)


def _pp(tensors: Sequence[torch.Tensor]):
    return _lib.ptr_array([t.data_ptr() for t in tensors])


class _Frozen:
    """Process-wide on purpose: autograd runs `backward` on its own device thread, which must see the scope the caller
    opened on the Python thread."""
    depth = 0
    touched: Optional[list] = None
    lock = threading.RLock()


_frozen = _Frozen()


class frozen_weights:
    """`with frozen_weights():` -- the caller's promise that no weight is written inside the block other than by something
    that bumps the tensors' autograd version counters (any torch optimizer or in-place op on the Parameter; this package's
    fused Adam kernels bump them explicitly).  Inside it the kernel-order weight operands (zns_conv2d_pack_f32: 0.5 GB per
    pass over cv2..cv6 at B = 16) are built once per weight version and reused from call to call; OUTSIDE it every call
    packs into its workspace and nothing is cached, so a write the counters cannot see (`p.data.mul_()`, EMA on `.data`,
    a raw-pointer write) can never be served stale weights -- the round-2 cache inferred "not training" from the grad mode
    and had exactly that blind spot.  The operands live as long as the OUTERMOST block: leaving it drops them (and their
    memory).  val_epoch (pretext.py / epochs.py), PretextTrainer.evaluate_epoch and a frozen-backbone train_epoch hold it
    around their own loops; an inference loop (sample_script.py:46-48 over many files) opts in explicitly.
    Process-wide (autograd's backward thread sees it); re-entrant."""

    def __enter__(self):
        with _frozen.lock:
            if _frozen.depth == 0:
                _frozen.touched = []
            _frozen.depth += 1
        return self

    def __exit__(self, *exc):
        with _frozen.lock:
            _frozen.depth -= 1
            if _frozen.depth == 0:
                for p in _frozen.touched or ():
                    p.__dict__.pop("_zns_pack", None)
                _frozen.touched = None
        return False


def weights_frozen() -> bool:
    return _frozen.depth > 0


def packed_weights(li: int, transposed: bool, nbr: int, B: int, T: int, w: Sequence[torch.Tensor]) -> Optional[torch.Tensor]:
    """The kernel-order operand of layer li's weights, or None (the C side then packs into the call's workspace).

    Only inside a `frozen_weights()` block is an operand cached: on the first branch's weight Parameter, keyed by the
    library's layout id (layer, tile variant, kernel family), the launch stream and the tensors' addresses, validated by their autograd
    version counters (a second guard: an optimizer step inside the block still rebuilds), dropped when the outermost
    block exits."""
    if not weights_frozen():
        return None
    name, cin, cout, kh, kw, pool = LAYERS[li]
    L = _lib.lib()
    lid = _lib.C.c_int(0)
    with on_device(w[0]):
        nbytes = L.zns_conv2d_packed_bytes(nbr, B, T, cin, cout, kh, kw, H_IN[li], int(transposed), _lib.C.byref(lid))
        if lid.value == 0:
            return None
        # the operand is built by a kernel on the CURRENT stream: a forward on another stream must not be handed a buffer
        # that may still be being written there (and must not return it to that stream's allocator pool while reading it)
        key = (int(transposed), lid.value, nbr, stream_ptr(w[0].device)) + tuple(t.data_ptr() for t in w)
        sig = tuple(t._version for t in w)
        with _frozen.lock:
            if _frozen.touched is None:        # the scope closed on another thread meanwhile
                return None
            cache = w[0].__dict__.get("_zns_pack")
            if cache is None:
                cache = w[0].__dict__["_zns_pack"] = {}
                _frozen.touched.append(w[0])
        hit = cache.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        buf = hit[1] if hit is not None else torch.empty(nbytes // 4, dtype=torch.float32, device=w[0].device)
        same_device(w[0], w, name)
        rc = L.zns_conv2d_pack_f32(nbr, B, T, cin, cout, kh, kw, H_IN[li], int(transposed), _pp(w), buf.data_ptr(),
                                   buf.numel() * 4, stream_ptr(w[0].device))
        _lib.check(rc, f"zns_conv2d_pack_f32[{name}]")
        cache[key] = (sig, buf)
    return buf


def invalidate_packed(module: torch.nn.Module) -> None:
    """Drop every cached packed operand of `module`'s parameters (inside a frozen_weights() block, after writing weights
    through `.data` or a raw pointer)."""
    for p in module.parameters():
        c = p.__dict__.get("_zns_pack")
        if c is not None:
            c.clear()


def bump_version(t: torch.Tensor) -> None:
    """Tell autograd (and packed_weights) that `t` -- and every view of it -- was written through its raw pointer."""
    torch.autograd.graph.increment_version(t)


def conv_fwd(li: int, nbr: int, B: int, T: int, x: torch.Tensor, w: Sequence[torch.Tensor],
             b: Sequence[torch.Tensor], drop_p: float, seed: int, want_argmax: bool, ws: torch.Tensor,
             x1: Optional[torch.Tensor] = None):
    """x1 (first layer of a two-branch stack only): branch 1's samples when the branches' inputs are separate tensors."""
    name, cin, cout, kh, kw, pool = LAYERS[li]
    H = H_IN[li]
    y = torch.empty((nbr * B, cout, H // pool, T), dtype=torch.float32, device=x.device)
    am = None
    if pool > 1 and want_argmax:
        am = torch.empty((nbr * B, cout, H // pool, T), dtype=torch.uint8, device=x.device)
    spec = None
    if want_argmax:  # = activations are being saved for a backward pass: keep the input spectrum for backward-weights
        with on_device(x):
            nspec = _lib.lib().zns_conv2d_spectrum_bytes(nbr, B, T, cin, cout, kh, kw, H)
        if nspec:
            spec = torch.empty(nspec // 4, dtype=torch.float32, device=x.device)
    same_device(x, (*w, *b, ws, x1), name)
    wpk = packed_weights(li, False, nbr, B, T, w)
    with on_device(x):
        rc = _lib.lib().zns_conv2d_fwd_f32(nbr, B, T, cin, cout, kh, kw, H, pool, x.data_ptr(),
                                           x1.data_ptr() if x1 is not None else None, _pp(w), _pp(b),
                                           y.data_ptr(), am.data_ptr() if am is not None else None,
                                           float(drop_p), int(seed) & 0xFFFFFFFF, li,
                                           spec.data_ptr() if spec is not None else None,
                                           wpk.data_ptr() if wpk is not None else None,
                                           ws.data_ptr(), ws.numel(), stream_ptr(x.device))
    _lib.check(rc, f"zns_conv2d_fwd_f32[{name}]")
    return y, am, spec


def conv_dgrad(li: int, nbr: int, B: int, T: int, dz: torch.Tensor, w: Sequence[torch.Tensor],
               x: Optional[torch.Tensor], inv_keep: float, prev_argmax: Optional[torch.Tensor], prev_pool: int,
               ws: torch.Tensor, dzspec: Optional[torch.Tensor] = None) -> torch.Tensor:
    name, cin, cout, kh, kw, pool = LAYERS[li]
    H = H_IN[li]
    dx = torch.empty((nbr * B, cin, H * prev_pool, T), dtype=torch.float32, device=dz.device)
    same_device(dz, (*w, x, prev_argmax, dzspec, ws), name)
    wpk = packed_weights(li, True, nbr, B, T, w)
    with on_device(dz):
        rc = _lib.lib().zns_conv2d_dgrad_f32(nbr, B, T, cin, cout, kh, kw, H, dz.data_ptr(), _pp(w), dx.data_ptr(),
                                             x.data_ptr() if x is not None else None, float(inv_keep),
                                             prev_argmax.data_ptr() if prev_argmax is not None else None, prev_pool,
                                             dzspec.data_ptr() if dzspec is not None else None,
                                             wpk.data_ptr() if wpk is not None else None,
                                             ws.data_ptr(), ws.numel(), stream_ptr(dz.device))
    _lib.check(rc, f"zns_conv2d_dgrad_f32[{name}]")
    return dx


def conv_wgrad(li: int, nbr: int, B: int, T: int, dz: torch.Tensor, x: torch.Tensor, dw: Sequence[torch.Tensor],
               db: Sequence[torch.Tensor], accumulate: bool, ws: torch.Tensor,
               xspec: Optional[torch.Tensor] = None, dzspec_out: Optional[torch.Tensor] = None,
               x1: Optional[torch.Tensor] = None) -> None:
    name, cin, cout, kh, kw, pool = LAYERS[li]
    H = H_IN[li]
    same_device(dz, (x, *dw, *db, xspec, dzspec_out, ws, x1), name)
    with on_device(dz):
        rc = _lib.lib().zns_conv2d_wgrad_f32(nbr, B, T, cin, cout, kh, kw, H, dz.data_ptr(), x.data_ptr(),
                                             x1.data_ptr() if x1 is not None else None, _pp(dw),
                                             _pp(db), int(bool(accumulate)),
                                             xspec.data_ptr() if xspec is not None else None,
                                             dzspec_out.data_ptr() if dzspec_out is not None else None,
                                             ws.data_ptr(), ws.numel(), stream_ptr(dz.device))
    _lib.check(rc, f"zns_conv2d_wgrad_f32[{name}]")


def dz_spectrum(li: int, nbr: int, B: int, T: int, device) -> Optional[torch.Tensor]:
    """Scratch for the dZ spectrum backward-weights hands to backward-data of the same layer (None: not applicable)."""
    name, cin, cout, kh, kw, pool = LAYERS[li]
    with torch.cuda.device(device):
        n = _lib.lib().zns_conv2d_dz_spectrum_bytes(nbr, B, T, cin, cout, kh, kw, H_IN[li])
    return torch.empty(n // 4, dtype=torch.float32, device=device) if n else None


def ntxent(a: torch.Tensor, p: torch.Tensor, rows: int, D: int, batch_len: int, tau: float, out3: torch.Tensor,
           ga: Optional[torch.Tensor], gp: Optional[torch.Tensor], grad_scale: float = 1.0,
           loss_out: Optional[torch.Tensor] = None) -> None:
    """zns_ntxent_f32 on a/p's device and current stream (scratch from workspace slot 2)."""
    same_device(a, (p, out3, ga, gp, loss_out), "NTXent (the reference moves anc_emb to the positives' device first, pretext.py:484)")
    L = _lib.lib()
    ws = workspace(L.zns_ntxent_workspace_bytes(rows), a.device, slot=2)
    with on_device(a):
        rc = L.zns_ntxent_f32(a.data_ptr(), p.data_ptr(), rows, D, batch_len, float(tau), out3.data_ptr(),
                              loss_out.data_ptr() if loss_out is not None else None,
                              ga.data_ptr() if ga is not None else None, gp.data_ptr() if gp is not None else None,
                              float(grad_scale), ws.data_ptr(), ws.numel(), stream_ptr(a.device))
    _lib.check(rc, "zns_ntxent_f32")


def scale_by(scale: torch.Tensor, x0: torch.Tensor, x1: Optional[torch.Tensor] = None):
    """x0 * scale (and x1 * scale) with `scale` a one-element DEVICE tensor -- the factor autograd hands a loss node's
    backward -- as one launch of zns_scale_f32: no host read-back, no ATen kernel.  Returns new tensors."""
    require_device(x0, "scale_by input")
    s = scale.reshape(-1)
    if s.numel() != 1 or s.dtype != torch.float32:
        raise RuntimeError("zeronotesamba_amd: the loss gradient must be one float32 element")
    x0 = x0.contiguous()
    y0 = torch.empty_like(x0)
    y1 = None
    if x1 is not None:
        x1 = x1.contiguous()
        y1 = torch.empty_like(x1)
    same_device(x0, (s, x1), "scale_by")
    with on_device(x0):
        rc = _lib.lib().zns_scale_f32(s.data_ptr(), x0.data_ptr(), y0.data_ptr(), x0.numel(),
                                      x1.data_ptr() if x1 is not None else None,
                                      y1.data_ptr() if y1 is not None else None, x1.numel() if x1 is not None else 0,
                                      stream_ptr(x0.device))
    _lib.check(rc, "zns_scale_f32")
    return (y0, y1) if x1 is not None else y0


def head_fwd(nbr: int, B: int, T: int, h: torch.Tensor, w: Sequence[torch.Tensor], b: Sequence[torch.Tensor]):
    emb = torch.empty((nbr * B, T), dtype=torch.float32, device=h.device)
    same_device(h, (*w, *b), "head_fwd")
    with on_device(h):
        rc = _lib.lib().zns_head_fwd_f32(nbr, B, T, 128, h.data_ptr(), _pp(w), _pp(b), emb.data_ptr(),
                                         stream_ptr(h.device))
    _lib.check(rc, "zns_head_fwd_f32")
    return emb


def head_bwd(nbr: int, B: int, T: int, h: torch.Tensor, emb: torch.Tensor, demb: torch.Tensor,
             w: Sequence[torch.Tensor], inv_keep: float, dw: Sequence[torch.Tensor], db: Sequence[torch.Tensor],
             accumulate: bool, ws: torch.Tensor, demb1: Optional[torch.Tensor] = None) -> torch.Tensor:
    """demb1: branch 1's gradient rows when the two embeddings' gradients are separate tensors (demb: branch 0's)."""
    dz = torch.empty_like(h)
    same_device(h, (emb, demb, *w, *dw, *db, ws, demb1), "head_bwd")
    with on_device(h):
        rc = _lib.lib().zns_head_bwd_f32(nbr, B, T, 128, h.data_ptr(), emb.data_ptr(), demb.data_ptr(),
                                         demb1.data_ptr() if demb1 is not None else None, _pp(w),
                                         float(inv_keep), dz.data_ptr(), _pp(dw), _pp(db), int(bool(accumulate)),
                                         ws.data_ptr(), ws.numel(), stream_ptr(h.device))
    _lib.check(rc, "zns_head_bwd_f32")
    return dz


class StackParams:
    """Per-branch parameter tensors of nbr DS_CNN stacks, in LAYERS order."""

    def __init__(self, conv_w: List[List[torch.Tensor]], conv_b: List[List[torch.Tensor]],
                 fc_w: List[torch.Tensor], fc_b: List[torch.Tensor]):
        self.conv_w, self.conv_b, self.fc_w, self.fc_b = conv_w, conv_b, fc_w, fc_b
        self.nbr = len(fc_w)

    def flat(self) -> List[torch.Tensor]:
        out: List[torch.Tensor] = []
        for br in range(self.nbr):
            for li in range(8):
                out += [self.conv_w[li][br], self.conv_b[li][br]]
            out += [self.fc_w[br], self.fc_b[br]]
        return out

    @staticmethod
    def from_flat(flat: Sequence[torch.Tensor], nbr: int) -> "StackParams":
        per = 18
        cw = [[flat[br * per + 2 * li] for br in range(nbr)] for li in range(8)]
        cb = [[flat[br * per + 2 * li + 1] for br in range(nbr)] for li in range(8)]
        fw = [flat[br * per + 16] for br in range(nbr)]
        fb = [flat[br * per + 17] for br in range(nbr)]
        return StackParams(cw, cb, fw, fb)


def stack_forward(x: torch.Tensor, P: StackParams, B: int, drop_p: float, seed: int, save: bool,
                  with_head: bool = True, ws_slot: int = 0, x1: Optional[torch.Tensor] = None):
    """x: [nbr*B, 1, 96, T] -> (emb [nbr*B, T] or None, acts [y1..y8], argmax {layer: u8}).
    x1: branch 1's [B, 1, 96, T] when the two branches' inputs are separate tensors (x is then branch 0's): nothing is
    concatenated, the first layer's kernels take both pointers.
    save=False is the inference form (nothing kept for a backward pass).  Packed weight operands are reused only inside
    a frozen_weights() block."""
    nbr = P.nbr
    T = x.shape[-1]
    with on_device(x):
        ws = workspace(conv_workspace_bytes(nbr, B, T), x.device, ws_slot)
    acts: List[torch.Tensor] = []
    argm: Dict[int, torch.Tensor] = {}
    cur = x
    for li in range(8):
        y, am, spec = conv_fwd(li, nbr, B, T, cur, P.conv_w[li], P.conv_b[li], drop_p, seed, save, ws,
                               x1=x1 if li == 0 else None)
        if save:
            acts.append(y)
            if am is not None:
                argm[li] = am
            if spec is not None:
                argm[100 + li] = spec   # input spectrum of layer li, reused by its backward-weights
        cur = y
    emb = head_fwd(nbr, B, T, cur, P.fc_w, P.fc_b) if with_head else None
    if not save:
        acts = [cur]
    return emb, acts, argm


def stack_backward(x: torch.Tensor, P: StackParams, B: int, acts: List[torch.Tensor], argm: Dict[int, torch.Tensor],
                   emb: Optional[torch.Tensor], demb: Optional[torch.Tensor], dh: Optional[torch.Tensor],
                   inv_keep: float, G: StackParams, accumulate: bool = False, conv_grads: bool = True,
                   after_layer=None, ws_slot: int = 0, x1: Optional[torch.Tensor] = None,
                   demb1: Optional[torch.Tensor] = None) -> None:
    """Writes parameter gradients into the tensors of G (same structure as P).

    demb: grad w.r.t. the embeddings (head fused) -- or dh: grad w.r.t. cv8's RAW output when the head
    was not part of the stack.  after_layer(name) is called after each layer's gradients are enqueued
    (fc1, cv8, ..., cv1): the hook the data-parallel trainer uses to start bucketed all-reduces.
    """
    nbr = P.nbr
    T = x.shape[-1]
    with on_device(x):
        ws = workspace(conv_workspace_bytes(nbr, B, T), x.device, ws_slot)
    if demb is not None:
        dz = head_bwd(nbr, B, T, acts[7], emb, demb, P.fc_w, inv_keep, G.fc_w, G.fc_b, accumulate, ws, demb1=demb1)
        if after_layer:
            after_layer("fc1")
    else:
        dz = dh
    if not conv_grads:
        return
    for li in range(7, 0, -1):
        xin = acts[li - 1]
        dzs = dz_spectrum(li, nbr, B, T, dz.device)     # dZ is transformed once for both calls below
        conv_wgrad(li, nbr, B, T, dz, xin, G.conv_w[li], G.conv_b[li], accumulate, ws, argm.get(100 + li), dzs)
        if after_layer:
            after_layer(LAYERS[li][0])
        prev_pool = LAYERS[li - 1][5]
        dz = conv_dgrad(li, nbr, B, T, dz, P.conv_w[li], xin, inv_keep, argm.get(li - 1), prev_pool, ws, dzs)
    conv_wgrad(0, nbr, B, T, dz, x, G.conv_w[0], G.conv_b[0], accumulate, ws, x1=x1)
    if after_layer:
        after_layer("cv1")
