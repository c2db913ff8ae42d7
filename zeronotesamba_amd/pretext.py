"""The pretext training step of /root/reference/zeroNoteSamba/pretext.py on MI355X.

Two layers of API:

* `train_epoch` / `val_epoch` -- the reference's functions (pretext.py:453-592) with the same
  arguments and return values, driving the drop-in modules through autograd with whatever optimizer
  the caller built.  Running means are accumulated on the device; one host sync per epoch instead of
  the reference's 49 per step.
* `PretextTrainer` -- the fused form used by bench.py: parameters, gradients and Adam moments live in
  three flat HBM buffers; one step = twin-branch forward -> fused NT-Xent fwd+bwd -> backward ->
  (data-parallel: bucketed RCCL all-reduce of the flat gradient, overlapped with the rest of the
  backward) -> ONE fused Adam launch.  No autograd graph, no per-parameter launches.

Data parallelism (new: the reference has none): one process per GPU, every rank owns
whole 16-item batches (16 shifts of one clip, pretext.py:312-318) and computes its local NT-Xent;
gradients are summed over ranks and scaled by 1/world_size inside the Adam kernel.
"""
from typing import Any, Iterable, List, Optional, Sequence, Tuple

import torch

from . import _lib, _ops
from .distributed import GradBucketReducer, make_buckets
from .models.loss_functions import NTXent
from .models.models import DS_CNN, Pretext_CNN

_BUCKET_MIN_FLOATS = 1 << 20  # fire an all-reduce once >= 4 MiB of gradients are ready


def split_batch(batch: torch.Tensor, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """pretext.py:476-477: anchors = batch[:,0:1], postves = batch[:,1:2], moved to the device."""
    return batch[:, 0:1, :, :].to(device), batch[:, 1:2, :, :].to(device)


def train_epoch(model: torch.nn.Module, train_loader: Iterable, criterion: NTXent,
                optimizer: torch.optim.Optimizer, pt_task: str = "zerons") -> Tuple[torch.nn.Module, float, float, float]:
    """pretext.py:453-524."""
    device = next(model.parameters()).device
    acc = torch.zeros(3, dtype=torch.float64, device=device)
    model.train()
    batch_idx = -1
    if pt_task not in ("zerons", "clmr"):
        raise ValueError("Which pretext task are we running?")
    for batch_idx, [batch] in enumerate(train_loader):
        anchors, postves = split_batch(batch, device)
        optimizer.zero_grad()
        if pt_task == "zerons":
            anc_emb, pos_emb = model(anchors, postves)
        else:  # clmr: one shared DS_CNN on both views (pretext.py:494-511)
            anc_emb = model(anchors)
            pos_emb = model(postves)
        loss, stats = _loss_and_stats(criterion, anc_emb, pos_emb)
        loss.backward()
        optimizer.step()
        acc += stats.double()
    n = batch_idx + 1
    l, p, ng = (acc / n).tolist()
    print("*** Mean training batch loss is {:.3f}.".format(l))
    print("*** Mean training anchor / positive similiarity is {:.3f}.".format(p))
    print("*** Mean training anchor / negative similiarity is {:.3f}.".format(ng))
    return model, l, p, ng


def val_epoch(model: torch.nn.Module, val_loader: Iterable, criterion: NTXent,
              optimizer: Optional[torch.optim.Optimizer] = None, pt_task: str = "zerons") -> Tuple[float, float, float]:
    """pretext.py:527-592 (model.eval(), torch.no_grad())."""
    device = next(model.parameters()).device
    acc = torch.zeros(3, dtype=torch.float64, device=device)
    model.eval()
    batch_idx = -1
    if pt_task not in ("zerons", "clmr"):
        raise ValueError("Which pretext task are we running?")
    with torch.no_grad():
        for batch_idx, [batch] in enumerate(val_loader):
            anchors, postves = split_batch(batch, device)
            if pt_task == "zerons":
                anc_emb, pos_emb = model(anchors, postves)
            else:
                anc_emb = model(anchors)
                pos_emb = model(postves)
            _, stats = _loss_and_stats(criterion, anc_emb, pos_emb)
            acc += stats.double()
    n = batch_idx + 1
    l, p, ng = (acc / n).tolist()
    return l, p, ng


def _loss_and_stats(criterion: NTXent, a: torch.Tensor, p: torch.Tensor):
    """criterion(a, p) without the device->host copy of the two cosine means."""
    from .models.loss_functions import _NTXentFn
    if a.shape[0] > criterion.batch_len:
        raise IndexError(f"NTXent: {a.shape[0]} rows > batch_len={criterion.batch_len}")
    loss, stats = _NTXentFn.apply(a, p, criterion.batch_len, criterion.temperature)
    return loss, stats


class ShiftBank:
    """Resident memory bank + on-device shift batching (pretext.py:255-263, 285-292, 308-321).

    The reference materialises `new_train_bank` on the host: for every clip it draws `batch_len` DISTINCT start
    frames with `random.sample(range(0, 313), batch_len)` and copies 313-frame crops of the 626-frame bank row
    (5.5 GB of host memcpy per sub-epoch).  Here the `(N, 2, 96, 626)` bank lives in HBM (28 800 clips = 13.8 GB)
    and one gather kernel cuts the 16 crops of a clip straight into the branch-major layout the twin launch
    consumes.  Same sampling law (Python's `random.sample`, so `random.seed` reproduces the reference's draws),
    same "one clip per batch, order preserved" grouping (DataLoader(shuffle=False), pretext.py:320-321).
    """

    def __init__(self, bank, batch_len: int = 16, crop: int = 313, device: str = "cuda"):
        b = torch.as_tensor(bank, dtype=torch.float32)
        if b.dim() != 4 or b.shape[1] != 2 or b.shape[2] != _ops.N_BINS:
            raise ValueError(f"bank must be (N, 2, 96, T), got {tuple(b.shape)}")
        if b.shape[3] - crop < batch_len:
            raise ValueError("not enough distinct start frames for batch_len crops")
        self.bank = b.to(device).contiguous()
        self.batch_len, self.crop = batch_len, crop
        self.n_clips, self.bank_T = b.shape[0], b.shape[3]

    def shuffle_(self, generator: Optional[torch.Generator] = None) -> None:
        """np.random.shuffle(train_bank) (pretext.py:282), on the device."""
        perm = torch.randperm(self.n_clips, generator=generator).to(self.bank.device)
        self.bank = self.bank[perm].contiguous()

    def draw_starts(self, rng=None) -> List[int]:
        import random as _random
        r = rng if rng is not None else _random
        return r.sample(range(0, self.bank_T - self.crop), self.batch_len)      # pretext.py:312

    def batch(self, clip: int, starts: Optional[Sequence[int]] = None, rng=None) -> torch.Tensor:
        """The 16 shifted crops of `clip`, branch-major [2*batch_len, 1, 96, crop] (anchors first)."""
        if not 0 <= clip < self.n_clips:
            raise IndexError(f"clip {clip} out of range [0, {self.n_clips})")
        if starts is None:
            starts = self.draw_starts(rng)
        if len(starts) != self.batch_len or min(starts) < 0 or max(starts) + self.crop > self.bank_T:
            raise ValueError("bad start frames")
        dev = self.bank.device
        idx = torch.tensor([[clip] * self.batch_len, list(starts)], dtype=torch.int32).to(dev, non_blocking=True)
        out = torch.empty((2 * self.batch_len, 1, _ops.N_BINS, self.crop), dtype=torch.float32, device=dev)
        rc = _lib.lib().zns_shift_gather_f32(self.bank.data_ptr(), self.bank_T, idx[0].data_ptr(), idx[1].data_ptr(),
                                             self.batch_len, self.crop, out.data_ptr(), _ops.stream_ptr())
        _lib.check(rc, "zns_shift_gather_f32")
        self._keep = idx     # the index tensor must outlive the enqueued kernel
        return out

    def batches(self, clips: Iterable[int], rng=None):
        for c in clips:
            yield self.batch(int(c), rng=rng)


class PretextTrainer:
    """Fused pretext step over flat parameter / gradient / moment buffers (see module docstring)."""

    def __init__(self, model: Pretext_CNN, batch_len: int = 16, temperature: float = 0.25, lr: float = 1e-6,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, seed: int = 0,
                 process_group: Any = None, train: bool = True):
        self.model = model
        self.batch_len, self.temperature = batch_len, temperature
        self.lr, self.betas, self.eps = lr, betas, eps
        self.seed, self.step_count = seed, 0
        self.train_mode = train
        self.pg = process_group
        self.world = 1
        self.rank = 0
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            self.rank = torch.distributed.get_rank(process_group)
        self._flatten()

    # ------------------------------------------------------------------ flat buffers
    def _flatten(self) -> None:
        m = self.model
        dev = next(m.parameters()).device
        _ops.require_device(next(m.parameters()), "model parameters")
        # backward order: fc1, cv8 .. cv1; inside a layer: anchor w, b, postve w, b
        order: List[Tuple[str, torch.nn.Parameter]] = []
        self.layer_slices = {}
        off = 0
        names = ["fc1"] + [n for n, *_ in reversed(_ops.LAYERS)]
        for lname in names:
            start = off
            for brname, br in (("anchor", m.anchor), ("postve", m.postve)):
                mod = br.fc1 if lname == "fc1" else getattr(br.pretrained, lname)
                for pname in ("weight", "bias"):
                    p = getattr(mod, pname)
                    order.append((f"{brname}.{lname}.{pname}", p))
                    off += (p.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned
            self.layer_slices[lname] = (start, off)
        self.n_flat = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(off, dtype=torch.float32, device=dev)
        o = 0
        self._views = []
        for name, p in order:
            n = p.numel()
            pv = self.flat_p[o:o + n].view_as(p)
            pv.copy_(p.data)
            p.data = pv
            gv = self.flat_g[o:o + n].view_as(p)
            p.grad = gv
            self._views.append((p, pv.data_ptr()))
            o += (n + 3) // 4 * 4
        self.P = _ops.StackParams.from_flat(m.anchor.stack_params() + m.postve.stack_params(), 2)
        grads = [p.grad for p in m.anchor.stack_params() + m.postve.stack_params()]
        self.G = _ops.StackParams.from_flat(grads, 2)
        # all-reduce buckets: consecutive layers (in backward order) until >= _BUCKET_MIN_FLOATS
        self.buckets = make_buckets([(n, *self.layer_slices[n]) for n in names], _BUCKET_MIN_FLOATS)
        self.reducer = GradBucketReducer(self.flat_g, self.buckets, self.pg) if self.world > 1 else None

    def _check_flat(self) -> None:
        for p, ptr in self._views:
            if p.data_ptr() != ptr:
                raise RuntimeError("PretextTrainer: the model's parameters were moved/replaced after the trainer "
                                   "was built (e.g. model.to(...)); build the trainer after placing the model")

    # ------------------------------------------------------------------ one step
    def _next_seed(self) -> int:
        s = (self.seed * 0x9E3779B1 + self.step_count * 0x85EBCA77 + self.rank * 0xC2B2AE3D + 1) & 0x7FFFFFFF
        return s

    def stack_inputs(self, batch: torch.Tensor) -> torch.Tensor:
        """[B,2,96,T] (the reference's DataLoader layout) -> branch-major [2B,1,96,T] on the device."""
        dev = self.flat_p.device
        b = batch.to(dev, non_blocking=True)
        return b.transpose(0, 1).reshape(2 * b.shape[0], 1, b.shape[2], b.shape[3]).contiguous()

    def step(self, x: torch.Tensor, update: bool = True) -> torch.Tensor:
        """x: [2B,1,96,T] branch-major.  Returns a device tensor (loss, mean pos cos, mean neg cos); no host sync."""
        self._check_flat()
        B = x.shape[0] // 2
        T = x.shape[-1]
        if B > self.batch_len:
            raise IndexError(f"PretextTrainer: {B} rows > batch_len={self.batch_len}")
        drop_p = _ops.DROP_P if self.train_mode else 0.0
        inv_keep = 1.0 / (1.0 - drop_p)
        seed = self._next_seed()
        emb, acts, argm = _ops.stack_forward(x, self.P, B, drop_p, seed, save=True)
        out3 = torch.empty(3, dtype=torch.float32, device=x.device)
        demb = torch.empty_like(emb)
        rc = _lib.lib().zns_ntxent_f32(emb.data_ptr(), emb[B:].data_ptr(), B, T, self.batch_len,
                                       float(self.temperature), out3.data_ptr(), demb.data_ptr(),
                                       demb[B:].data_ptr(), 1.0, _ops.stream_ptr())
        _lib.check(rc, "zns_ntxent_f32")
        hook = self.reducer.layer_done if self.reducer is not None else None
        _ops.stack_backward(x, self.P, B, acts, argm, emb, demb, None, inv_keep, self.G, accumulate=False,
                            after_layer=hook)
        if self.reducer is not None:
            self.reducer.wait()
        if update:
            self.step_count += 1
            rc = _lib.lib().zns_adam_step_f32(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                              self.flat_v.data_ptr(), self.n_flat, self.lr, self.betas[0],
                                              self.betas[1], self.eps, self.step_count, 1.0 / self.world,
                                              _ops.stream_ptr())
            _lib.check(rc, "zns_adam_step_f32")
        return out3

    @torch.no_grad()
    def evaluate(self, x: torch.Tensor) -> torch.Tensor:
        B = x.shape[0] // 2
        T = x.shape[-1]
        emb, _, _ = _ops.stack_forward(x, self.P, B, 0.0, 0, save=False)
        out3 = torch.empty(3, dtype=torch.float32, device=x.device)
        rc = _lib.lib().zns_ntxent_f32(emb.data_ptr(), emb[B:].data_ptr(), B, T, self.batch_len,
                                       float(self.temperature), out3.data_ptr(), None, None, 1.0, _ops.stream_ptr())
        _lib.check(rc, "zns_ntxent_f32")
        return out3

    def train_epoch(self, loader: Iterable) -> Tuple[float, float, float]:
        """Fused equivalent of pretext.train_epoch: returns (mean loss, mean pos cos, mean neg cos).
        `loader` yields either the reference's `[batch]` with batch [B,2,96,T] (DataLoader over a TensorDataset) or
        branch-major device tensors [2B,1,96,T] (ShiftBank.batches)."""
        acc = torch.zeros(3, dtype=torch.float64, device=self.flat_p.device)
        n = 0
        for item in loader:
            x = item if torch.is_tensor(item) else self.stack_inputs(item[0])
            acc += self.step(x).double()
            n += 1
        l, p, ng = (acc / max(n, 1)).tolist()
        return l, p, ng
