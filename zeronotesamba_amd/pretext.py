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
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, _ops
from .distributed import GradBucketReducer, make_buckets
from .models.loss_functions import NTXent
from .models.models import DS_CNN, Pretext_CNN

_BUCKET_MIN_FLOATS = 1 << 20  # fire an all-reduce once >= 4 MiB of gradients are ready


def split_batch(batch: torch.Tensor, device, device1=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """pretext.py:476-477: anchors = batch[:,0:1] -> device0, postves = batch[:,1:2] -> device1."""
    return batch[:, 0:1, :, :].to(device), batch[:, 1:2, :, :].to(device if device1 is None else device1)


def _branch_devices(model: torch.nn.Module, pt_task: str):
    """(device0, device1) of pretext.py:24-25,199-200: the anchor branch's and the positive branch's device (the same
    one unless the caller placed the branches on two GPUs as the reference does); "clmr": the one model's device."""
    if pt_task == "zerons" and hasattr(model, "anchor") and hasattr(model, "postve"):
        return next(model.anchor.parameters()).device, next(model.postve.parameters()).device
    d = next(model.parameters()).device
    return d, d


class _StatsLog:
    """The (loss, pos cos, neg cos) triples of an epoch's steps: the NT-Xent kernel writes each straight into a row of a
    device table, and the epoch means are ONE reduction at the end -- no per-step accumulate launches, no host sync (the
    reference adds three `.item()`s per step, pretext.py:513-515)."""

    def __init__(self, device, rows: int = 1024):
        self.device, self.rows, self.tables, self.n = device, rows, [], 0

    def row(self) -> torch.Tensor:
        i = self.n % self.rows
        if i == 0:
            self.tables.append(torch.empty((self.rows, 3), dtype=torch.float32, device=self.device))
        self.n += 1
        return self.tables[-1][i]

    def means(self, empty=float("nan")) -> Tuple[float, float, float]:
        if self.n == 0:
            return (empty, empty, empty)
        tot = None
        for k, t in enumerate(self.tables):
            used = min(self.rows, self.n - k * self.rows)
            s = t[:used].double().sum(0)
            tot = s if tot is None else tot + s
        l, p, ng = (tot / self.n).tolist()
        return l, p, ng


def train_epoch(model: torch.nn.Module, train_loader: Iterable, criterion: NTXent,
                optimizer: torch.optim.Optimizer, pt_task: str = "zerons") -> Tuple[torch.nn.Module, float, float, float]:
    """pretext.py:453-524."""
    if pt_task not in ("zerons", "clmr"):
        raise ValueError("Which pretext task are we running?")
    device0, device1 = _branch_devices(model, pt_task)
    log = _StatsLog(device1)
    model.train()
    for [batch] in train_loader:
        anchors, postves = split_batch(batch, device0, device1)
        optimizer.zero_grad()
        if pt_task == "zerons":
            anc_emb, pos_emb = model(anchors, postves)
            anc_emb = anc_emb.to(device1)                      # pretext.py:484 (a no-op on one device)
        else:  # clmr: one shared DS_CNN on both views (pretext.py:494-511)
            anc_emb = model(anchors)
            pos_emb = model(postves)
        loss, _ = _loss_and_stats(criterion, anc_emb, pos_emb, log.row())
        loss.backward()
        optimizer.step()
    l, p, ng = log.means()
    print("*** Mean training batch loss is {:.3f}.".format(l))
    print("*** Mean training anchor / positive similiarity is {:.3f}.".format(p))
    print("*** Mean training anchor / negative similiarity is {:.3f}.".format(ng))
    return model, l, p, ng


def val_epoch(model: torch.nn.Module, val_loader: Iterable, criterion: NTXent,
              optimizer: Optional[torch.optim.Optimizer] = None, pt_task: str = "zerons") -> Tuple[float, float, float]:
    """pretext.py:527-592 (model.eval(), torch.no_grad())."""
    if pt_task not in ("zerons", "clmr"):
        raise ValueError("Which pretext task are we running?")
    device0, device1 = _branch_devices(model, pt_task)
    log = _StatsLog(device1)
    model.eval()
    # nothing writes the weights inside this loop: the packed weight operands are built once for the whole epoch
    with torch.no_grad(), _ops.frozen_weights():
        for [batch] in val_loader:
            anchors, postves = split_batch(batch, device0, device1)
            if optimizer is not None:
                optimizer.zero_grad()        # pretext.py:552,572 clear the gradients here too
            if pt_task == "zerons":
                anc_emb, pos_emb = model(anchors, postves)
                anc_emb = anc_emb.to(device1)
            else:
                anc_emb = model(anchors)
                pos_emb = model(postves)
            _loss_and_stats(criterion, anc_emb, pos_emb, log.row())
    return log.means()


def _loss_and_stats(criterion: NTXent, a: torch.Tensor, p: torch.Tensor, out3: Optional[torch.Tensor] = None):
    """criterion(a, p) without the device->host copy of the two cosine means; the triple goes to `out3` when given."""
    from .models.loss_functions import _NTXentFn
    if a.shape[0] > criterion.batch_len:
        raise IndexError(f"NTXent: {a.shape[0]} rows > batch_len={criterion.batch_len}")
    loss, stats = _NTXentFn.apply(a, p, criterion.batch_len, criterion.temperature, out3)
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
        import numpy as np
        self.bank = b.to(device).contiguous()
        self.batch_len, self.crop = batch_len, crop
        self.n_clips, self.bank_T = b.shape[0], b.shape[3]
        self.order = np.arange(self.n_clips)          # logical clip (after shuffles) -> physical bank row

    def shuffle_(self, rng=None) -> None:
        """np.random.shuffle(train_bank) (pretext.py:282) without moving 13.8 GB: the permutation is drawn exactly as
        numpy draws it for an in-place shuffle of the first axis (`rng.permutation(n)` consumes the same stream as
        `rng.shuffle(bank)`: one Fisher-Yates pass), from numpy's global state by default so that `np.random.seed`
        reproduces the reference's clip order, and is applied through the clip -> bank-row table `self.order` that
        `batch()` consults.  rng: None (np.random), a np.random.RandomState or a np.random.Generator."""
        import numpy as np
        r = np.random if rng is None else rng
        perm = np.asarray(r.permutation(self.n_clips))
        self.order = self.order[perm]

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
        row = int(self.order[clip])
        idx = torch.tensor([[row] * self.batch_len, list(starts)], dtype=torch.int32).to(dev, non_blocking=True)
        out = torch.empty((2 * self.batch_len, 1, _ops.N_BINS, self.crop), dtype=torch.float32, device=dev)
        with _ops.on_device(self.bank):
            rc = _lib.lib().zns_shift_gather_f32(self.bank.data_ptr(), self.bank_T, idx[0].data_ptr(),
                                                 idx[1].data_ptr(), self.batch_len, self.crop, out.data_ptr(),
                                                 _ops.stream_ptr(dev))
        _lib.check(rc, "zns_shift_gather_f32")
        self._keep = idx     # the index tensor must outlive the enqueued kernel
        return out

    def batches(self, clips: Iterable[int], rng=None):
        """One batch per clip, in order.  As the reference does when it materialises `new_train_bank`
        (pretext.py:308-318: ALL clips' start frames are drawn before the first training step), the draws for every clip
        are made up front, in clip order -- and travel to the device as ONE pinned index table, so the per-batch work on
        the host is a single kernel launch with no host-to-device copy."""
        clips = [int(c) for c in clips]
        if not clips:
            return
        for c in clips:
            if not 0 <= c < self.n_clips:
                raise IndexError(f"clip {c} out of range [0, {self.n_clips})")
        bl = self.batch_len
        table = torch.empty((len(clips), 2, bl), dtype=torch.int32, pin_memory=True)
        tn = table.numpy()
        for i, c in enumerate(clips):
            tn[i, 0, :] = int(self.order[c])
            tn[i, 1, :] = self.draw_starts(rng)
        dev = self.bank.device
        dtab = table.to(dev, non_blocking=True)
        L = _lib.lib()
        for i in range(len(clips)):
            out = torch.empty((2 * bl, 1, _ops.N_BINS, self.crop), dtype=torch.float32, device=dev)
            with _ops.on_device(self.bank):
                rc = L.zns_shift_gather_f32(self.bank.data_ptr(), self.bank_T, dtab[i, 0].data_ptr(),
                                            dtab[i, 1].data_ptr(), bl, self.crop, out.data_ptr(), _ops.stream_ptr(dev))
            _lib.check(rc, "zns_shift_gather_f32")
            out._zns_keep = (dtab, table)     # the index table must outlive the enqueued kernels
            yield out


class PretextTrainer:
    """Fused pretext step over flat parameter / gradient / moment buffers (see module docstring)."""

    def __init__(self, model: torch.nn.Module, batch_len: int = 16, temperature: float = 0.25, lr: float = 1e-6,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, seed: int = 0,
                 process_group: Any = None, train: bool = True, pt_task: Optional[str] = None,
                 data_parallel: Optional[bool] = None):
        """model: Pretext_CNN (pt_task "zerons", two independent branches) or DS_CNN (pt_task "clmr": ONE
        network applied to both views, pretext.py:494-511 -- run here as a single stack over the 2B rows, so
        its weight gradient is the sum over both views exactly as two autograd passes would give).

        Data parallelism is opt-in: pass `process_group=` (a torch.distributed group) or `data_parallel=True` (the
        default group).  An initialised default group alone does NOT switch it on.  When on, rank 0's parameters
        (and Adam moments) are broadcast at construction, as DistributedDataParallel does, so replicas start
        identical whatever each rank's initialisation was; every rank must then call step() the same number of
        times (each step all-reduces: see distributed.shard_units for equal-length shards)."""
        if pt_task is None:
            pt_task = "zerons" if isinstance(model, Pretext_CNN) else "clmr"
        if pt_task not in ("zerons", "clmr"):
            raise ValueError("Which pretext task are we running?")
        if (pt_task == "zerons") != isinstance(model, Pretext_CNN) or (pt_task == "clmr" and not isinstance(model, DS_CNN)):
            raise TypeError(f"pt_task={pt_task!r} needs a {'Pretext_CNN' if pt_task == 'zerons' else 'DS_CNN'}, "
                            f"got {type(model).__name__}")
        self.pt_task = pt_task
        self.model = model
        self.batch_len, self.temperature = batch_len, temperature
        self.lr, self.betas, self.eps = lr, betas, eps
        self.seed, self.step_count, self.micro_count = seed, 0, 0
        self.train_mode = train
        self.pg = process_group
        self.world = 1
        self.rank = 0
        if data_parallel is None:
            data_parallel = process_group is not None
        if data_parallel:
            if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
                raise RuntimeError("PretextTrainer(data_parallel=True): torch.distributed is not initialised "
                                   "(zeronotesamba_amd.distributed.init_from_env)")
            self.world = torch.distributed.get_world_size(process_group)
            self.rank = torch.distributed.get_rank(process_group)
        self._flatten()
        if self.world > 1:
            self.broadcast_state()

    # ------------------------------------------------------------------ flat buffers
    def _flatten(self) -> None:
        m = self.model
        dev = next(m.parameters()).device
        _ops.require_device(next(m.parameters()), "model parameters")
        if any(p.device != dev for p in m.parameters()):
            raise RuntimeError("PretextTrainer: every parameter must live on ONE device (the fused step runs both "
                               "branches in one launch per layer); the reference's branch-per-device placement "
                               "(pretext.py:199-200) is served by the module path, pretext.train_epoch")
        # backward order: fc1, cv8 .. cv1; inside a layer: anchor w, b, postve w, b
        branches = (("anchor", m.anchor), ("postve", m.postve)) if self.pt_task == "zerons" else (("model", m),)
        self.nbr = len(branches)
        order: List[Tuple[str, torch.nn.Parameter]] = []
        self.layer_slices = {}
        off = 0
        names = ["fc1"] + [n for n, *_ in reversed(_ops.LAYERS)]
        for lname in names:
            start = off
            for brname, br in branches:
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
        plist = [p for _, br in branches for p in br.stack_params()]
        self.P = _ops.StackParams.from_flat(plist, self.nbr)
        self.G = _ops.StackParams.from_flat([p.grad for p in plist], self.nbr)
        # all-reduce buckets: consecutive layers (in backward order) until >= _BUCKET_MIN_FLOATS
        self.buckets = make_buckets([(n, *self.layer_slices[n]) for n in names], _BUCKET_MIN_FLOATS)
        self.reducer = GradBucketReducer(self.flat_g, self.buckets, self.pg) if self.world > 1 else None

    def broadcast_state(self, src: int = 0) -> None:
        """Make every replica's parameters, Adam moments and step counter equal to rank `src`'s (construction, and
        after loading a checkpoint on one rank only)."""
        from .distributed import broadcast_tensors
        sc = torch.tensor([self.step_count], dtype=torch.int64, device=self.flat_p.device)
        broadcast_tensors((self.flat_p, self.flat_m, self.flat_v, sc), src=src, group=self.pg)
        self.step_count = int(sc.item())
        self._params_written()

    def _params_written(self) -> None:
        """The flat buffer was written through its raw pointer (fused Adam, broadcast): bump every parameter's autograd
        version -- a Parameter whose `.data` is a view of the flat buffer keeps its OWN version counter, so bumping the
        buffer alone would leave the cached packed operands (_ops.packed_weights) looking current."""
        _ops.bump_version(self.flat_p)
        for p, _ in self._views:
            _ops.bump_version(p)

    def _check_flat(self) -> None:
        for p, ptr in self._views:
            if p.data_ptr() != ptr:
                raise RuntimeError("PretextTrainer: the model's parameters were moved/replaced after the trainer "
                                   "was built (e.g. model.to(...)); build the trainer after placing the model")

    # ------------------------------------------------------------------ one step
    def _drop_p(self) -> float:
        """Dropout probability of a training step: the modules' own `dp.p` (models.py:30), 0 when train=False."""
        if not self.train_mode:
            return 0.0
        m = self.model
        cnns = (m.anchor.pretrained, m.postve.pretrained) if self.pt_task == "zerons" else (m.pretrained,)
        ps = {float(c.dp.p) for c in cnns}
        if len(ps) != 1:
            raise ValueError(f"PretextTrainer: the branches have different dropout probabilities {sorted(ps)}")
        p = ps.pop()
        if not 0.0 <= p < 1.0:
            raise ValueError(f"dropout probability has to be in [0, 1), but got {p}")
        return p

    def _next_seed(self) -> int:
        # step_count only advances with an update: the micro-batch counter keeps the dropout masks of the batches of one
        # gradient accumulation apart
        s = (self.seed * 0x9E3779B1 + self.step_count * 0x85EBCA77 + self.rank * 0xC2B2AE3D
             + self.micro_count * 0x27D4EB2F + 1) & 0x7FFFFFFF
        return s

    def stack_inputs(self, batch: torch.Tensor) -> torch.Tensor:
        """[B,2,96,T] (the reference's DataLoader layout) -> branch-major [2B,1,96,T] on the device."""
        dev = self.flat_p.device
        b = batch.to(dev, non_blocking=True)
        return b.transpose(0, 1).reshape(2 * b.shape[0], 1, b.shape[2], b.shape[3]).contiguous()

    def step(self, x: torch.Tensor, update: bool = True, accumulate: bool = False,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: [2B,1,96,T] branch-major.  Returns a device tensor (loss, mean pos cos, mean neg cos) -- `out` when given (three
        contiguous float32: a row of an epoch's statistics table) -- ; no host sync.
        update=False leaves the gradient of this batch in `flat_g` and the parameters alone; accumulate=True ADDS this
        batch's gradient to what `flat_g` holds (the backward-weights kernels accumulate in place) -- gradient
        accumulation over k batches is `step(x1, update=False)`, `step(x2, update=False, accumulate=True)`, ...,
        `apply_update(1 / k)`.  Every micro-batch draws its own dropout masks.  Not with data_parallel (the bucketed
        all-reduce sums what is in the buffer: reduce once, after the last micro-batch)."""
        self._check_flat()
        if accumulate and self.reducer is not None:
            raise RuntimeError("PretextTrainer.step(accumulate=True) is not supported with data_parallel=True")
        B = x.shape[0] // 2
        T = x.shape[-1]
        if B > self.batch_len:
            raise IndexError(f"PretextTrainer: {B} rows > batch_len={self.batch_len}")
        _ops.same_device(self.flat_p, (x,), "PretextTrainer.step")
        with _ops.on_device(x):
            return self._step(x, B, T, update, accumulate, out)

    def _step(self, x: torch.Tensor, B: int, T: int, update: bool, accumulate: bool = False,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
        drop_p = self._drop_p()
        inv_keep = 1.0 / (1.0 - drop_p)
        seed = self._next_seed()
        Bs = 2 * B // self.nbr     # rows per stack: B for the twin launch, 2B for the shared (clmr) network
        emb, acts, argm = _ops.stack_forward(x, self.P, Bs, drop_p, seed, save=True)
        out3 = out if out is not None else torch.empty(3, dtype=torch.float32, device=x.device)
        demb = torch.empty_like(emb)
        _ops.ntxent(emb[:B], emb[B:], B, T, self.batch_len, self.temperature, out3, demb[:B], demb[B:])
        hook = self.reducer.layer_done if self.reducer is not None else None
        _ops.stack_backward(x, self.P, Bs, acts, argm, emb, demb, None, inv_keep, self.G, accumulate=accumulate,
                            after_layer=hook)
        self.micro_count += 1
        if self.reducer is not None:
            self.reducer.wait()
        if update:
            self.apply_update(1.0 / self.world)
        return out3

    def apply_update(self, grad_scale: float = 1.0) -> None:
        """ONE fused Adam launch over the flat buffers with the gradient read as `grad_scale * flat_g` (1/world after a
        summing all-reduce; 1/k after accumulating k batches with step(update=False, accumulate=True))."""
        self.step_count += 1
        self.micro_count = 0
        dev = self.flat_p.device
        with _ops.on_device(self.flat_p):
            rc = _lib.lib().zns_adam_step_f32(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                              self.flat_v.data_ptr(), self.n_flat, self.lr, self.betas[0],
                                              self.betas[1], self.eps, self.step_count, float(grad_scale),
                                              _ops.stream_ptr(dev))
        _lib.check(rc, "zns_adam_step_f32")
        self._params_written()

    @torch.no_grad()
    def evaluate(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(loss, mean pos cos, mean neg cos) of one branch-major batch without dropout and without a backward pass.
        Inside a `_ops.frozen_weights()` block (evaluate_epoch holds one) the packed weight operands are reused."""
        B = x.shape[0] // 2
        T = x.shape[-1]
        _ops.same_device(self.flat_p, (x,), "PretextTrainer.evaluate")
        emb, _, _ = _ops.stack_forward(x, self.P, 2 * B // self.nbr, 0.0, 0, save=False)
        out3 = out if out is not None else torch.empty(3, dtype=torch.float32, device=x.device)
        _ops.ntxent(emb[:B], emb[B:], B, T, self.batch_len, self.temperature, out3, None, None)
        return out3

    def evaluate_epoch(self, loader: Iterable) -> Tuple[float, float, float]:
        """Fused equivalent of pretext.val_epoch (pretext.py:527-592): mean (loss, pos cos, neg cos) over the loader."""
        log = _StatsLog(self.flat_p.device)
        with _ops.frozen_weights():
            for item in loader:
                x = item if torch.is_tensor(item) else self.stack_inputs(item[0])
                self.evaluate(x, out=log.row())
        return log.means(empty=0.0)

    def train_epoch(self, loader: Iterable) -> Tuple[float, float, float]:
        """Fused equivalent of pretext.train_epoch: returns (mean loss, mean pos cos, mean neg cos).
        `loader` yields either the reference's `[batch]` with batch [B,2,96,T] (DataLoader over a TensorDataset) or
        branch-major device tensors [2B,1,96,T] (ShiftBank.batches)."""
        log = _StatsLog(self.flat_p.device)
        for item in loader:
            x = item if torch.is_tensor(item) else self.stack_inputs(item[0])
            self.step(x, out=log.row())
        return log.means(empty=0.0)


# ---------------------------------------------------------------------------------------------------------------
# Online path: stems -> mix -> RMS-gated random crop -> VQT -> resident bank, all on
# the device.  pretext.py:30-86 (drum_anchor_positive) and :143-170 (bank fill) without the pickle round trip.

def mix_stems(stems: Dict[str, Any], device=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """pretext.py:46-64: positive = the "drums" stem, anchor = float32 sum of every other stem in dict order."""
    from .processing.stem_check import to_device_signal
    others = [to_device_signal(sig, f"stem {name!r}") for name, sig in stems.items() if name != "drums"]
    if not others:
        raise Exception("Anchor is still None.")
    if "drums" not in stems:
        raise KeyError("stems has no 'drums' entry")
    pos = to_device_signal(stems["drums"], "stem 'drums'")
    n = others[0].numel()
    if any(o.numel() != n for o in others) or pos.numel() != n:
        raise ValueError("all stems must have the same length")
    anchor = torch.empty(n, dtype=torch.float32, device=pos.device)
    _ops.same_device(pos, others, "mix_stems")
    with _ops.on_device(pos):
        rc = _lib.lib().zns_stem_mix_f32(_lib.ptr_array([o.data_ptr() for o in others]), len(others), n,
                                         anchor.data_ptr(), _ops.stream_ptr(pos.device))
    _lib.check(rc, "zns_stem_mix_f32")
    anchor._zns_keep = others      # sources must outlive the enqueued kernel
    return anchor, pos


def select_clip(anchor: torch.Tensor, positive: torch.Tensor, length: int, lower_p: float, upper_p: float,
                rng=None, speculate: int = 8) -> int:
    """The `while rms_bool == False` loop of pretext.py:66-84: draw `randint(0, stop)` until the crop passes the RMS
    gate, halving lower_p after the 10th try.  `speculate` candidates are gated per launch; the generator is rewound
    so it ends up having consumed exactly the draws the reference would have made."""
    import random as _random
    from .processing.stem_check import gate_counts
    r = rng if rng is not None else _random
    n = int(length) * 16000
    stop = anchor.numel() - n - 1
    if stop < 0:
        raise ValueError(f"signal of {anchor.numel()} samples is shorter than a {length} s clip")
    tries = 0
    while True:
        state = r.getstate()
        cands = [r.randint(0, stop) for _ in range(speculate)]
        counts, n_frames, _, _ = gate_counts(anchor, positive, cands, n)
        for j, c in enumerate(counts.tolist()):
            ok = lower_p < c / n_frames <= upper_p
            tries += 1
            if tries > 9:
                lower_p = lower_p / 2
            if ok:
                r.setstate(state)
                for _ in range(j + 1):
                    r.randint(0, stop)
                return cands[j]


def _yml(ymldict) -> Tuple[int, str, float, float]:
    return (int(float(ymldict.get("clip_len", -1))), str(ymldict.get("input_mode", "")),
            float(ymldict.get("lower_p", -1.0)), float(ymldict.get("upper_p", -1.0)))


def _yml_exact(ymldict) -> bool:
    try:
        return bool(ymldict.get("xqt_exact", False))
    except AttributeError:
        return False


def drum_anchor_positive(stems: Dict[str, Any], ymldict, rng=None, exact: bool = False):
    """
    Function for generating anchor and positive samples.
    -- stems: dictionary of stem names and signals
    -- ymldict: YAML dictionary (`xqt_exact: true` there selects the same as exact=True)
    -- exact: the XQT pyramid that rounds like resampy tap by tap (processing/input_rep.py::generate_XQT_batch)
    Same 4-tuple as pretext.py:30-86 (numpy arrays): full-length anchor mix, drums, and the two XQTs of the crop.
    """
    from .processing.input_rep import generate_XQT_batch
    length, mode, lower_p, upper_p = _yml(ymldict)
    anchor, pos = mix_stems(stems)
    start = select_clip(anchor, pos, length, lower_p, upper_p, rng)
    n = length * 16000
    crops = torch.stack([anchor[start:start + n], pos[start:start + n]])
    xqt = generate_XQT_batch(crops, 16000, mode, exact=exact or _yml_exact(ymldict))
    return anchor.cpu().numpy(), pos.cpu().numpy(), xqt[0].cpu().numpy(), xqt[1].cpu().numpy()


def _first_pass(count: int, n_frames: int, lower_p: float, upper_p: float) -> bool:
    return lower_p < count / n_frames <= upper_p            # stem_check.py:47-51, first try (no halving yet)


def fill_memory_bank(all_stems: Dict[str, Dict[str, Any]], fps: Sequence[str], ymldict, rng=None,
                     chunk: int = 256, files_per_sync: int = 32, exact: bool = False) -> torch.Tensor:
    """The bank-filling loop of create_memory_bank (pretext.py:143-170) for stems that are already loaded:
    one (anchor, positive) XQT pair per file path, in `fps` order, written straight into a device-resident
    `(len(fps), 2, 96, T)` bank (feed it to ShiftBank).  Audio decoding and the pickle dump are out of scope.
    exact (or `xqt_exact: true` in the YAML dictionary): the bank is computed ONCE and trained on for every epoch -- the XQT
    pyramid that rounds like resampy tap by tap costs about twice the default's time here and nothing afterwards.

    The reference consumes ONE random stream across files (each file draws `randint(0, stop)` until its crop passes the
    RMS gate), so file i+1's draws depend on how many file i needed.  To keep that stream exact without a host round
    trip -- or even a kernel launch -- per file, `files_per_sync` files are handled per synchronisation under the guess
    that each passes at its FIRST draw: ONE launch mixes and crops all of them into the staging rows the XQT will read
    (zns_stage_crops_f32, a pinned pointer / start table), ONE launch gates the rows, and all counts are read back at
    once.  Files up to the first miss are final; the generator is rewound to just before the missing file's first draw,
    that file runs the reference's loop (select_clip, 8 speculative candidates per launch) and is staged again, and the
    guessing resumes behind it.  The draws consumed are exactly the reference's whatever the pass rate is
    (tests/test_gpu_online.py)."""
    import random as _random
    from .processing.input_rep import generate_XQT_batch
    from .processing.stem_check import to_device_signal
    length, mode, lower_p, upper_p = _yml(ymldict)
    if mode not in ("vqt", "cqt"):
        raise Exception("Mode can only be vqt or cqt!")
    if not len(fps):
        raise ValueError("fill_memory_bank: no file paths")
    r = rng if rng is not None else _random
    L = _lib.lib()
    n = length * 16000
    T = L.zns_xqt_frames(n)
    n_frames = L.zns_rms_frames(n)
    MAX_OTHER = 8
    row_len = 3 + MAX_OTHER
    chunk = max(1, min(chunk, len(fps)))

    N = len(fps)
    # ---- every file is walked ONCE: its stems validated, their device pointers recorded in a numpy table, so that a group's
    # crop table is three array assignments; the walk over the NEXT group's files runs while the GPU works on the current one
    ptrs = np.zeros((chunk, 1 + MAX_OTHER), dtype=np.int64)       # [drums, other stems in dict order ...]
    n_oth = np.zeros(chunk, dtype=np.int64)
    lens = [0] * chunk
    alive: list = [None] * chunk                                   # (others, drums) of the chunk's files: keeps uploads alive
    dev = None

    loaded = 0                                                     # rows [0, loaded) of the current chunk are in the tables

    def load_rows(base: int, upto: int) -> None:
        nonlocal dev, loaded
        for k in range(loaded, min(upto, chunk, N - base)):
            ptrs[k] = 0
            name_ = fps[base + k].strip()
            others, drums = [], None
            for name, sig in all_stems[name_].items():
                t = sig if (type(sig) is torch.Tensor and sig.is_cuda and sig.dtype == torch.float32 and sig.dim() == 1
                            and sig.is_contiguous()) else to_device_signal(sig, f"stem {name!r}")
                if name == "drums":
                    drums = t
                else:
                    others.append(t)
            if not others:
                raise Exception("Anchor is still None.")
            if drums is None:
                raise KeyError("stems has no 'drums' entry")
            if len(others) > MAX_OTHER:
                raise ValueError(f"more than {MAX_OTHER} non-drum stems")
            m = drums.numel()
            if dev is None:
                dev = drums.device
            for q, o in enumerate(others):
                if o.numel() != m:
                    raise ValueError("all stems must have the same length")
                if o.device != dev:       # the crop table hands raw pointers to zns_stage_crops_f32, which cannot tell whose memory they are
                    _ops.same_device(drums, [o], f"stems of {name_!r}")
                ptrs[k, 1 + q] = o.data_ptr()
            if drums.device != dev:
                raise RuntimeError(f"zeronotesamba_amd: stems of {name_!r}: tensors on different devices ({dev} and {drums.device})")
            if m - n - 1 < 0:
                raise ValueError(f"signal of {m} samples is shorter than a {length} s clip")
            ptrs[k, 0] = drums.data_ptr()
            n_oth[k] = len(others)
            lens[k] = m
            alive[k] = (others, drums)
            loaded = k + 1

    load_rows(0, max(1, files_per_sync))
    bank = torch.empty((N, 2, 96, T), dtype=torch.float32, device=dev)
    staged = torch.empty((chunk, 2, n), dtype=torch.float32, device=dev)
    row_off = torch.arange(chunk, dtype=torch.int64, device=dev) * (2 * n)       # start of staging row k, for the gate
    counts = torch.empty(chunk, dtype=torch.int32, device=dev)
    fsync = max(1, files_per_sync)
    # pinned crop tables: every group synchronises (the counts) before the next one refills `tab`; a missed file's single
    # row goes through `tab1`, whose previous use lies at least one group synchronisation back
    tab = torch.zeros((chunk, row_len), dtype=torch.int64, pin_memory=True)
    tab1 = torch.zeros((1, row_len), dtype=torch.int64, pin_memory=True)

    def stage(first_row: int, k0: int, starts, table=None):
        """Staging rows first_row .. of the chunk's files k0 .. k0 + len(starts) - 1 cropped at `starts`."""
        F = len(starts)
        table = tab if table is None else table
        tn = table.numpy()
        tn[:F, 0] = n_oth[k0:k0 + F]
        tn[:F, 1] = starts
        tn[:F, 2:] = ptrs[k0:k0 + F]
        dtab = table[:F].to(dev, non_blocking=True)
        with _ops.on_device(staged):
            rc = L.zns_stage_crops_f32(dtab.data_ptr(), F, MAX_OTHER, n, staged[first_row].data_ptr(), _ops.stream_ptr(dev))
        _lib.check(rc, "zns_stage_crops_f32")
        return dtab

    pos = 0                       # next file whose crop start is unknown
    xqt_from = 0                  # first file whose rows have not gone through the XQT yet
    group = fsync                 # files per synchronisation: doubles after a group without a miss, back to `files_per_sync` on one
    randint = r.randint
    while pos < N:
        base = pos - pos % chunk                                           # this chunk's first file = staging row 0
        if pos == base and base > 0:
            loaded = 0
        k0 = pos - base
        F = min(group, N - pos, chunk - k0)
        load_rows(base, k0 + F)
        state0 = r.getstate()                 # one snapshot per group (it is a 625-word tuple): a miss replays the draws before it
        starts = [randint(0, lens[k] - n - 1) for k in range(k0, k0 + F)]
        keep = stage(k0, k0, starts)
        with _ops.on_device(staged):
            rc = L.zns_rms_gate_f32(staged.data_ptr(), staged.data_ptr() + 4 * n, row_off[k0:].data_ptr(), F, n, None, None,
                                    counts.data_ptr(), _ops.stream_ptr(dev))
        _lib.check(rc, "zns_rms_gate_f32")
        load_rows(base, k0 + F + min(2 * group, chunk))                    # the next group's files, while the GPU works
        got = counts[:F].tolist()                                          # the one synchronisation of this group
        miss = next((j for j in range(F) if not _first_pass(got[j], n_frames, lower_p, upper_p)), F)
        pos += miss
        if miss < F:
            # a miss: this file runs the reference's loop from its first draw; later files' draws start after its last
            r.setstate(state0)
            for q in range(miss):
                randint(0, lens[k0 + q] - n - 1)
            others, drums = alive[k0 + miss]
            # the same float32 sum in the same (dict) order, from the vectors already on the device
            anchor, _ = mix_stems({**{f"other{q}": o for q, o in enumerate(others)}, "drums": drums})
            st = select_clip(anchor, drums, length, lower_p, upper_p, r)
            keep = stage(k0 + miss, k0 + miss, [st], tab1)
            pos += 1
            group = fsync
        else:
            group = min(2 * group, chunk)
        del keep
        # rows that are final go through the XQT in spans of at least 96 files (192 clips: the fused kernel's batch), so that
        # the transform of one span runs under the host's walk over the next
        if pos - base == chunk or pos == N or pos - xqt_from >= 96:
            k_lo, k_hi = xqt_from - base, pos - base
            generate_XQT_batch(staged[k_lo:k_hi].reshape(2 * (k_hi - k_lo), n), 16000, mode,
                               out=bank[xqt_from:pos].view(2 * (k_hi - k_lo), 96, T), exact=exact or _yml_exact(ymldict))
            xqt_from = pos
    return bank
