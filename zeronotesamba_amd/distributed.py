"""Data-parallel plumbing for the pretext step: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm; "gloo" in the CPU tests).  No kernels here, so it runs on CPU tensors too.

The pretext path shards by CLIP: a batch is 16 shifts of one clip (pretext.py:312-318) and NT-Xent only
couples items inside that batch, so ranks exchange nothing on the data path.  The only collective is the
sum of the flat gradient, issued bucket by bucket while the backward pass is still producing the
earlier layers' gradients (fc1, cv8, ... first), and the 1/world_size scale is folded into the Adam
kernel.  xGMI is point-to-point, so buckets are few and large (>= 4 MiB) rather than per parameter.
"""
import os
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, timeout_s: Optional[float] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  A single process (no env) needs no group.  timeout_s: rendezvous and
    collective timeout (torch's default is 10-30 minutes: a rank that never shows up should fail the job sooner)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        kw = {}
        if timeout_s is not None:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_units(n_units: int, rank: int, world: int, drop_last: bool = True) -> List[int]:
    """Clips (= 16-item batches) owned by `rank`.  Every rank gets the SAME number of units: each training step
    all-reduces the gradient, so a rank holding one batch more than the others would wait in RCCL for peers that have
    already left the epoch (until the watchdog kills the job).  As torch's DistributedSampler: drop_last=True drops
    the n_units % world remainder (contiguous blocks of n_units // world); drop_last=False pads the tail by wrapping
    around to the first units, so nothing is skipped and a few units are seen twice per epoch."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} not in a world of {world}")
    if drop_last:
        per = n_units // world
        return list(range(rank * per, (rank + 1) * per))
    per = -(-n_units // world) if n_units > 0 else 0
    return [(rank * per + i) % n_units for i in range(per)]


def broadcast_tensors(tensors: Sequence[torch.Tensor], src: int = 0, group: Any = None) -> None:
    """Overwrite every rank's `tensors` with those of the group's rank `src` (replica initialisation: parameters, Adam
    moments).  `src` is a rank INSIDE `group`; torch.distributed.broadcast wants the global rank."""
    gsrc = dist.get_global_rank(group, src) if group is not None else src
    for t in tensors:
        dist.broadcast(t, src=gsrc, group=group)


def make_buckets(layer_slices: Sequence[Tuple[str, int, int]], min_floats: int) -> List[Tuple[str, int, int]]:
    """Merge consecutive (name, start, end) slices of the flat gradient, given in the order the backward
    pass completes them, into buckets of at least `min_floats`.  A bucket is named after its LAST layer:
    it may be reduced as soon as that layer's gradient has been enqueued."""
    out: List[Tuple[str, int, int]] = []
    bstart = None
    for i, (name, s, e) in enumerate(layer_slices):
        if bstart is None:
            bstart = s
        if e - bstart >= min_floats or i == len(layer_slices) - 1:
            out.append((name, bstart, e))
            bstart = None
    return out


class GradBucketReducer:
    """Sums slices of one flat gradient buffer across ranks, asynchronously, bucket by bucket."""

    def __init__(self, flat_grad: torch.Tensor, buckets: Sequence[Tuple[str, int, int]], group: Any = None,
                 always: bool = False):
        """always=True issues the collectives even in a one-rank group (smoke-testing RCCL on a single GPU)."""
        self.always = always
        self.flat = flat_grad
        self.by_last = {name: (s, e) for name, s, e in buckets}
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.handles: List[Any] = []
        covered = sorted((s, e) for _, s, e in buckets)
        pos = covered[0][0] if covered else 0
        for s, e in covered:
            if s != pos:
                raise ValueError("buckets must tile the flat gradient without gaps or overlap")
            pos = e
        self.span = (covered[0][0], pos) if covered else (0, 0)

    def layer_done(self, name: str) -> None:
        """Call after layer `name`'s gradient kernels have been enqueued on the current stream."""
        if (self.world > 1 or self.always) and name in self.by_last:
            s, e = self.by_last[name]
            self.handles.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                                async_op=True))

    def wait(self) -> float:
        """Block the current stream on every outstanding bucket; returns the scale (1/world) the
        optimizer must apply to turn the sum into the mean."""
        for h in self.handles:
            h.wait()
        self.handles = []
        return 1.0 / self.world
