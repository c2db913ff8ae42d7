"""CPU, gloo, world_size 2: the data-parallel plumbing of the pretext step (no kernels involved)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zeronotesamba_amd.distributed import GradBucketReducer, broadcast_tensors, make_buckets, shard_units


def test_shard_units_are_equal_length():
    """Every rank must run the same number of steps (each step is a collective): shards have EQUAL length."""
    for n, w in ((1440, 8), (10, 3), (2, 4), (0, 2), (7, 2)):
        parts = [shard_units(n, r, w) for r in range(w)]
        assert len({len(p) for p in parts}) == 1                       # equal length on every rank
        flat = sum(parts, [])
        assert len(set(flat)) == len(flat) == (n // w) * w             # drop_last: disjoint, remainder dropped
        assert all(0 <= u < n for u in flat)
        if n:
            padded = [shard_units(n, r, w, drop_last=False) for r in range(w)]
            assert len({len(p) for p in padded}) == 1 and len(padded[0]) == -(-n // w)
            assert set(sum(padded, [])) == set(range(n))               # padding: every unit is seen at least once
    with pytest.raises(ValueError):
        shard_units(4, 2, 2)


def test_make_buckets_tiles_in_backward_order():
    slices = [("fc1", 0, 260), ("cv8", 260, 820000), ("cv7", 820000, 2330000), ("cv6", 2330000, 16100000),
              ("cv1", 16100000, 16104300)]
    b = make_buckets(slices, 1 << 20)
    assert b[0] == ("cv7", 0, 2330000) and b[1] == ("cv6", 2330000, 16100000) and b[-1] == ("cv1", 16100000, 16104300)
    assert [x[1] for x in b[1:]] == [x[2] for x in b[:-1]]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 3000
        flat = torch.arange(n, dtype=torch.float32) * (rank + 1)          # rank-specific "gradient"
        slices = [("fc1", 0, 100), ("cv8", 100, 1500), ("cv7", 1500, 2000), ("cv1", 2000, 3000)]
        buckets = make_buckets(slices, 1000)
        red = GradBucketReducer(flat, buckets)
        for name, _, _ in slices:                                          # backward order
            red.layer_done(name)
        scale = red.wait()
        expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ok = torch.allclose(flat, expect) and abs(scale - 1.0 / world) < 1e-12
        # mean gradient == what one process would see on the concatenated batches (8e: local NT-Xent, averaged)
        ok = ok and torch.allclose(flat * scale, torch.arange(n, dtype=torch.float32) * 1.5)
        # replicas that were initialised differently start from rank 0's state (PretextTrainer.broadcast_state)
        pr = torch.full((5,), float(rank + 7))
        mo = torch.full((3,), float(rank))
        broadcast_tensors((pr, mo), src=0)
        ok = ok and bool((pr == 7.0).all()) and bool((mo == 0.0).all())
        # every rank owns whole clips
        mine = list(shard_units(7, rank, world, drop_last=False))
        q.put((rank, bool(ok), mine))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    units = sum((m for _, _, m in res), [])
    assert set(units) == set(range(7)) and len(units) == 8      # equal shards of 4, one unit wrapped around


def test_gaps_in_buckets_are_rejected():
    with pytest.raises(ValueError):
        GradBucketReducer(torch.zeros(10), [("a", 0, 4), ("b", 5, 10)])
