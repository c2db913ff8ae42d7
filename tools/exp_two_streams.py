"""Experiment: run the two independent branches on two HIP streams so that one branch's HBM-bound transform kernels
can overlap the other's MFMA-bound GEMM kernels."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from zeronotesamba_amd import _lib, _ops

dev = torch.device("cuda", 0)
B, T = bench.B, bench.T
m = bench.build_model(dev)
P = [_ops.StackParams.from_flat(m.anchor.stack_params(), 1), _ops.StackParams.from_flat(m.postve.stack_params(), 1)]
G = [_ops.StackParams.from_flat([torch.zeros_like(p) for p in br.stack_params()], 1) for br in (m.anchor, m.postve)]
x = bench.synth_batch(1, dev).transpose(0, 1).reshape(2 * B, 1, 96, T).contiguous()
xs = [x[:B].contiguous(), x[B:].contiguous()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
P2 = _ops.StackParams.from_flat(m.anchor.stack_params() + m.postve.stack_params(), 2)
G2 = _ops.StackParams.from_flat([torch.zeros_like(p) for p in m.anchor.stack_params() + m.postve.stack_params()], 2)


SKEW = int(sys.argv[1]) if len(sys.argv) > 1 else 0       # cycles stream 1 sleeps before each pass (breaks the lockstep)


def step_two_streams(stagger: bool):
    cur = torch.cuda.current_stream()
    outs = [None, None]
    for i in (0, 1):
        streams[i].wait_stream(cur)
    for i in (0, 1):
        with torch.cuda.stream(streams[i]):
            if i == 1 and SKEW:
                torch.cuda._sleep(SKEW)
            outs[i] = _ops.stack_forward(xs[i], P[i], B, 0.1, 7 + i, True, ws_slot=2 + i)
    for i in (0, 1):
        cur.wait_stream(streams[i])
    emb = torch.cat([outs[0][0], outs[1][0]])
    demb = torch.ones_like(emb) * 1e-3     # stands in for NT-Xent (0.14 ms)
    for i in (0, 1):
        streams[i].wait_stream(cur)
    for i in (0, 1):
        with torch.cuda.stream(streams[i]):
            if i == 1 and SKEW:
                torch.cuda._sleep(SKEW)
            e, acts, argm = outs[i]
            _ops.stack_backward(xs[i], P[i], B, acts, argm, e, demb[i * B:(i + 1) * B], None, 1 / 0.9, G[i], ws_slot=2 + i)
    for i in (0, 1):
        cur.wait_stream(streams[i])


def step_batched():
    e, acts, argm = _ops.stack_forward(x, P2, B, 0.1, 7, True)
    demb = torch.ones_like(e) * 1e-3
    _ops.stack_backward(x, P2, B, acts, argm, e, demb, None, 1 / 0.9, G2)


for name, fn in (("batched (nbr=2, one stream)", step_batched), ("two streams", lambda: step_two_streams(False))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(name, "ms/step (fwd+bwd only):", (time.perf_counter() - t0) / 10 * 1e3)
