"""Experiment: the two branches on two CU-MASKED streams (hipExtStreamCreateWithCUMask), so that one branch's HBM-bound
transform kernels run beside the other's MFMA-bound GEMM kernels on disjoint halves of the chip.  Timing only: the two
streams share cached scratch (dZ spectra), results are not checked."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from zeronotesamba_amd import _ops

dev = torch.device("cuda", 0)
torch.cuda.init()
torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


B, T = bench.B, bench.T
m = bench.build_model(dev)
P = [_ops.StackParams.from_flat(m.anchor.stack_params(), 1), _ops.StackParams.from_flat(m.postve.stack_params(), 1)]
G = [_ops.StackParams.from_flat([torch.zeros_like(p) for p in br.stack_params()], 1) for br in (m.anchor, m.postve)]
x = bench.synth_batch(1, dev).transpose(0, 1).reshape(2 * B, 1, 96, T).contiguous()
xs = [x[:B].contiguous(), x[B:].contiguous()]
P2 = _ops.StackParams.from_flat(m.anchor.stack_params() + m.postve.stack_params(), 2)
G2 = _ops.StackParams.from_flat([torch.zeros_like(p) for p in m.anchor.stack_params() + m.postve.stack_params()], 2)


def step_two(streams):
    cur = torch.cuda.current_stream()
    outs = [None, None]
    for i in (0, 1):
        streams[i].wait_stream(cur)
    for i in (0, 1):
        with torch.cuda.stream(streams[i]):
            outs[i] = _ops.stack_forward(xs[i], P[i], B, 0.1, 7 + i, True, ws_slot=2 + i)
    for i in (0, 1):
        cur.wait_stream(streams[i])
    emb = torch.cat([outs[0][0], outs[1][0]])
    demb = torch.ones_like(emb) * 1e-3
    for i in (0, 1):
        streams[i].wait_stream(cur)
    for i in (0, 1):
        with torch.cuda.stream(streams[i]):
            e, acts, argm = outs[i]
            _ops.stack_backward(xs[i], P[i], B, acts, argm, e, demb[i * B:(i + 1) * B], None, 1 / 0.9, G[i], ws_slot=2 + i)
    for i in (0, 1):
        cur.wait_stream(streams[i])


def step_batched():
    e, acts, argm = _ops.stack_forward(x, P2, B, 0.1, 7, True)
    demb = torch.ones_like(e) * 1e-3
    _ops.stack_backward(x, P2, B, acts, argm, e, demb, None, 1 / 0.9, G2)


def timed(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(name, "ms/step (fwd+bwd only): %.3f" % ((time.perf_counter() - t0) / 10 * 1e3), flush=True)


F = 0xFFFFFFFF
timed("batched (nbr=2, one stream)", step_batched)
timed("two plain streams", lambda s=[torch.cuda.Stream(), torch.cuda.Stream()]: step_two(s))
timed("two masked streams, low/high 128 CUs", lambda s=[masked_stream([F] * 4 + [0] * 4), masked_stream([0] * 4 + [F] * 4)]: step_two(s))
timed("two masked streams, even/odd CUs", lambda s=[masked_stream([0x55555555] * 8), masked_stream([0xAAAAAAAA] * 8)]: step_two(s))
timed("two masked streams, 3/4 + 3/4 overlapping", lambda s=[masked_stream([F] * 6 + [0] * 2), masked_stream([0] * 2 + [F] * 6)]: step_two(s))
