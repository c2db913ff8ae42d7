"""Experiment: does the filter bank (matrix cores) of batch i overlap the decimation pyramid (float64 vector pipes) of
batch i+1 when they run on two streams?  Prints serial vs pipelined time per batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zeronotesamba_amd import _lib
from zeronotesamba_amd.processing.input_rep import generate_XQT_batch

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
g = torch.Generator().manual_seed(1)
y = (0.1 * torch.randn(64, 160000, generator=g)).cuda().repeat(n_clips // 64, 1).contiguous()
nb = _lib.lib().zns_xqt_workspace_bytes(n_clips, 160000, 0)
ws = [torch.empty(nb, dtype=torch.uint8, device="cuda") for _ in range(2)]
out = [torch.empty((n_clips, 96, 626), device="cuda") for _ in range(2)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
NB = 8

def serial():
    for i in range(NB):
        generate_XQT_batch(y, 16000, "vqt", out=out[i & 1], ws=ws[i & 1])

def pipelined():
    ev_p = [torch.cuda.Event() for _ in range(NB)]
    ev_f = [torch.cuda.Event() for _ in range(NB)]
    for i in range(NB):
        with torch.cuda.stream(s1):                      # pyramid of batch i (needs ws[i&1] free: filter i-2 done)
            if i >= 2:
                s1.wait_event(ev_f[i - 2])
            generate_XQT_batch(y, 16000, "vqt", out=out[i & 1], ws=ws[i & 1], stages=1)
            ev_p[i].record(s1)
        with torch.cuda.stream(s2):                      # filter bank of batch i
            s2.wait_event(ev_p[i])
            generate_XQT_batch(y, 16000, "vqt", out=out[i & 1], ws=ws[i & 1], stages=2)
            ev_f[i].record(s2)

for fn in (serial, pipelined, serial, pipelined):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(fn.__name__, n_clips, "clips:", (time.perf_counter() - t0) / NB * 1e3, "ms per batch")
