"""Experiment: does running the two branches as two half-size launch sequences (spectra of one branch ~ the 256 MB
Infinity Cache) beat the fused twin launch?  Forward only (eval, no_grad) and full trainer-style step via modules."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
m = bench.build_model(dev).eval()
batch = bench.synth_batch(100, dev)
a, p = batch[:, 0:1].contiguous(), batch[:, 1:2].contiguous()

def timed(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3

with torch.no_grad():
    print("twin forward   ", timed(lambda: m(a, p)), "ms")
    print("split forward  ", timed(lambda: (m.anchor(a), m.postve(p))), "ms")
    h = a.shape[0] // 2
    print("quarter forward", timed(lambda: (m.anchor(a[:h]), m.anchor(a[h:]), m.postve(p[:h]), m.postve(p[h:]))), "ms")
