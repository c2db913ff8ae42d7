"""Latency of the reference signature generate_XQT(signal: np.ndarray, sr, mode) -> np.ndarray, one 10 s clip per call
(host -> device -> host), on the GPU box: python3 tools/host_xqt_latency.py"""
import time, numpy as np, torch, sys
sys.path.insert(0,'.')
from zeronotesamba_amd.processing.input_rep import generate_XQT
rng=np.random.default_rng(0); y=(0.1*rng.standard_normal(160000)).astype(np.float32)
for _ in range(5): o=generate_XQT(y,16000,"vqt")
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(200): o=generate_XQT(y,16000,"vqt")
dt=(time.perf_counter()-t0)/200
print("generate_XQT host->host one clip: %.1f us, %.0f clips/s"%(dt*1e6,1/dt), o.shape, o.dtype)
