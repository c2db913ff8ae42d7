#!/usr/bin/env python3
"""generate_XQT throughput (BASELINE config 2): n_clips x 160 000 samples -> n_clips x 96 x 626, inputs resident in HBM.
Reports clips/s and achieved GB/s against the algorithmic bytes (640 KB in + 240 KB out per clip)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zeronotesamba_amd.processing.input_rep import generate_XQT_batch

STAGES = 3


def run(n_clips, iters):
    g = torch.Generator().manual_seed(1234)
    y = (0.1 * torch.randn(n_clips, 160000, generator=g)).cuda()
    res = torch.empty((n_clips, 96, 626), dtype=torch.float32, device="cuda")

    def timed(stages):
        for _ in range(2):
            generate_XQT_batch(y, 16000, "vqt", out=res, stages=stages)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            generate_XQT_batch(y, 16000, "vqt", out=res, stages=stages)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters

    dt = timed(STAGES)
    flags = STAGES & ~3
    pyr, fir = timed(1 | flags), timed(2 | flags)
    alg = n_clips * (160000 * 4 + 96 * 626 * 4)
    return {"n_clips": n_clips, "stages": STAGES, "ms": dt * 1e3, "pyramid_ms": pyr * 1e3, "filter_ms": fir * 1e3, "clips_per_s": n_clips / dt, "algorithmic_GBps": alg / dt / 1e9,
            "frac_of_8TBps": alg / dt / 8e12}

def cpu_oracle(n_clips=2):
    """The numpy restatement of the reference's librosa/resampy path, one process, per-clip loop as the reference does."""
    import numpy as np
    from oracle import vqt_ref
    rng = np.random.default_rng(1234)
    y = (0.1 * rng.standard_normal((n_clips, 160000))).astype(np.float32)
    vqt_ref.generate_XQT(y[0], 16000, "vqt")          # builds and caches the filter bank
    t0 = time.perf_counter()
    for i in range(n_clips):
        vqt_ref.generate_XQT(y[i], 16000, "vqt")
    dt = (time.perf_counter() - t0) / n_clips
    return {"cpu_oracle_s_per_clip": dt, "cpu_oracle_clips_per_s": 1.0 / dt, "kind": "port (numpy, 1 process)"}


if __name__ == "__main__":
    if "--stages" in sys.argv:       # e.g. 7 = 3 | ZNS_XQT_SEPARATE, 11 = 3 | ZNS_XQT_FUSED, 19 = 3 | ZNS_XQT_EXACT
        STAGES = int(sys.argv[sys.argv.index("--stages") + 1])
    if "--dbg" in sys.argv:          # phase-skipping bits of the fused kernel (WRONG RESULTS: timing bounds only), include/zns_test.h
        from zeronotesamba_amd import _lib as _l
        _l.check(_l.lib().zns_test_set_tuning(_l.ZNS_TUNE_XQT_DBG, int(sys.argv[sys.argv.index("--dbg") + 1])), "dbg")
    if "--waves16" in sys.argv:      # the fused kernel's wave layout: 0 = 16 waves (default), -1 = 12 waves (include/zns_test.h)
        from zeronotesamba_amd import _lib as _l
        _l.check(_l.lib().zns_test_set_tuning(_l.ZNS_TUNE_XQT_WAVES16, int(sys.argv[sys.argv.index("--waves16") + 1])), "waves16")
    if "--plan" in sys.argv:         # decimation chains as digits, e.g. 223 (include/zns_test.h: ZNS_TUNE_XQT_PLAN)
        from zeronotesamba_amd import _lib as _l
        _l.check(_l.lib().zns_test_set_tuning(_l.ZNS_TUNE_XQT_PLAN, int(sys.argv[sys.argv.index("--plan") + 1])), "plan")
    sizes = ((32, 50), (64, 30), (128, 20), (256, 10), (512, 10), (4096, 3))
    if "--only" not in sys.argv: sizes = (sizes[0], sizes[4], sizes[5])
    if "--only" in sys.argv:
        only = int(sys.argv[sys.argv.index("--only") + 1])
        sizes = tuple(x for x in sizes if x[0] == only)
    for n, it in sizes:
        print(json.dumps(run(n, it)))
    if "--cpu" in sys.argv:
        print(json.dumps(cpu_oracle()))
