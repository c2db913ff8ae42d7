#!/usr/bin/env python3
"""Where a step of the fused generate_XQT kernel goes: s_memtime stamps of workgroup 0 (ZNS_TUNE_XQT_STAMPS), steps 64..95.
Stamps per wave: 0 step start, 1 (pyramid: inputs requested / filter: tiles staged), 2 (pyramid: outputs written / filter:
big octave multiplied and stored), 3 arrival at the barrier, 4 release.  Prints per-wave medians in cycles.
Run on the GPU box: python tools/xqt_stamps.py [n_clips]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zeronotesamba_amd import _lib
from zeronotesamba_amd.processing.input_rep import generate_XQT_batch

_lib.lib().zns_test_set_tuning(_lib.ZNS_TUNE_XQT_STAMPS, 1)
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator().manual_seed(1)
y = (0.1 * torch.randn(min(n_clips, 512), 160000, generator=g)).cuda()
y = y.repeat((n_clips + y.shape[0] - 1) // y.shape[0], 1)[:n_clips].contiguous()
for _ in range(3):
    generate_XQT_batch(y, 16000, "vqt")
buf = (C.c_ulonglong * (32 * 12 * 8))()
_lib.check(_lib.lib().zns_xqt_debug_stamps(buf, 32 * 12 * 8), "stamps")
t = np.array(list(buf), dtype=np.int64).reshape(32, 12, 8)[:, :, :5]
step = np.median(np.diff(t[:, 0, 0]))
print(f"step length (wave 0, start to start): median {step:.0f} cycles")
names = ["start->1", "1->2", "2->3 (to barrier)", "3->4 (in barrier)", "start->barrier"]
print("wave  " + "  ".join(f"{n:>18s}" for n in names))
for w in range(12):
    d = [np.median(t[:, w, 1] - t[:, w, 0]), np.median(t[:, w, 2] - t[:, w, 1]), np.median(t[:, w, 3] - t[:, w, 2]),
         np.median(t[:, w, 4] - t[:, w, 3]), np.median(t[:, w, 3] - t[:, w, 0])]
    print(f"{w:4d}  " + "  ".join(f"{v:18.0f}" for v in d))
