#!/usr/bin/env python3
"""pretext.val_epoch (pretext.py:527-592) over 8 validation batches, for rocprofv3: the weight-pack kernels must appear
ONCE per layer however many batches run (the packed operands are cached per weight version).
rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o val -- python3 tools/prof_val_epoch.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from zeronotesamba_amd.models.loss_functions import NTXent
from zeronotesamba_amd.pretext import val_epoch

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
crit = NTXent(bench.B, bench.TAU)
batches = [[bench.synth_batch(200 + i, "cpu")] for i in range(8)]
print("val_epoch (loss, pos, neg):", val_epoch(model, batches, crit))
