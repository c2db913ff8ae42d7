#!/usr/bin/env python3
"""The literal drop-in step (modules + autograd + NTXent + torch.optim.Adam) alone, for rocprofv3:
rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o mp -- python3 tools/prof_module_path.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(bench.measure_module_path(torch.device("cuda", 0), 6))
