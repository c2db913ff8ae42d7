#!/usr/bin/env python3
"""Sum of kernel durations per generate_XQT call from a rocprofv3 --kernel-trace csv (GPU time, independent of host
launch overhead): usage trace_sum.py <dir> <kernels per call> <calls to average>."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
per, calls = int(sys.argv[2]), int(sys.argv[3])
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "zns::" in r["Kernel_Name"]][-per * calls:]
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
filt = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "filter" in r["Kernel_Name"])
print(f"kernel us per call {tot / calls / 1e3:.1f}  (filter bank {filt / calls / 1e3:.1f})")
