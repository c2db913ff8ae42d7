#!/usr/bin/env python3
"""Print the last N kernels of a rocprofv3 --kernel-trace csv: name, start offset, duration (ns), grid, block."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(r["Kernel_Name"][:44], int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
          r["Grid_Size_X"], r["Workgroup_Size_X"])
