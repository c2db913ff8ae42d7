#!/usr/bin/env python3
"""Per-dispatch durations of the LAST pretext step in a rocprofv3 --kernel-trace csv, in launch order:
   rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras
   python3 tools/trace_step.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "zns::" in r["Kernel_Name"]]
# one step = from one cv1_fwd_kernel to the next adam_kernel
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
starts = [i for i, r in enumerate(rows) if "cv1_fwd_kernel" in r["Kernel_Name"]]
s, e = starts[-1], ends[-1]
t0 = int(rows[s]["Start_Timestamp"])
prev_end = t0
for r in rows[s:e + 1]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("zns::", "").replace("void ", "")[:70]
    print(f"t {(st - t0) / 1e3:9.1f}  gap {(st - prev_end) / 1e3:6.1f}  dur {(en - st) / 1e3:8.1f}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>9}  {name}")
    prev_end = en
print("step span us", (int(rows[e]["End_Timestamp"]) - t0) / 1e3)
