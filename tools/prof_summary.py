"""Summarise a rocprofv3 --kernel-trace output directory: per-kernel count / total / avg / max (us)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
paths = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
if not paths:
    sys.exit("no kernel_trace.csv under " + root)
d = collections.defaultdict(list)
for p in paths:
    for r in csv.DictReader(open(p)):
        d[r["Kernel_Name"][:int(sys.argv[2]) if len(sys.argv) > 2 else 60]].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v) / tot * 100:6.2f}%  n={len(v):6d}  total={sum(v):12.1f}us  avg={sum(v) / len(v):10.2f}  max={max(v):10.2f}  {k}")
if len(sys.argv) > 3:   # also: GPU busy time vs span (idle gaps = host-side launch overhead)
    iv = []
    for p in paths:
        for r in csv.DictReader(open(p)):
            iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    iv.sort()
    lo = iv[len(iv) // 3][0]          # skip the warm-up third
    iv = [x for x in iv if x[0] >= lo]
    busy = sum(e - s for s, e in iv)
    span = iv[-1][1] - iv[0][0]
    print(f"# last two thirds: span={span / 1e6:.2f} ms  kernel-busy={busy / 1e6:.2f} ms  idle={100 * (1 - busy / span):.1f}%")
