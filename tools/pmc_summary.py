"""Per-kernel means of the rocprofv3 --pmc passes under <dir>/pmc*/ (one CSV line per kernel).

A counter sampled on several XCDs/SEs shows up as several rows of one dispatch: rows are summed per
(dispatch, counter), then averaged over the dispatches of a kernel.  FETCH_SIZE / WRITE_SIZE are KB.
hbm_read_lo takes FETCH_SIZE as reported, hbm_read_hi doubles it (gfx950 counts the 128-byte requests of
16-byte loads as 64 B; MI355X_MICROARCH.md) -- kernels that mix 4- and 16-byte loads lie in between.
"""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(dict))   # kernel -> counter -> dispatch -> sum
for path in glob.glob(root + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\bzns::", "", r["Kernel_Name"])
        d = vals[k][r["Counter_Name"]]
        key = (path, r["Dispatch_Id"])
        d[key] = d.get(key, 0.0) + float(r["Counter_Value"])
counters = sorted({c for k in vals for c in vals[k]})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "launches"] + counters + ["hbm_read_lo_bytes", "hbm_read_hi_bytes", "hbm_write_bytes",
                                                  "mfma_busy_frac"])
for k in sorted(vals, key=lambda k: -sum(vals[k].get("SQ_WAVE_CYCLES", {0: 0}).values())):
    mean = {c: (sum(vals[k][c].values()) / len(vals[k][c]) if c in vals[k] else float("nan")) for c in counters}
    n = max(len(v) for v in vals[k].values())
    fetch, write = mean.get("FETCH_SIZE", float("nan")), mean.get("WRITE_SIZE", float("nan"))
    busy, gui = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")), mean.get("GRBM_GUI_ACTIVE", float("nan"))
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs can be MFMA-busy per active cycle
    frac = busy / (gui / 8 * 1024) if gui == gui and gui > 0 else float("nan")
    w.writerow([k, n] + [f"{mean[c]:.6g}" for c in counters] +
               [f"{fetch * 1024:.6g}", f"{2 * fetch * 1024:.6g}", f"{write * 1024:.6g}", f"{frac:.3f}"])
