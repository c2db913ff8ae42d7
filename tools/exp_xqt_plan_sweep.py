#!/usr/bin/env python3
"""generate_XQT on BASELINE config 2's own batch (32 clips): decimation launch plans (ZNS_TUNE_XQT_PLAN through bench_xqt.py --plan) against the default."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
plans = ["default", "1,1,1,4", "1,1,2,3", "1,2,4", "2,2,3", "1,2,2,2", "2,1,4", "3,4", "4,3", "2,2,2,1", "1,3,3", "2,4,1", "1,1,1,1,1,1,1"]
n = sys.argv[1] if len(sys.argv) > 1 else "32"
for p in plans:
    extra = [] if p == "default" else ["--plan", p.replace(",", "")]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_xqt.py"), "--only", n, "--stages", "7"] + extra, capture_output=True, text=True)
    try:
        r = json.loads(out.stdout.strip().splitlines()[0])
        print(p, "whole", round(r["ms"] * 1e3, 1), "us  pyramid", round(r["pyramid_ms"] * 1e3, 1), "us  filter", round(r["filter_ms"] * 1e3, 1), "us")
    except Exception:
        print(p, "failed", out.stderr[-200:])
