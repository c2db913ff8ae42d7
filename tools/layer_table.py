#!/usr/bin/env python3
"""Per-kernel table of a rocprofv3 kernel_stats.csv of the fused step: launches per step, average time, useful FLOP and
fraction of the fp32-MFMA peak for the GEMM kernels, GB/s of algorithmic traffic is NOT derived (transforms: see DESIGN).
    python tools/layer_table.py gpurun_out/prof_r05/kernel_stats.csv <steps> [frames] [samples]"""
import csv, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
path, steps = sys.argv[1], int(sys.argv[2])
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 512
samples = int(sys.argv[4]) if len(sys.argv) > 4 else 32
PEAK = bench.FP32_MFMA_PEAK_TFLOPS * 1e12
geo = {(v[0], v[1], v[2]): k for k, v in bench.LAYER_GEOM.items()}          # (cin, cout, kh) -> layer
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step: {tot / steps / 1e6:.2f} ms")
out = []
for r in rows:
    name = r["Name"].replace("zns::", "").replace("void ", "")
    avg = float(r["AverageNs"]) / 1e3
    per_step = int(r["Calls"]) / steps
    note, frac = "", ""
    m = re.match(r"(conv_core|wgrad_core)<(?:ConvCfg|WgCfg)<(\d+), (\d+), (\d+), (\d+), (\d+),(.*?)> >", name)
    if m:
        kind, a, b, kh_, kwt, H = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6))
        if kh_ == 1 and H == 1 and kwt < 23:       # spectral: taps = the layer's kernel rows
            lay = geo.get((a, b, kwt)) or geo.get((b, a, kwt))
            which = "backward-weights" if kind == "wgrad_core" else ("forward" if geo.get((a, b, kwt)) == lay and (a != b or True) else "backward-data")
            if kind == "conv_core" and a != b:
                which = "forward" if geo.get((a, b, kwt)) else "backward-data"
            elif kind == "conv_core":
                which = "forward / backward-data"
            fl = bench.spectral_flop(lay, True, frames, samples)
            frac = f"{fl / (avg * 1e-6) / PEAK:.3f}"
            note = f"{lay} {which}, {bench.window_length(lay, frames)}-point windows"
        else:
            note = "direct cv7 / cv8" + (" (K-split part)" if a in (16, 32) else "")
    out.append((float(r["TotalDurationNs"]) / steps / 1e3, per_step, avg, frac, note, name[:100]))
for t, n, avg, frac, note, name in sorted(out, reverse=True)[:45]:
    print(f"{t:8.0f} us/step  x{n:4.1f}  avg {avg:8.1f} us  {frac:>6s}  {note:55s} {name}")
