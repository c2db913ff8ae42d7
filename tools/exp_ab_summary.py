#!/usr/bin/env python3
"""Two prof_summary.py outputs side by side, per kernel group and per kernel: tools/exp_ab_summary.py a.txt b.txt [steps]"""
import re, sys
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r'\s*([\d.]+)%\s+n=\s*(\d+)\s+total=\s*([\d.]+)us\s+avg=\s*([\d.]+)\s+max=\s*([\d.]+)\s+(.*)', l)
        if m: d[m.group(6).strip()] = (int(m.group(2)), float(m.group(3)))
    return d
def group(k):
    if 'conv_core' in k or 'wgrad_core' in k: return 'mfma'
    for g in ('spec_fwd', 'spec_inv', 'pack_freq', 'spec_reduce'):
        if g in k: return g
    return 'other'
a, b = load(sys.argv[1]), load(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ga, gb = {}, {}
for d, g in ((a, ga), (b, gb)):
    for k, (n, t) in d.items(): g[group(k)] = g.get(group(k), 0) + t
print("us per step:", {k: (round(ga.get(k, 0) / steps), round(gb.get(k, 0) / steps)) for k in sorted(set(ga) | set(gb))},
      "total", round(sum(ga.values()) / steps), round(sum(gb.values()) / steps))
short = lambda k: re.sub(r'\(.*', '', k.replace('void ', '').replace('zns::', ''))[:110]
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[1] + b.get(k, (0, 0))[1])):
    na, ta = a.get(k, (0, 0)); nb, tb = b.get(k, (0, 0))
    if ta != 0 and tb != 0 and abs(ta - tb) < 0.02 * ta: continue
    print("%8.0f (n=%3d) %8.0f (n=%3d)  %s" % (ta / steps, na, tb / steps, nb, short(k)))
