#!/usr/bin/env python3
"""Distance of the two pyramid arithmetics (default float32 FMA chain / ZNS_XQT_EXACT float64 triple) to the numpy restatement
and to each other, over many seeds and the four test lengths: max |log difference| and the linear error relative to the
loudest bin (tests/test_vqt.py::lin_err).  Run on the GPU box: python tools/xqt_accuracy.py [n_seeds]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import vqt_ref
from zeronotesamba_amd import _lib
from zeronotesamba_amd.processing.input_rep import generate_XQT_batch


def lin_err(g, r):
    g, r = np.exp(g.astype(np.float64)), np.exp(r.astype(np.float64))
    return float(np.abs(g - r).max() / r.max())


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rows = []
    for mode in ("vqt", "cqt"):
        for n in (160000, 80000, 33333, 4097):
            if mode == "cqt" and n < 2 ** 15:
                continue
            worst = {"def_vs_ref_log": 0.0, "ex_vs_ref_log": 0.0, "def_vs_ex_log": 0.0, "def_vs_ref_lin": 0.0,
                     "ex_vs_ref_lin": 0.0, "def_vs_ex_lin": 0.0}
            for seed in range(n_seeds):
                rng = np.random.default_rng(1000 * seed + n % 7 + 10)
                y = (0.1 * rng.standard_normal(n)).astype(np.float32)
                yt = torch.from_numpy(y).cuda().reshape(1, -1)
                d = generate_XQT_batch(yt, 16000, mode)[0].cpu().numpy()
                e = generate_XQT_batch(yt, 16000, mode, stages=3 | _lib.ZNS_XQT_EXACT)[0].cpu().numpy()
                r = vqt_ref.generate_XQT(y, 16000, mode)
                for k, (a, b) in {"def_vs_ref": (d, r), "ex_vs_ref": (e, r), "def_vs_ex": (d, e)}.items():
                    worst[k + "_log"] = max(worst[k + "_log"], float(np.abs(a - b).max()))
                    worst[k + "_lin"] = max(worst[k + "_lin"], lin_err(a, b))
            rows.append({"mode": mode, "n": n, "seeds": n_seeds, **worst})
            print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
