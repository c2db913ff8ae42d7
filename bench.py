#!/usr/bin/env python3
"""Pretext-step benchmark (BASELINE.json config 3 / 5): batch=16 pairs of synthetic 1x96x512 VQT tensors per GPU,
twin-branch Pretext_CNN forward + NT-Xent + backward + Adam, dropout active, fp32.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` = pretext steps/sec summed over ranks (each rank steps on its own
16-item batch: weak scaling, no data-path collective, gradient all-reduce over RCCL).  `roofline` is for the
dominant kernel (see DESIGN.md "Measurement"), timed with HIP events recorded by the library on the launch
stream; `cpu_baseline` is the CPU oracle timed on this box's host cores (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

B, T, TAU, LR = 16, 512, 0.25, 1e-6
SR, HOP = 16000, 256
# algorithmic work, SURVEY.md section 8(d): MACs per sample per branch at T=512 (linear in T)
MACS = {"cv1": 1.038e8, "cv2": 1.832e10, "cv3": 1.007e10, "cv4": 4.107e10, "cv5": 7.650e9, "cv6": 2.819e10,
        "cv7": 3.859e8, "cv8": 2.097e8}
STEP_FLOP = B * 2 * 2 * (3 * sum(MACS.values()) - MACS["cv1"])   # fwd + dgrad + wgrad, no cv1 dgrad
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
LAYER_GEOM = {"cv2": (64, 64, 7, 13, 96), "cv3": (64, 128, 5, 15, 32), "cv4": (128, 128, 9, 17, 32),
              "cv5": (128, 256, 3, 19, 8), "cv6": (256, 256, 5, 21, 8)}   # cin, cout, kh, kw, H


def spectral_flop(layer: str, useful: bool) -> float:
    """FLOP of ONE pass (fwd, dgrad or wgrad) of `layer` in the frequency-domain form, both branches, B=16:
    32 frequencies x 3 real GEMMs (Karatsuba) of cout x (cin*kh) per window position (executed); frequency slot 0
    (DC + Nyquist, both real) needs only two of its three (useful).  Tile padding is not counted."""
    cin, cout, kh, kw, H = LAYER_GEOM[layer]
    V = 64 - kw + 1
    P = H * (-(-T // V))
    slots = (31 * 3 + 2) if useful else 32 * 3
    return 2.0 * slots * cout * cin * kh * P * (2 * B)


# MFMA FLOP actually executed per step: cv2..cv6 in the frequency domain (3 passes), cv7/cv8 direct (3 passes)
EXEC_FLOP = 3 * sum(spectral_flop(l, False) for l in LAYER_GEOM) + 3 * 2 * 2 * B * (MACS["cv7"] + MACS["cv8"])
# dominant kernel (profiles/): the frequency-domain forward GEMM-conv of cv4 (conv_core<128,128,1x9 taps, 12-wave HET layout>)
PROBE = dict(kind=0, cin=128, cout=128, kh=9, name="conv_core<spectral cv4 forward: 96 real 128x1152 GEMM-convs per sample, 9 row taps>",
             flop=spectral_flop("cv4", True),
             # HBM bytes per launch from separate rocprofv3 --pmc passes (profiles/r01h_pmc_summary.csv):
             # FETCH_SIZE 608 304 KB x 2 (gfx950 counts the 128-B requests of 16-byte loads as 64 B) + WRITE_SIZE 540 672 KB
             traffic=(2 * 608304 + 540672) * 1024.0)


def synth_batch(seed: int, device) -> torch.Tensor:
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn((B, 2, 96, T), generator=g) * 3.0 - 6.0
    return x.clamp_(-20.72, 3.0).to(device)


def build_model(device):
    from zeronotesamba_amd.models.models import Pretext_CNN
    torch.manual_seed(0)
    m = Pretext_CNN()
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("weight"):
                p.mul_(2.0)   # keep the sigmoid embeddings from collapsing to a constant
    return m.to(device)


def host_cores() -> int:
    """Cores this process may actually use: min(affinity, cgroup CPU quota) -- the GPU box's container is quota-limited."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(max_seconds: float = 30.0):
    """CPU oracle (stock torch CPU ops, numerically the reference: tests/test_oracle.py) on the same workload, timed on
    this box's host cores.  Bounded: a T=64 warm-up step, then full B=16,T=512 steps until ~max_seconds are spent."""
    from oracle import step_ref
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = {k: v.clone() for k, v in build_model("cpu").state_dict().items()}
    op = step_ref.OracleParams(sd, LR)
    batch = synth_batch(0, "cpu")
    step_ref.pretext_step(op, batch[..., :64].contiguous(), B, TAU, drop_seed=1)   # warm-up (oneDNN primitives, threads)
    times = []
    while len(times) < 3 and (sum(times) < max_seconds * 0.6 or not times):
        t0 = time.perf_counter()
        step_ref.pretext_step(op, batch, B, TAU, drop_seed=2 + len(times))
        times.append(time.perf_counter() - t0)
    per = sum(times) / len(times)
    return {"value": 1.0 / per, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} full pretext step(s) (B=16, T=512, dropout on, fwd+NT-Xent+bwd+Adam) after a T=64 warm-up "
                      f"step; torch {torch.__version__} CPU, {cores} threads = container CPU quota",
            "s_per_step": per}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from zeronotesamba_amd import _lib
    from zeronotesamba_amd.distributed import init_from_env
    from zeronotesamba_amd.pretext import PretextTrainer

    # ZNS_BENCH_BACKEND=gloo is a functional dry run of the N > 1 path on a box with fewer GPUs than ranks (ranks then
    # share devices and gradients travel through the host): never a performance number.
    backend = os.environ.get("ZNS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_dev)
    rank, world, local = init_from_env(backend if args.gpus > 1 else None)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    model = build_model(dev)
    trainer = PretextTrainer(model, batch_len=B, temperature=TAU, lr=LR, seed=1234, train=True)
    x = trainer.stack_inputs(synth_batch(100 + rank, dev))   # resident in HBM before the timed region

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(x)
    barrier()

    # events around the dominant kernel, recorded by the library on the launch stream
    K = args.steps
    L = _lib.lib()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    for e in ev0 + ev1:
        e.record()   # materialise the hipEvent_t handles
    torch.cuda.synchronize()
    a0 = _lib.ptr_array([e.cuda_event for e in ev0])
    a1 = _lib.ptr_array([e.cuda_event for e in ev1])
    _lib.check(L.zns_probe_arm(PROBE["kind"], PROBE["cin"], PROBE["cout"], PROBE["kh"], a0, a1, K), "probe_arm")

    barrier()
    t0 = time.perf_counter()
    out = None
    for _ in range(K):
        out = trainer.step(x)
    barrier()
    dt = time.perf_counter() - t0
    nprobe = L.zns_probe_count()
    L.zns_probe_arm(0, 0, 0, 0, None, None, 0)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        kms = [a.elapsed_time(b) for a, b in zip(ev0[:min(K, nprobe)], ev1[:min(K, nprobe)])]
        kavg = sum(kms) / max(len(kms), 1)
        achieved = PROBE["flop"] / (kavg * 1e-3) / 1e12 if kavg > 0 else 0.0
        steps_per_s = world * K / dt
        res = {
            "metric": "pretext steps/sec (batch=16, 1x96x512 VQT)",
            "value": steps_per_s,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: Pretext twin-CNN fwd/bwd + NT-Xent + Adam, batch=16 synthetic "
                                   "1x96x512 VQT pairs per GPU, dropout 0.1 on"
                                   + (f"; configs[4]-style DDP over {world} GPUs (16/GPU), RCCL grad all-reduce" if world > 1 else ""),
                       "batch_per_gpu": B, "frames": T, "global_batch": B * world, "parallelism": f"dp{world}",
                       **({"collective_backend": backend + " (dry run, not a performance number)"} if backend != "nccl" and world > 1 else {}),
                       "audio_seconds_per_s": steps_per_s * B * T * HOP / SR,
                       "direct_conv_tflop_per_step": STEP_FLOP / 1e12,
                       "direct_conv_equivalent_tflops_per_gpu": STEP_FLOP * K / dt / 1e12,
                       "executed_mfma_tflop_per_step": EXEC_FLOP / 1e12,
                       "executed_frac_of_fp32_mfma_peak": (EXEC_FLOP * K / dt) / (FP32_MFMA_PEAK_TFLOPS * 1e12),
                       "last_loss": float(out[0].item())},
            "roofline": {"bound": "mfma", "kernel": PROBE["name"], "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": PROBE["traffic"],
                         "kernel_ms": kavg, "launches_timed": len(kms)},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
