#!/usr/bin/env python3
"""Pretext-step benchmark (BASELINE.json config 3 / 5): batch=16 pairs of synthetic 1x96x512 VQT tensors per GPU,
twin-branch Pretext_CNN forward + NT-Xent + backward + Adam, dropout active, fp32.

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: starts its own N rank processes, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` = pretext steps/sec summed over ranks (each rank steps on its own
16-item batch: weak scaling, no data-path collective, gradient all-reduce over RCCL).  `roofline` is for the
dominant kernel (see DESIGN.md "Measurement"), timed with HIP events recorded by the library on the launch
stream; `cpu_baseline` is the CPU oracle timed on this box's host cores (rank 0, N=1 only).

After the timed region (never inside `ms_per_step`) rank 0 at N=1 also measures, into `config`:
  module_path_ms_per_step   the literal drop-in path: Pretext_CNN autograd + NTXent + torch.optim.Adam (pretext.py:475-492)
  xqt                       BASELINE configs[1]: generate_XQT on 32 and on 4096 resident 10 s clips
  finetune                  BASELINE configs[3]: Down_CNN + BCE + Adam on one 30 s clip, T = 1876 (epochs.py:48-63)
  epoch_path, bank_fill     the path around the step: ShiftBank -> train_epoch at T = 313; stems -> RMS gate -> VQT -> bank
and at N>1: comm_exposed_ms (step time with minus without the gradient all-reduce) and allreduce_bytes.
"""
import argparse
import ctypes
import json
import math
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
              "cv5": (128, 256, 3, 19, 8), "cv6": (256, 256, 5, 21, 8),
              "cv7": (256, 128, 1, 23, 1), "cv8": (128, 128, 1, 25, 1)}   # cin, cout, kh, kw, H


def window_length(layer: str, frames: int) -> int:
    """Overlap-save window of `layer` at this row length, as the library picks it (csrc/zns_conv_fft.hip: use_win128):
    128 samples for cv2 / cv3 / cv4 where that needs fewer frequency-windows per row than 64-sample windows (cv2: rows of
    one segment only), else 64."""
    kw = LAYER_GEOM[layer][3]
    n128 = -(-frames // (129 - kw))
    if layer in ("cv2", "cv3", "cv4") and 2 * n128 < -(-frames // (65 - kw)) and not (layer == "cv2" and n128 > 6):
        return 128
    return 64


def spectral_flop(layer: str, useful: bool, frames: int = T, samples: int = 2 * B) -> float:
    """FLOP of ONE pass (fwd, dgrad or wgrad) of `layer` in the frequency-domain form over `samples` spectrograms of
    `frames` columns (default: both branches, B=16, T=512): FN/2 frequencies x 3 real GEMMs (Karatsuba) of
    cout x (cin*kh) per window position; frequency slot 0 (DC + Nyquist, both real) has only two of the three, and since
    round 5 the launches skip its third item: executed == useful (`useful` is kept for callers; both give the same count).
    Rows longer than 12 (64-point) / 6 (128-point) windows run as NS segments of NB windows; cv7 / cv8 (one image row, kh = 1)
    are one GEMM per (branch, slot) over all samples' windows, never cut.  Tile padding is not counted."""
    cin, cout, kh, kw, H = LAYER_GEOM[layer]
    fn = window_length(layer, frames)
    V = fn - kw + 1
    nbt = -(-frames // V)
    if H == 1 and kh == 1:
        P = nbt
    else:
        ns = -(-nbt // (12 if fn == 64 else 6))
        nb = -(-nbt // ns)
        P = H * nb * ns
    ff = fn // 2
    slots = (ff - 1) * 3 + 2
    return 2.0 * slots * cout * cin * kh * P * samples


def exec_flop(frames: int, samples: int) -> float:
    """MFMA FLOP executed by one training step: cv2..cv8 in the frequency domain (3 passes each; cv1 runs K = 34 GEMMs that
    are bound by its 2 x 201 MB of activations, not counted)."""
    return 3 * sum(spectral_flop(l, False, frames, samples) for l in LAYER_GEOM)


# MFMA FLOP actually executed per step: cv2..cv8 in the frequency domain (3 passes)
EXEC_FLOP = exec_flop(T, 2 * B)
# dominant kernel (profiles/): the frequency-domain forward GEMM-conv of cv4 (conv_core<128,128,1x9 taps>: 128-point windows,
# 192 slots per sample, four waves of (1 channel tile x 5 column tiles))
PROBE = dict(kind=0, cin=128, cout=128, kh=9, name="conv_core<spectral cv4 forward: 192 real 128x1152 GEMM-convs per sample (128-point windows), 9 row taps>",
             flop=spectral_flop("cv4", True),
             # HBM bytes per launch from separate rocprofv3 --pmc passes, NOT measured by this run: 2 x FETCH_SIZE (gfx950
             # counts the 128-B requests of 16-byte loads as 64 B) + WRITE_SIZE of this kernel's row in PROBE_PMC_FILE.
             # tests/test_bench_contract.py fails when the two drift apart (a new profile set must update both).
             # (1.227 GB against 1.226 GB algorithmic: 501 MB of input spectrum + 501 MB of output + 225 MB of operators -- 191 of
             # the 192 slots: frequency slot 0 has no third Karatsuba item)
             traffic=(2 * 354613 + 488960) * 1024.0,
             traffic_source="constant from separate rocprofv3 --pmc passes (profiles/r05_pmc_summary.csv), 2 x FETCH_SIZE + WRITE_SIZE; not measured by this run")
PROBE_PMC_FILE = "profiles/r05_pmc_summary.csv"
PROBE_PMC_KERNEL = "conv_core<ConvCfg<128, 128, 1, 9, 1, 1, 1, 5, 4, 1, 1, 4,"      # cv4 forward / backward-data on 128-point windows: 4 waves x (1 x 5)


def _lowbias32(x):
    import numpy as np
    x = x.astype(np.uint32, copy=True)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def uniform_pm1(seed: int, n: int):
    """Exactly reproducible U(-1, 1) float32 stream from an integer hash: the same closed-form generator the golden
    fixtures use (tests/golden/make_golden.py via the oracle), re-implemented here so bench.py's product path imports
    nothing from oracle/."""
    import numpy as np
    with np.errstate(over="ignore"):
        h = _lowbias32((np.arange(n, dtype=np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32) * np.uint32(0x9E3779B1)
                       ^ np.uint32(seed & 0xFFFFFFFF))
    return ((h >> np.uint32(8)).astype(np.float64) / 8388608.0 - 1.0).astype(np.float32)


def synth_state_dict(model: torch.nn.Module, seed: int = 5, gain: float = 2.0, head_gain: float = 2.0, head_bias: float = -5.0):
    """Closed-form random init that keeps the embeddings from collapsing: weights U(-b, b) with
    b = gain / sqrt(fan_in), biases U(-0.05, 0.05); the sigmoid head gets a negative bias and a larger gain so the
    pulse embedding is sparse (mean ~0.08) instead of a constant ~0.7 whose cosine to anything is 1."""
    import numpy as np
    sd = {}
    for k, (name, ref) in enumerate(model.state_dict().items()):
        shape = tuple(ref.shape)
        n = int(np.prod(shape))
        u = uniform_pm1(seed * 1000 + k, n)
        if name.endswith("bias"):
            v = u * np.float32(0.05)
            if ".fc1." in name or name.startswith("fc1."):
                v = v + np.float32(head_bias)
        else:
            v = u * np.float32(gain / np.sqrt(int(np.prod(shape[1:]))))
            if ".fc1." in name or name.startswith("fc1."):
                v = v * np.float32(head_gain)
        sd[name] = torch.from_numpy(v.reshape(shape).copy())
    return sd


def synth_batch(seed: int, device, frames: int = None) -> torch.Tensor:
    """One pretext batch as pretext.py:308-318 builds it: 16 DISTINCT time shifts of ONE clip.  The clip is a synthetic
    VQT-like bank row (log-magnitudes in [-20.72, 3]); the positive stem shares 70 % of the anchor's content (drums vs the
    rest of the same music).  [B, 2, 96, T]."""
    T_ = T if frames is None else frames
    g = torch.Generator(device="cpu").manual_seed(seed)
    base = torch.randn((96, 2 * T_), generator=g) * 3.0 - 6.0
    other = torch.randn((96, 2 * T_), generator=g) * 3.0 - 6.0
    pos = 0.7 * base + 0.3 * other
    starts = torch.randperm(T_, generator=g)[:B].tolist()
    x = torch.stack([torch.stack([base[:, s:s + T_], pos[:, s:s + T_]]) for s in starts])
    return x.clamp_(-20.72, 3.0).to(device)


def build_model(device):
    from zeronotesamba_amd.models.models import Pretext_CNN
    m = Pretext_CNN()
    m.load_state_dict(synth_state_dict(m))
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
    batch = synth_batch(100, "cpu")
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


def _timed(fn, iters: int, warm: int = 2) -> float:
    """ms per call of fn() on the current stream, wall clock bracketed by synchronize (inputs already resident)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def measure_module_path(dev, iters: int = 10) -> dict:
    """The literal drop-in path INTEGRATION.md section 1 gives a user who only switches imports: the reference's own
    step structure (pretext.py:475-492) -- Pretext_CNN under autograd, NTXent, stock torch.optim.Adam over the 36
    parameter tensors -- on the same model, batch and dropout as the fused PretextTrainer the headline times."""
    from zeronotesamba_amd.models.loss_functions import NTXent
    model = build_model(dev).train()
    crit = NTXent(B, TAU)
    opt = torch.optim.Adam(model.parameters(), lr=LR)
    batch = synth_batch(100, dev)
    anchors, postves = batch[:, 0:1].contiguous(), batch[:, 1:2].contiguous()
    torch.manual_seed(0)

    def step():
        opt.zero_grad()
        a, p = model(anchors, postves)
        loss, _, _ = crit(a, p)          # includes the reference's device->host copy of the two cosine means
        loss.backward()
        opt.step()

    out = {"module_path_ms_per_step": _timed(step, iters, 3)}
    # the validation step of pretext.val_epoch (:527-592): eval mode, no_grad, same batch -- nothing is saved for a
    # backward pass and the packed frequency weights are reused from call to call
    model.eval()

    def val_step():
        with torch.no_grad():
            a, p = model(anchors, postves)
            crit(a, p)

    import zeronotesamba_amd
    with zeronotesamba_amd.frozen_weights():      # as val_epoch holds it around its loop: weight operands packed once
        out["val_batch_ms"] = _timed(val_step, iters, 3)
    return out


XQT_SAMPLES = 160000                     # 10 s at 16 kHz (BASELINE configs[1])
XQT_BYTES_PER_CLIP = XQT_SAMPLES * 4 + 96 * (1 + XQT_SAMPLES // 256) * 4      # 640 KB in + 240 KB out
XQT_TAPS_PER_CLIP = 63 * sum(-(-XQT_SAMPLES // (1 << i)) for i in range(1, 8))   # decimation: 63 taps per pyramid sample
XQT_FILTER_FLOP_PER_CLIP = 2 * 2 * 12 * (1 + XQT_SAMPLES // 256) * (128 * 4 + 64 + 32 + 32 + 16)   # complex FIR bank, VQT
F64_TAP_CEILING = 12.7e12                # taps/s/chip of the EXACT pyramid: fma_f64 + cvt_f32_f64 + cvt_f64_f32 = 12.35 cycles per wave (tools/ubench_f64.hip)
VALU_CYCLES_PER_INSTR = 4.2              # any vector instruction, one wave64 per SIMD (tools/ubench_valu.hip: v_fma_f32 75 TFLOP/s, v_pk_fma_f32 147)


def measure_xqt(dev, cpu: bool) -> dict:
    """BASELINE configs[1]: generate_XQT on 32 resident clips and on a bandwidth-saturating 4096.  The default
    plan for vqt is ONE fused launch (pyramid in LDS, HBM traffic = clip in + spectrogram out); beside it the per-stage
    launches of the same arithmetic (pyramid / filter bank timed alone) and the exact validation pyramid.  Bounds: the whole
    transform against the 8 TB/s HBM roofline (algorithmic bytes), and against the fp32 instruction-issue floor of its own
    arithmetic -- the vector FMAs of the pyramid and the matrix-pipe FMAs of the filter bank share one issue port per SIMD."""
    from zeronotesamba_amd import _lib
    from zeronotesamba_amd.processing.input_rep import generate_XQT_batch
    out = {}
    # packed taps: 67 vector instructions per output pair (59 v_pk_fma_f32 + 8 v_fma_f32) instead of 126; filter bank: one
    # v_mfma_f32_16x16x4_f32 (32 cycles) per 4 taps x 16 frames x 16 rows, two row tiles for the 24 real rows of an octave
    issue_cycles_per_clip = (XQT_TAPS_PER_CLIP / 63.0 / 2.0) * 67 / 64.0 * VALU_CYCLES_PER_INSTR \
        + (1 + XQT_SAMPLES // 256) / 16.0 * 2 * (128 * 4 + 64 + 32 + 32 + 16) / 4.0 * 32.0
    issue_floor_s_per_clip = issue_cycles_per_clip / (1024 * 2.4e9)          # 1024 SIMDs at the 2.4 GHz peak clock
    for n_clips, iters in ((32, 50), (4096, 3)):
        g = torch.Generator(device="cpu").manual_seed(1234)
        y = (0.1 * torch.randn((min(n_clips, 64), XQT_SAMPLES), generator=g)).to(dev)
        if n_clips > y.shape[0]:
            y = y.repeat(n_clips // y.shape[0], 1).contiguous()
        res = torch.empty((n_clips, 96, 1 + XQT_SAMPLES // 256), dtype=torch.float32, device=dev)
        ms = _timed(lambda: generate_XQT_batch(y, SR, "vqt", out=res), iters)
        ms_sep = _timed(lambda: generate_XQT_batch(y, SR, "vqt", out=res, stages=3 | _lib.ZNS_XQT_SEPARATE), iters)
        ms_pyr = _timed(lambda: generate_XQT_batch(y, SR, "vqt", out=res, stages=1), iters)
        ms_fir = _timed(lambda: generate_XQT_batch(y, SR, "vqt", out=res, stages=2), iters)
        ms_exact = _timed(lambda: generate_XQT_batch(y, SR, "vqt", out=res, stages=3 | _lib.ZNS_XQT_EXACT), iters)
        ms_pyr_exact = _timed(lambda: generate_XQT_batch(y, SR, "vqt", out=res, stages=1 | _lib.ZNS_XQT_EXACT), iters)
        gbps = n_clips * XQT_BYTES_PER_CLIP / (ms * 1e-3) / 1e9
        out[f"clips_{n_clips}"] = {
            "ms": ms, "plan": "fused (one launch, pyramid in LDS)" if n_clips >= 192 else "per-stage launches (three pyramid chains + filter bank)",
            "clips_per_s": n_clips / (ms * 1e-3),
            "algorithmic_GBps": gbps, "frac_of_8TBps": gbps / 8000.0,
            "frac_of_fp32_issue_floor": n_clips * issue_floor_s_per_clip / (ms * 1e-3),
            "per_stage_launches_ms": ms_sep, "pyramid_ms": ms_pyr, "filter_ms": ms_fir,
            "filter_frac_of_fp32_peak": n_clips * XQT_FILTER_FLOP_PER_CLIP / (ms_fir * 1e-3) / (FP32_MFMA_PEAK_TFLOPS * 1e12),
            "exact_pyramid_plan_ms": ms_exact, "exact_pyramid_ms": ms_pyr_exact,
            "exact_pyramid_frac_of_f64_issue_ceiling": n_clips * XQT_TAPS_PER_CLIP / (ms_pyr_exact * 1e-3) / F64_TAP_CEILING}
        del y, res
    out["bound"] = ("hbm by SURVEY 8(d) (0.88 MB algorithmic per clip); the fused kernel moves just those bytes (profiles/r04_xqt_*) and "
                    "is bound by fp32 instruction issue: 1.0e7 decimation taps + 1.3e7 matrix-pipe multiply-adds per clip on one issue "
                    "port per SIMD (issue floor = frac_of_fp32_issue_floor), see DESIGN.md section 5")
    if cpu:
        import numpy as np
        from oracle import vqt_ref
        rng = np.random.default_rng(1234)
        yc = (0.1 * rng.standard_normal((3, XQT_SAMPLES))).astype(np.float32)
        vqt_ref.generate_XQT(yc[0], SR, "vqt")                       # builds and caches the filter bank
        t0 = time.perf_counter()
        for i in (1, 2):
            vqt_ref.generate_XQT(yc[i], SR, "vqt")
        out["cpu_oracle_s_per_clip"] = (time.perf_counter() - t0) / 2
        out["cpu_oracle_sample"] = "2 clips, numpy restatement of librosa/resampy, 1 process, per-clip loop as the reference"
    return out


FT_T = 1876                              # 30 s Ballroom-like clip


def measure_finetune(dev, cpu: bool, iters: int = 10) -> dict:
    """BASELINE configs[3] (epochs.py:48-63 through loader.load_models("pretrained", "finetune")): batch of one 30 s clip,
    Down_CNN(max) forward -> BCELoss vs a 120-BPM pulse mask -> backward -> Adam(lr = 0.5 * 1e-5 * 10e-2)."""
    from zeronotesamba_amd.loader import load_models
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        crit, opt, model = load_models("pretrained", "finetune", 1e-5, state_dict=build_model("cpu").state_dict(),
                                       device=str(dev))
    model.train()
    g = torch.Generator(device="cpu").manual_seed(1)
    vqt = (torch.randn((2, 96, FT_T), generator=g) * 3.0 - 6.0).clamp_(-20.72, 3.0)
    msk = torch.zeros((1, FT_T))
    msk[0, 5::31] = 1.0
    msk[0, 4::31] = 0.5
    msk[0, 6::31] = 0.5
    v1, v2, m = vqt[0].reshape(1, 1, 96, FT_T).to(dev), vqt[1].reshape(1, 1, 96, FT_T).to(dev), msk.to(dev)
    torch.manual_seed(0)

    def step():
        opt.zero_grad()
        loss = crit(model(v1, v2), m)
        loss.backward()
        opt.step()

    ms = _timed(step, iters, 3)
    flop = exec_flop(FT_T, 2)
    out = {"frames": FT_T, "ms_per_step": ms, "steps_per_s": 1e3 / ms,
           "direct_conv_tflop_per_step": 2 * 2 * (3 * sum(MACS.values()) - MACS["cv1"]) * FT_T / 512.0 / 1e12,
           "executed_mfma_tflop_per_step": flop / 1e12,
           "executed_frac_of_fp32_mfma_peak": flop / (ms * 1e-3) / (FP32_MFMA_PEAK_TFLOPS * 1e12)}
    if cpu:
        from oracle import step_ref
        torch.set_num_threads(host_cores())
        sd = {"pretext." + k: v.clone() for k, v in build_model("cpu").state_dict().items()}
        op = step_ref.OracleParams(sd, 5e-7)
        step_ref.finetune_step(op, vqt[..., :64].contiguous(), msk[0, :64], drop_seed=1)     # warm-up
        t0 = time.perf_counter()
        step_ref.finetune_step(op, vqt, msk[0], drop_seed=2)
        out["cpu_oracle_s_per_step"] = time.perf_counter() - t0
        out["cpu_oracle_sample"] = f"1 full fine-tune step after a T=64 warm-up; {host_cores()} threads"
    return out


def cu_masked_stream(dev, reserve_per_xcd: int = 1):
    """A HIP stream whose kernels may not use `reserve_per_xcd` CUs of every XCD (hipExtStreamCreateWithCUMask: one bit per
    CU, 32-bit words), wrapped so that torch -- and with it every launch of the trainer, which enqueues on torch's current
    stream -- can use it.  Mask bit n is a CU of XCD n % 8 (the driver deals the bits round-robin over the XCDs), MEASURED on
    the box with tools/ubench_cu_mask.hip (profiles/r05_ubench_cu_mask.txt): clearing bits 0..7 leaves 31 CUs in every XCD;
    clearing bits 31, 63, .. 255 -- what round 4 did, believing bits [32 i, 32 i + 32) to be XCD i -- takes eight CUs from
    XCD 7 alone."""
    import ctypes
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    n_xcd = 8
    words = [0xFFFFFFFF] * ((ncu + 31) // 32)
    for k in range(reserve_per_xcd):
        for xcd in range(n_xcd):
            bit = k * n_xcd + xcd                    # the k-th CU of this XCD
            words[bit // 32] &= ~(1 << (bit % 32))
    if ncu % 32:
        words[-1] &= (1 << (ncu % 32)) - 1
    hip = ctypes.CDLL("libamdhip64.so")
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    with torch.cuda.device(dev):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    if rc != 0 or not s.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(s.value, device=dev)


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without torchrun's environment: start N FRESH rank processes (this file, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would) and relay rank 0's JSON line.
    The parent makes no HIP call before or after (counting devices does not initialise the GPU on this image), never
    exec()s, and kills by exact PID only.  Returns the exit code: non-zero if any rank failed."""
    import socket
    import subprocess
    backend = os.environ.get("ZNS_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n:
        print(f"bench.py: --gpus {n} but this node exposes {have} GPU(s); RCCL needs one device per rank "
              "(ZNS_BENCH_BACKEND=gloo runs a functional dry run with ranks sharing devices)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import collections
    import threading
    procs, tails, readers = [], [], []
    found = []

    def pump(r, stream, is_out):
        """Pass a rank's output through to our stderr as it comes and keep its last 50 lines (rank 0's stdout: the one
        JSON line is held back for our stdout)."""
        for raw in stream:
            if is_out and raw.startswith("{") and not found:
                found.append(raw.rstrip("\n"))
                continue
            tails[r].append(raw.rstrip("\n"))
            sys.stderr.write(raw if raw.startswith("[rank") else f"[rank {r}] {raw}")
            sys.stderr.flush()

    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL's intra-node transport on this pool
        env.setdefault("OMP_NUM_THREADS", "1")                # as torch.distributed.run sets it
        # rank 0's stdout carries the one JSON line; everything else a rank prints is diagnostics
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, bufsize=1)
        procs.append(p)
        tails.append(collections.deque(maxlen=50))
        for stream, is_out in ((p.stdout, r == 0), (p.stderr, False)):
            th = threading.Thread(target=pump, args=(r, stream, is_out), daemon=True)
            th.start()
            readers.append(th)
    rc, failed, timed_out = 0, None, False
    try:
        # the driver gives the whole command 1800 s: fail well inside that, with the ranks' own words on stderr
        deadline = time.time() + float(os.environ.get("ZNS_BENCH_LAUNCH_TIMEOUT", "900"))
        pending = list(procs)
        while pending and time.time() < deadline:
            for p in list(pending):
                c = p.poll()
                if c is not None:
                    pending.remove(p)
                    if c and not rc:
                        rc, failed = c, procs.index(p)
            if rc:                                             # one rank died: its peers would wait in a collective
                break
            time.sleep(0.05)
        if pending and not rc:
            rc, timed_out = 1, True
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    for th in readers:
        th.join(timeout=10)
    line = found[0] if found else None
    if rc == 0 and line is None:
        rc = 1
    if rc == 0:
        print(line, flush=True)
    else:
        why = "timed out (ZNS_BENCH_LAUNCH_TIMEOUT)" if timed_out else f"rank {failed} exited with code {rc}" if failed is not None else "no result line from rank 0"
        print(f"bench.py: {why}; the last lines of every rank follow (each rank reports its phases as it passes them)", file=sys.stderr)
        for r in range(n):
            print(f"---- rank {r}: last {len(tails[r])} lines ----", file=sys.stderr)
            for t in tails[r]:
                print(f"    {t}", file=sys.stderr)
    return rc


def measure_pipeline(dev) -> dict:
    """The path AROUND the step, after the timed region:
    epoch_path   pretext.py:308-330 at the reference's own tile: a resident bank of 256 clips x 626 frames, ShiftBank
                 draws 16 start frames per clip and crops 313-frame batches on the device, PretextTrainer.train_epoch
                 steps through them (one host sync at the end of the epoch) -- next to the same step on ONE resident
                 T = 313 batch, i.e. what the kernels alone take.
    bank_fill    pretext.py:143-170 for stems already in HBM: 256 files x 4 stems x 12 s -> mix -> RMS-gated random
                 10 s crop (config.yaml: lower_p .3, upper_p 1.) -> VQT pair -> resident bank; files/s, next to the VQT
                 kernels alone on the same 512 clips."""
    import random
    from zeronotesamba_amd.pretext import PretextTrainer, ShiftBank, fill_memory_bank
    from zeronotesamba_amd.processing.input_rep import generate_XQT_batch
    out = {}
    n_clips, crop = 256, 313
    g = torch.Generator(device=dev).manual_seed(7)
    bank = (torch.randn((n_clips, 2, 96, 2 * crop), generator=g, device=dev) * 3.0 - 6.0).clamp_(-20.72, 3.0)
    bank[:, 1] = 0.7 * bank[:, 0] + 0.3 * bank[:, 1]
    sb = ShiftBank(bank, batch_len=B, crop=crop, device=str(dev))
    trainer = PretextTrainer(build_model(dev), batch_len=B, temperature=TAU, lr=LR, seed=1234, train=True)
    rng = random.Random(0)
    trainer.train_epoch(sb.batches(range(4), rng=rng))                       # warm-up
    torch.cuda.synchronize()
    steps = 96
    t0 = time.perf_counter()
    trainer.train_epoch(sb.batches(range(steps), rng=rng))                   # ends with the epoch's one host sync
    ep = (time.perf_counter() - t0) / steps * 1e3
    x = sb.batch(0, rng=rng)
    res = _timed(lambda: trainer.step(x), steps, 3)
    out["epoch_path"] = {"frames": crop, "bank_clips": n_clips, "steps": steps, "ms_per_step": ep,
                         "resident_batch_ms_per_step": res, "vs_resident_batch": ep / res,
                         "us_per_frame_pair": ep * 1e3 / (B * crop),
                         "executed_frac_of_fp32_mfma_peak": exec_flop(crop, 2 * B) / (ep * 1e-3) / (FP32_MFMA_PEAK_TFLOPS * 1e12),
                         "audio_seconds_per_s": 1e3 / ep * B * crop * HOP / SR}
    del trainer, sb, bank, x
    torch.cuda.empty_cache()
    # ---- bank fill
    n_files, n = 256, 12 * SR
    g = torch.Generator(device=dev).manual_seed(11)
    drums = 0.1 * torch.randn((n_files, n), generator=g, device=dev)
    t = torch.arange(n, device=dev) % (5 * SR)
    gain = torch.where(t < 2 * SR, 0.6, 0.1)                                  # ~40 % of the frames pass the ratio test
    stems = {}
    for i in range(n_files):
        bass = drums[i] * gain * 0.5
        stems[f"f{i}"] = {"bass": bass, "drums": drums[i], "other": bass, "vocals": 0.01 * drums[(i + 1) % n_files]}
    fps = [f"f{i}\n" for i in range(n_files)]
    yml = {"clip_len": 10, "input_mode": "vqt", "lower_p": 0.3, "upper_p": 1.0}
    fill_memory_bank(stems, fps[:64], yml, rng=random.Random(1))             # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fill_memory_bank(stems, fps, yml, rng=random.Random(2))
    torch.cuda.synchronize()
    fill_s = time.perf_counter() - t0
    crops = torch.empty((2 * n_files, 10 * SR), dtype=torch.float32, device=dev).normal_(0.0, 0.1, generator=g)
    res = torch.empty((2 * n_files, 96, 1 + 10 * SR // 256), dtype=torch.float32, device=dev)
    xq = _timed(lambda: generate_XQT_batch(crops, SR, "vqt", out=res), 5)
    out["bank_fill"] = {"files": n_files, "stems_per_file": 4, "seconds_per_stem": 12, "files_per_s": n_files / fill_s,
                        "ms_total": fill_s * 1e3, "xqt_kernels_alone_ms": xq, "vs_xqt_kernels_alone": fill_s * 1e3 / xq,
                        "audio_seconds_per_s": n_files * 2 * 10 / fill_s}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the module-path / generate_XQT / fine-tune measurements")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    t_start = time.time()
    env_rank = int(os.environ.get("RANK", "0"))

    def phase(what):
        """One stderr line per phase and rank: when a multi-GPU run hangs or dies, the launcher's relay of each rank's
        last lines shows which rank stopped where."""
        if args.gpus > 1:
            print(f"[rank {env_rank}] {what} ok (+{time.time() - t_start:.1f} s)", file=sys.stderr, flush=True)

    from zeronotesamba_amd import _lib
    from zeronotesamba_amd.distributed import init_from_env
    from zeronotesamba_amd.pretext import PretextTrainer
    phase("import")

    # ZNS_BENCH_BACKEND=gloo is a functional dry run of the N > 1 path on a box with fewer GPUs than ranks (ranks then
    # share devices and gradients travel through the host): never a performance number.
    backend = os.environ.get("ZNS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_dev)
    rank, world, local = init_from_env(backend if args.gpus > 1 else None,
                                       timeout_s=float(os.environ.get("ZNS_BENCH_PG_TIMEOUT", "300")))
    phase(f"init_process_group({backend})")
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    model = build_model(dev)
    phase("model built")
    if os.environ.get("ZNS_BENCH_FAIL_RANK") == str(env_rank):      # test hook: a rank that dies before its first collective
        print(f"[rank {env_rank}] ZNS_BENCH_FAIL_RANK: exiting with code 7", file=sys.stderr, flush=True)
        sys.exit(7)
    trainer = PretextTrainer(model, batch_len=B, temperature=TAU, lr=LR, seed=1234, train=True, data_parallel=world > 1)
    phase("trainer built, rank 0's parameters broadcast")
    x = trainer.stack_inputs(synth_batch(100 + rank, dev))   # resident in HBM before the timed region

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # N > 1: which stream should the step's kernels run on?  The conv workgroups take whole CUs, so RCCL's reduction
    # kernels (on RCCL's own stream, which torch gives no handle on) only get CUs as a launch drains.
    # The knob is on the COMPUTE side: a stream whose CU mask leaves one CU per XCD free.  Time three steps each way in the
    # warm-up, keep the faster (the max over ranks decides, so every rank picks the same), say which in `config`.
    compute_stream, stream_choice = None, {}
    if world > 1 and os.environ.get("ZNS_BENCH_CU_MASK", "auto") != "off":
        try:
            masked = cu_masked_stream(dev, reserve_per_xcd=1)
        except Exception as e:                                # noqa: BLE001 -- an optional knob: report, do not fail the run
            masked, stream_choice = None, {"compute_stream": "plain", "cu_mask_error": repr(e)[:200]}
        if masked is not None:
            def timed3(stream):
                with torch.cuda.stream(stream):
                    trainer.step(x)
                    barrier()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        trainer.step(x)
                    barrier()
                    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                return float(t.item()) / 3 * 1e3
            ms_plain = timed3(torch.cuda.current_stream())
            ms_masked = timed3(masked)
            use_masked = ms_masked < ms_plain or os.environ.get("ZNS_BENCH_CU_MASK") == "on"
            compute_stream = masked if use_masked else None
            stream_choice = {"compute_stream": "cu-masked (one CU per XCD left to RCCL)" if use_masked else "plain",
                             "warmup_ms_per_step_plain_stream": ms_plain, "warmup_ms_per_step_cu_masked_stream": ms_masked}
        phase(f"compute stream chosen: {stream_choice.get('compute_stream')}")
    import contextlib
    on_stream = (lambda: torch.cuda.stream(compute_stream)) if compute_stream is not None else contextlib.nullcontext

    with on_stream():
        for _ in range(args.warmup):
            trainer.step(x)
        barrier()
    phase(f"warm-up ({args.warmup} steps)")

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

    with on_stream():
        barrier()
        t0 = time.perf_counter()
        out = None
        for _ in range(K):
            out = trainer.step(x)
        barrier()
        dt = time.perf_counter() - t0
    nprobe = L.zns_probe_count()
    L.zns_probe_arm(0, 0, 0, 0, None, None, 0)
    phase(f"timed region ({K} steps, {dt / K * 1e3:.2f} ms per step on this rank)")

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())

    comm = {}
    replicas_identical = None
    if world > 1:
        # every rank stepped on its OWN batch: equal parameters afterwards mean the gradient all-reduce (and the initial
        # broadcast) did their job.  Bit patterns summed as integers, gathered, compared on rank 0 -- before the probe
        # below lets the replicas drift.
        csum = trainer.flat_p.view(torch.int32).to(torch.int64).sum().reshape(1)
        sums = [torch.zeros_like(csum) for _ in range(world)]
        torch.distributed.all_gather(sums, csum)
        replicas_identical = all(int(s.item()) == int(sums[0].item()) for s in sums)
        # exposed communication: the same K steps with the gradient all-reduce switched off (replicas drift apart, which
        # is irrelevant after the timed region); exposed = with - without.  Every rank runs it: same barriers.
        reducer, trainer.reducer = trainer.reducer, None
        with on_stream():
            barrier()
            t1 = time.perf_counter()
            for _ in range(K):
                trainer.step(x)
            barrier()
            dt_nc = time.perf_counter() - t1
        trainer.reducer = reducer
        tnc = torch.tensor([dt_nc], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tnc, op=torch.distributed.ReduceOp.MAX)
        dt_nc = float(tnc.item())
        comm = {"comm_exposed_ms": (dt - dt_nc) / K * 1e3, "ms_per_step_without_allreduce": dt_nc / K * 1e3,
                "allreduce_bytes": trainer.n_flat * 4, "allreduce_buckets": len(trainer.buckets),
                "replicas_identical_after_timed_region": replicas_identical, **stream_choice}
        phase("exposed-communication probe")

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
                       "last_loss": float(out[0].item()), "ln_batch": math.log(B),
                       "mean_cos_anchor_positive": float(out[1].item()), "mean_cos_anchor_negative": float(out[2].item()),
                       "init": "closed-form hash init (the golden fixtures' generator), sparse sigmoid head; independent "
                               "anchor / positive branches, so NT-Xent starts near ln(batch) by construction -- the "
                               "cosines (not 1.0000) show the embeddings are not collapsed",
                       **comm},
            "roofline": {"bound": "mfma", "kernel": PROBE["name"], "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": PROBE["traffic"],
                         "traffic_source": PROBE["traffic_source"], "kernel_ms": kavg, "launches_timed": len(kms)},
        }
        # everything below runs AFTER the timed region; if any of it raises, the headline line is still printed
        try:
            if world == 1 and not args.no_extras:
                del trainer, model
                torch.cuda.empty_cache()
                cpu = not args.no_cpu_baseline
                res["config"].update(measure_module_path(dev))
                res["config"]["module_path_vs_fused"] = res["config"]["module_path_ms_per_step"] / res["ms_per_step"]
                res["config"]["xqt"] = measure_xqt(dev, cpu)
                res["config"]["finetune"] = measure_finetune(dev, cpu)
                res["config"].update(measure_pipeline(dev))
            if world == 1 and not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline()
                res["cpu_baseline"]["sample"] += ("; GPU-utilisation samplers see little on this command: the timed GPU region is "
                                                  "about a second inside tens of seconds of CPU baseline")
        except Exception as e:                                # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            res["config"]["extras_error"] = repr(e)[:300]
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    phase("done")


if __name__ == "__main__":
    main()
