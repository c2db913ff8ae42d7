"""bench.py's output contract (the driver parses this line) and its work accounting."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_work_accounting_matches_the_survey():
    sys.path.insert(0, ROOT)
    import bench
    # 2.035e13 FLOP per step in the direct form (fwd + both backward passes, no cv1 dgrad)
    assert abs(bench.STEP_FLOP / 2.035e13 - 1) < 2e-3
    # frequency-domain cv4 forward on 128-point windows: 63 * 3 + 2 = 191 useful real GEMM-convs of 128 x (128*9) per window
    # position, 32 rows x 5 windows = 160 positions, 32 samples
    assert bench.window_length("cv4", 512) == 128 and bench.window_length("cv4", 40) == 64 and bench.window_length("cv6", 512) == 64
    assert bench.spectral_flop("cv4", True) == 2.0 * 191 * 128 * 128 * 9 * 160 * 32
    assert bench.spectral_flop("cv4", False) == 2.0 * 191 * 128 * 128 * 9 * 160 * 32        # (slot 2 is skipped: executed == useful)
    assert bench.spectral_flop("cv6", False) == 2.0 * 95 * 256 * 256 * 5 * 96 * 32          # 64-point windows: 12 per row of 8
    assert bench.spectral_flop("cv7", True) == 2.0 * 95 * 128 * 256 * 13 * 32                # one GEMM per (branch, slot): 13 windows of 42 per row
    assert bench.spectral_flop("cv8", True, 313, 32) == 2.0 * 95 * 128 * 128 * 8 * 32
    assert bench.PROBE["flop"] == bench.spectral_flop("cv4", True)
    assert 2.3e12 < bench.EXEC_FLOP < 2.45e12
    assert bench.host_cores() >= 1


@pytest.mark.gpu
def test_bench_line_has_every_contract_field():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["higher_is_better"] is True
    assert r["scaling"] == "weak" and r["vs_baseline"] is None and r["dtype"] == "f32" and r["data"] == "synthetic"
    assert "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - 1000.0 / r["ms_per_step"]) < 1e-6 * r["value"]
    rf = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["launches_timed"] == 2
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.3 < rf["frac"] < 1.0
    assert 5.0 < r["value"] < 200.0
    # the other BASELINE configs ride in `config`, measured after the timed region
    c = r["config"]
    assert c["module_path_ms_per_step"] > 0 and 0.8 < c["module_path_vs_fused"] < 3.0
    for n in ("clips_32", "clips_4096"):
        x = c["xqt"][n]
        for k in ("ms", "clips_per_s", "algorithmic_GBps", "frac_of_8TBps", "frac_of_fp32_issue_floor", "per_stage_launches_ms",
                  "pyramid_ms", "filter_ms", "filter_frac_of_fp32_peak", "exact_pyramid_plan_ms", "exact_pyramid_ms",
                  "exact_pyramid_frac_of_f64_issue_ceiling"):
            assert x[k] > 0, (n, k)
        assert x["exact_pyramid_frac_of_f64_issue_ceiling"] < 1.0 and x["frac_of_8TBps"] < 1.0 and x["frac_of_fp32_issue_floor"] < 1.0
        assert x["ms"] < x["exact_pyramid_plan_ms"]                    # the default arithmetic is the faster one
    ft = c["finetune"]
    assert ft["frames"] == 1876 and ft["ms_per_step"] > 0 and 0 < ft["executed_frac_of_fp32_mfma_peak"] < 1.0
    # the benchmarked embeddings are not collapsed to a constant (cosines would read 1.0000)
    assert c["mean_cos_anchor_positive"] < 0.9 and abs(c["last_loss"] - c["ln_batch"]) < 0.5


@pytest.mark.gpu
def test_bench_two_ranks_dry_run_on_one_gpu():
    """The N > 1 path of bench.py exactly as the driver launches it (torch.distributed.run, one process per rank,
    rendezvous on 127.0.0.1), with the collective backend switched to gloo so that two ranks can share this box's one
    GPU: process group, bucketed gradient all-reduce inside the step, barrier-bracketed timing, max over ranks, one
    JSON line from rank 0 with the whole-job value."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, ZNS_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["scaling"] == "weak"
    assert r["config"]["global_batch"] == 32 and r["config"]["parallelism"] == "dp2"
    # what SCALE needs to read the overlap from: step time with minus without the gradient all-reduce, and its size
    assert "comm_exposed_ms" in r["config"] and r["config"]["allreduce_bytes"] >= 26819202 * 4
    assert r["config"]["ms_per_step_without_allreduce"] > 0 and r["config"]["allreduce_buckets"] >= 3
    assert "cpu_baseline" not in r
    assert abs(r["value"] - 2 * 1000.0 / r["ms_per_step"]) < 1e-6 * r["value"]      # whole-job steps/s


@pytest.mark.gpu
def test_bench_eight_ranks_dry_run_on_one_gpu():
    """BASELINE configs[4]'s shape of launch -- eight ranks, one process each, as the driver starts them -- on this box's one
    GPU over gloo: eight process-group members, six buckets in flight per step, eight parameter broadcasts, the
    exposed-communication probe, and afterwards eight replicas that stepped on eight different batches and hold bit-identical
    parameters.  A functional dry run of the first 8-GPU run, never a performance number."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, ZNS_BENCH_BACKEND="gloo", ZNS_BENCH_CU_MASK="off")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-extras"],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    c = r["config"]
    assert r["n_gpus"] == 8 and c["global_batch"] == 128 and c["parallelism"] == "dp8" and r["scaling"] == "weak"
    assert c["allreduce_buckets"] == 6 and 26819202 * 4 <= c["allreduce_bytes"] <= 26819202 * 4 + 256
    assert "comm_exposed_ms" in c and c["ms_per_step_without_allreduce"] > 0
    assert c["replicas_identical_after_timed_region"] is True
    assert abs(r["value"] - 8 * 1000.0 / r["ms_per_step"]) < 1e-6 * r["value"]
    for rk in range(8):
        for ph in ("init_process_group(gloo) ok", "trainer built, rank 0's parameters broadcast ok", "timed region (2 steps", "done ok"):
            assert f"[rank {rk}] {ph}" in out.stderr, (rk, ph, out.stderr[-2000:])


@pytest.mark.gpu
def test_bench_self_launches_its_ranks_without_torchrun():
    """`python bench.py --gpus 2` with NO torchrun environment (how a driver may call it): bench.py starts the two rank
    processes itself before touching the GPU and relays rank 0's one JSON line.  gloo backend: both ranks share this
    box's one GPU -- a functional run, not a performance number."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ZNS_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.stdout.strip() == lines[0], out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["config"]["parallelism"] == "dp2"
    assert abs(r["value"] - 2 * 1000.0 / r["ms_per_step"]) < 1e-6 * r["value"]
    # every rank reports its phases on stderr as it passes them (what a hung 8-GPU run would be read from) ...
    for rk in (0, 1):
        for ph in ("import ok", "init_process_group(gloo) ok", "trainer built, rank 0's parameters broadcast ok", "warm-up (1 steps) ok",
                   "timed region (2 steps", "done ok"):
            assert f"[rank {rk}] {ph}" in out.stderr, (rk, ph, out.stderr[-2000:])
    # ... and the warm-up chose the compute stream (plain, or CU-masked so that RCCL's kernels find free CUs) by timing both
    c = r["config"]
    assert c["compute_stream"] in ("plain", "cu-masked (one CU per XCD left to RCCL)")
    assert c.get("cu_mask_error") or (c["warmup_ms_per_step_plain_stream"] > 0 and c["warmup_ms_per_step_cu_masked_stream"] > 0)
    # a rank that dies takes the job down quickly, and the launcher relays every rank's last lines
    env_fail = dict(env, ZNS_BENCH_FAIL_RANK="1", ZNS_BENCH_PG_TIMEOUT="60")
    dead = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                          capture_output=True, text=True, timeout=600, cwd=ROOT, env=env_fail)
    assert dead.returncode != 0 and not [l for l in dead.stdout.splitlines() if l.startswith("{")]
    assert "rank 1 exited with code 7" in dead.stderr and "---- rank 1: last" in dead.stderr and "---- rank 0: last" in dead.stderr
    assert "ZNS_BENCH_FAIL_RANK: exiting with code 7" in dead.stderr.split("---- rank 1: last")[1]
    # and a launch that cannot work says so with a non-zero exit instead of hanging or printing a number
    env.pop("ZNS_BENCH_BACKEND")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_more_ranks_than_gpus_without_a_result_line():
    """CPU side of the same contract: no GPU here, so `--gpus 2` must exit non-zero, quickly, with no JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ZNS_BENCH_BACKEND")}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "{" not in out.stdout


def test_roofline_traffic_constant_is_the_committed_pmc_row():
    """`roofline.traffic` is a constant taken from separate `--pmc` passes (counters cannot ride in the timed run): it
    must equal 2 x FETCH_SIZE + WRITE_SIZE of the probe kernel's row in the profile summary bench.py names, so that a
    new profile set (or a changed dominant kernel) cannot leave a stale number in the bench line."""
    import csv
    sys.path.insert(0, ROOT)
    import bench
    rows = [r for r in csv.DictReader(open(os.path.join(ROOT, bench.PROBE_PMC_FILE))) if bench.PROBE_PMC_KERNEL in r["kernel"]]
    assert len(rows) == 1, [r["kernel"] for r in rows]
    want = (2 * float(rows[0]["FETCH_SIZE"]) + float(rows[0]["WRITE_SIZE"])) * 1024.0
    assert abs(bench.PROBE["traffic"] - want) <= 1e-3 * want
    assert os.path.basename(bench.PROBE_PMC_FILE) in bench.PROBE["traffic_source"]
