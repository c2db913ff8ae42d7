import os, time, torch, torch.nn.functional as F
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except Exception as e: pass
x = torch.randn(4, 128, 32, 128); w = torch.randn(128, 128, 9, 17)
for nt in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    F.conv2d(x, w, padding=(4, 8))
    t = time.perf_counter(); F.conv2d(x, w, padding=(4, 8)); dt = time.perf_counter() - t
    print(nt, "threads: %.3f s  %.1f GFLOP/s" % (dt, 2 * 4 * 128 * 128 * 153 * 32 * 128 / dt / 1e9))
