import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from zeronotesamba_amd.pretext import PretextTrainer
dev = torch.device("cuda", 0)
tr = PretextTrainer(bench.build_model(dev), batch_len=16, lr=float(sys.argv[1]) if len(sys.argv) > 1 else 1e-6, seed=5)
xs = [tr.stack_inputs(bench.synth_batch(i, dev)) for i in range(4)]
torch.cuda.reset_peak_memory_stats()
losses = []
t0 = time.perf_counter()
for i in range(300):
    out = tr.step(xs[i % 4])
    if i % 50 == 0:
        losses.append(out.tolist())
torch.cuda.synchronize()
print("300 steps in", time.perf_counter() - t0, "s; peak mem GB", torch.cuda.max_memory_allocated() / 2**30)
print(losses)
print("finite params:", bool(torch.isfinite(tr.flat_p).all()), "finite grads:", bool(torch.isfinite(tr.flat_g).all()))
