import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import B, T, build_model, synth_batch
from zeronotesamba_amd.pretext import PretextTrainer
dev = torch.device("cuda", 0)
tr = PretextTrainer(build_model(dev), batch_len=B, seed=1)
x = tr.stack_inputs(synth_batch(1, dev))
for _ in range(3): tr.step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): tr.step(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue ms/step", (t1 - t0) / 10 * 1e3, "total ms/step", (t2 - t0) / 10 * 1e3)
