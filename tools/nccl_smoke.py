"""Single-rank RCCL smoke on the GPU box: the bucketed all-reduce path of PretextTrainer with a real NCCL
process group (world_size 1), checked bitwise against the collective-free step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

from bench import B, T, build_model, synth_batch
from zeronotesamba_amd.distributed import GradBucketReducer
from zeronotesamba_amd.pretext import PretextTrainer

torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29531", rank=0, world_size=1)
dev = torch.device("cuda", 0)
ta = PretextTrainer(build_model(dev), batch_len=B, seed=3)
tb = PretextTrainer(build_model(dev), batch_len=B, seed=3)
tb.reducer = GradBucketReducer(tb.flat_g, tb.buckets, None, always=True)
x = ta.stack_inputs(synth_batch(5, dev))
for _ in range(3):
    oa, ob = ta.step(x), tb.step(x)
torch.cuda.synchronize()
print("buckets", [(n, e - s) for n, s, e in tb.buckets])
print("loss", oa.tolist(), ob.tolist())
print("bitwise grads", torch.equal(ta.flat_g, tb.flat_g), "params", torch.equal(ta.flat_p, tb.flat_p))
dist.destroy_process_group()
