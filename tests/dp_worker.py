"""One rank of the data-parallel PretextTrainer consistency test (tests/test_gpu_trainer.py): a FRESH process per rank,
as bench.py / torch.distributed.run start them.  Ranks share cuda:0 and exchange gradients over gloo (this box has one
GPU); the trainer code path -- broadcast at construction, bucketed all-reduce issued from the backward hooks, 1/world
folded into the fused Adam -- is the one RCCL runs.

    python tests/dp_worker.py RANK WORLD PORT OUTDIR TRAIN(0|1) STEPS
"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

B, T, TAU, LR, SEED = 4, 64, 0.25, 1e-3, 3


def rank_batch(step: int, rank: int):
    from oracle import cnn_ref
    return cnn_ref.synth_vqt(200 + 10 * step + rank, (B, 2, 96, T))


def rank_state_dict(rank: int):
    from oracle import cnn_ref
    return cnn_ref.synth_state_dict(5 + 10 * rank, 2.0)          # every rank starts from DIFFERENT weights


def main():
    rank, world, port, outdir, train, steps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4],
                                               bool(int(sys.argv[5])), int(sys.argv[6]))
    import torch
    import torch.distributed as dist
    from zeronotesamba_amd.models.models import Pretext_CNN
    from zeronotesamba_amd.pretext import PretextTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = Pretext_CNN()
        m.load_state_dict(rank_state_dict(rank))
        tr = PretextTrainer(m.to("cuda:0"), batch_len=B, temperature=TAU, lr=LR, seed=SEED, train=train,
                            data_parallel=True)
        p0 = tr.flat_p.detach().cpu().clone()                     # after the construction-time broadcast
        outs = []
        for s in range(steps):
            outs.append(tr.step(tr.stack_inputs(rank_batch(s, rank))).cpu())
        torch.cuda.synchronize()
        torch.save({"p0": p0, "p": tr.flat_p.cpu(), "m": tr.flat_m.cpu(), "v": tr.flat_v.cpu(), "out": torch.stack(outs),
                    "step_count": tr.step_count, "world": tr.world, "buckets": len(tr.buckets)},
                   os.path.join(outdir, f"rank{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
