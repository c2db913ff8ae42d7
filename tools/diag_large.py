"""Large-batch consistency on the device: one launch sequence over B_big clips per branch against the same clips in
chunks of 16 (dropout off, so a clip's activations do not depend on its batch) -- embeddings must be bitwise equal,
parameter gradients equal the sum over chunks up to fp32 reassociation.  Looks for 32-bit index / 4-GiB descriptor
overflows at sizes the 288 GB of an MI355X invite.  Also generate_XQT over more clips than one 4-GiB buffer holds.
usage: python3 tools/diag_large.py [B_big ...] [T=512] [xqt=n_clips ...] [n=160000] [chunk=4096]   (default 64 128, xqt=8192 xqt=12288)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from test_gpu_fullsize import DEV, build, run, synth
from zeronotesamba_amd import _lib


def cnn(Bbig, T=512, chunk=16):
    assert Bbig % chunk == 0, "B_big must be a multiple of the chunk (16)"
    model = build()
    x = synth(Bbig, T, 3)
    e_big, _, g_big = run(_lib.ZNS_IMPL_AUTO, model, x, Bbig)
    torch.cuda.synchronize()
    e_parts_a, e_parts_p, g_sum = [], [], None
    for i in range(0, Bbig, chunk):
        xc = torch.cat([x[i:i + chunk], x[Bbig + i:Bbig + i + chunk]]).contiguous()
        e, _, g = run(_lib.ZNS_IMPL_AUTO, model, xc, chunk)
        # the upstream gradient of run() depends on the position inside emb: redo it with the big batch's slice
        e_parts_a.append(e[:chunk]); e_parts_p.append(e[chunk:])
    e_chunks = torch.cat(e_parts_a + e_parts_p)
    same = torch.equal(e_big, e_chunks)
    d = (e_big - e_chunks).abs().max().item()
    print(f"B={Bbig} T={T}: embeddings bitwise equal to chunks of {chunk}: {same} (max diff {d:.3e}); "
          f"grad finite: {all(torch.isfinite(g).all().item() for g in g_big)}", flush=True)
    return same


def cnn_grads(Bbig, T=512, chunk=16):
    """gradients: the same upstream gradient rows for the big batch and for the chunks"""
    from zeronotesamba_amd import _ops
    model = build()
    x = synth(Bbig, T, 4)
    P = _ops.StackParams.from_flat(model.anchor.stack_params() + model.postve.stack_params(), 2)
    gen = torch.Generator().manual_seed(9)
    demb_all = torch.randn((2 * Bbig, T), generator=gen).to(DEV)

    def fb(xx, B, demb):
        with torch.no_grad():
            emb, acts, argm = _ops.stack_forward(xx, P, B, 0.0, 0, save=True)
            grads = [torch.zeros_like(p) for p in P.flat()]
            G = _ops.StackParams.from_flat(grads, 2)
            _ops.stack_backward(xx, P, B, acts, argm, emb, demb.contiguous(), None, 1.0, G)
        torch.cuda.synchronize()
        return grads

    g_big = fb(x, Bbig, demb_all)
    g_sum = None
    for i in range(0, Bbig, chunk):
        xc = torch.cat([x[i:i + chunk], x[Bbig + i:Bbig + i + chunk]]).contiguous()
        dc = torch.cat([demb_all[i:i + chunk], demb_all[Bbig + i:Bbig + i + chunk]])
        g = fb(xc, chunk, dc)
        g_sum = g if g_sum is None else [a + b for a, b in zip(g_sum, g)]
    worst = max(((a - b).norm() / b.norm().clamp_min(1e-30)).item() for a, b in zip(g_big, g_sum))
    print(f"B={Bbig} T={T}: parameter gradients vs sum over chunks: worst relative L2 {worst:.3e}", flush=True)
    return worst < 1e-4


def xqt(n_clips, n=160000, chunk=4096):
    import zeronotesamba_amd.processing.input_rep as IR
    g = torch.Generator().manual_seed(5)
    base = torch.randn((64, n), generator=g).to(DEV) * 0.1
    wav = base.repeat((n_clips + 63) // 64, 1)[:n_clips].contiguous()
    wav += torch.linspace(0, 1e-3, n_clips, device=DEV)[:, None]          # every clip differs
    big = IR.generate_XQT_batch(wav, 16000, "vqt")
    torch.cuda.synchronize()
    ok = True
    for i in range(0, n_clips, chunk):
        part = IR.generate_XQT_batch(wav[i:i + chunk].contiguous(), 16000, "vqt")
        ok = ok and torch.equal(part, big[i:i + chunk])
    print(f"generate_XQT {n_clips} clips x {n} samples ({wav.numel() * 4 / 2**30:.1f} GiB): bitwise equal to chunks of {chunk}: {ok}",
          flush=True)
    return ok


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or ([] if len(sys.argv) > 1 else [64, 128])
    T = next((int(a[2:]) for a in sys.argv[1:] if a.startswith("T=")), 512)
    clips = [int(a[4:]) for a in sys.argv[1:] if a.startswith("xqt=")] or [8192, 12288]
    n_samples = next((int(a[2:]) for a in sys.argv[1:] if a.startswith("n=")), 160000)
    xchunk = next((int(a[6:]) for a in sys.argv[1:] if a.startswith("chunk=")), 4096)
    ok = True
    for B in sizes:
        ok = cnn(B, T) and ok
        ok = cnn_grads(B, T) and ok
        torch.cuda.empty_cache()
    for n in clips:
        ok = xqt(n, n_samples, xchunk) and ok
        torch.cuda.empty_cache()
    print("ALL OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
