#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference modules (this container only).

Run from the repo root:  PYTHONPATH=/root/reference:. python tests/golden/make_golden.py

Imports zeroNoteSamba.models.models / models.loss_functions from /root/reference
(they need only torch), drives them with closed-form synthetic weights/inputs
(oracle.cnn_ref.synth_state_dict / synth_vqt, exactly reproducible from integer
hashes) and stores only DATA: inputs are regenerated from seeds, expected outputs
are saved as small .npz files next to this script.  The reference never travels.

Cases
  ntxent_*.npz   NTXent forward + autograd grads  (loss_functions.py:24-55)
  cnn_eval.npz   Pretext_CNN / Down_CNN eval-mode forward, per-layer taps  (models.py)
  step_eval.npz  one + three pretext steps (pretext.py:475-492, dropout off), grads + Adam
  finetune.npz   Down_CNN + BCELoss + Adam step (epochs.py:48-63), max and mean reductions
  clmr_eval.npz  the "clmr" pretext task (pretext.py:494-511, 567-583): ONE shared DS_CNN on both views,
                 Adam lr 1e-5 (pretext.py:208); three steps, grads + Adam, and the eval-mode loss triple
  manifest.npz   state-dict keys and shapes
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")

from zeroNoteSamba.models.loss_functions import NTXent  # noqa: E402
from zeroNoteSamba.models.models import DS_CNN, Down_CNN, Pretext_CNN  # noqa: E402

from oracle import cnn_ref, dropout_ref  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def sample_idx(n, k=16):
    return np.unique(np.linspace(0, n - 1, k).astype(np.int64))


def tstats(t: torch.Tensor):
    f = t.detach().double().flatten()
    idx = sample_idx(f.numel())
    return np.array([f.sum().item(), (f * f).sum().sqrt().item()]), f[idx].numpy().astype(np.float32)


def make_ntxent():
    cases = {
        "u313": (16, 313, 0.25, "uniform"),
        "u512": (16, 512, 0.25, "uniform"),
        "collinear": (16, 313, 0.25, "collinear"),
        "b4": (4, 64, 0.25, "uniform"),
        "short": (16, 64, 0.5, "short"),  # fewer rows than batch_len (last-batch behaviour, :30)
    }
    for name, (b, d, tau, kind) in cases.items():
        rows = b
        if kind == "short":
            rows = 10
        a = (dropout_ref.uniform_pm1(11, rows * d).reshape(rows, d) * 0.5 + 0.5).astype(np.float32)
        p = (dropout_ref.uniform_pm1(12, rows * d).reshape(rows, d) * 0.5 + 0.5).astype(np.float32)
        if kind == "collinear":
            p = (a * 0.999 + 0.001 * p).astype(np.float32)
        ta = torch.tensor(a, requires_grad=True)
        tp = torch.tensor(p, requires_grad=True)
        crit = NTXent(b, tau)
        if kind == "short":
            # reference indexes poss[xx] for xx < rows only; works for rows < batch_len
            loss, pos, neg = crit(ta, tp)
        else:
            loss, pos, neg = crit(ta, tp)
        loss.backward()
        np.savez(os.path.join(HERE, f"ntxent_{name}.npz"), a=a, p=p, batch_len=b, tau=tau,
                 loss=np.float32(loss.item()), pos=np.float64(pos), neg=np.float64(neg),
                 ga=ta.grad.numpy(), gp=tp.grad.numpy())
        print("ntxent", name, loss.item(), pos, neg)


def load_ref_pretext(seed, gain):
    sd = cnn_ref.synth_state_dict(seed, gain)
    m = Pretext_CNN()
    m.load_state_dict(sd)
    return m, sd


def make_cnn_eval():
    B, T = 2, 64
    seed, gain = 3, 2.0
    m, sd = load_ref_pretext(seed, gain)
    m.eval()
    anc = cnn_ref.synth_vqt(101, (B, 1, 96, T))
    pos = cnn_ref.synth_vqt(102, (B, 1, 96, T))
    taps = {}

    def hook(name):
        def f(mod, inp, out):
            taps.setdefault(name, []).append(out.detach().clone())
        return f

    # per-conv raw outputs of the anchor branch (pre pool/relu)
    for li, (lname, *_r) in enumerate(cnn_ref.LAYERS):
        getattr(m.anchor.pretrained, lname).register_forward_hook(hook(lname))
    with torch.no_grad():
        a, p = m(anc, pos)
    out = {"B": B, "T": T, "seed": seed, "gain": gain, "anc_emb": a.numpy(), "pos_emb": p.numpy()}
    for lname, v in taps.items():
        st, smp = tstats(v[0])
        out[f"tap_{lname}_stats"] = st
        out[f"tap_{lname}_samples"] = smp
    # Down_CNN on the same weights, both reductions (models.py:144-148)
    for red in ("max", "mean"):
        d = Down_CNN(reduction=red)
        d.pretext.load_state_dict(sd)
        d.eval()
        with torch.no_grad():
            out[f"down_{red}"] = d(anc, pos).numpy()
    # a second length that is not a multiple of 32 (fully-convolutional in T)
    T2 = 45
    anc2 = cnn_ref.synth_vqt(103, (1, 1, 96, T2))
    with torch.no_grad():
        out["anc_emb_T45"] = m.anchor(anc2).numpy()
    np.savez(os.path.join(HERE, "cnn_eval.npz"), **out)
    print("cnn_eval emb range", a.min().item(), a.max().item(), a.std().item())


def grad_record(model, out, tag):
    for k, v in model.named_parameters():
        st, smp = tstats(v.grad)
        out[f"{tag}grad_{k}_stats"] = st
        out[f"{tag}grad_{k}_samples"] = smp


def param_record(model, out, tag):
    for k, v in model.named_parameters():
        st, smp = tstats(v)
        out[f"{tag}param_{k}_stats"] = st
        out[f"{tag}param_{k}_samples"] = smp


def make_step_eval():
    B, T, tau = 4, 64, 0.25
    seed, gain = 5, 2.0
    out = {"B": B, "T": T, "seed": seed, "gain": gain, "tau": tau}
    for lr_tag, lr in (("lr1e-6", 1e-6), ("lr3e-5", 3e-5)):
        m, sd = load_ref_pretext(seed, gain)
        m.eval()  # dropout off; there is no BatchNorm, so eval differs from train only by dropout
        crit = NTXent(B, tau)
        opt = torch.optim.Adam(m.parameters(), lr=lr)
        losses = []
        for it in range(3):
            batch = cnn_ref.synth_vqt(200 + it, (B, 2, 96, T))
            anchors = batch[:, 0:1, :, :]
            postves = batch[:, 1:2, :, :]
            opt.zero_grad()
            a, p = m(anchors, postves)
            loss, cp, cn = crit(a, p)
            loss.backward()
            if it == 0 and lr_tag == "lr1e-6":
                grad_record(m, out, "s0_")
                out["s0_anc_emb"] = a.detach().numpy()
                out["s0_pos_emb"] = p.detach().numpy()
            opt.step()
            losses.append([loss.item(), cp, cn])
            if it in (0, 2):
                param_record(m, out, f"{lr_tag}_s{it}_")
        out[f"{lr_tag}_losses"] = np.array(losses)
        print("step", lr_tag, losses)
    np.savez(os.path.join(HERE, "step_eval.npz"), **out)


def make_clmr():
    """pretext.py:205-208 (DS_CNN, Adam lr=1e-5) driven by the clmr branch of train_epoch (:494-511): anc_emb =
    model(anchors); pos_emb = model(postves); NTXent; backward; step.  Dropout off (eval): there is no BatchNorm."""
    B, T, tau, lr = 4, 64, 0.25, 1e-5
    seed, gain = 9, 2.0
    out = {"B": B, "T": T, "seed": seed, "gain": gain, "tau": tau, "lr": lr}
    sd = {k[len("anchor."):]: v for k, v in cnn_ref.synth_state_dict(seed, gain).items() if k.startswith("anchor.")}
    m = DS_CNN()
    m.load_state_dict(sd)
    m.eval()
    crit = NTXent(B, tau)
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    losses = []
    for it in range(3):
        batch = cnn_ref.synth_vqt(210 + it, (B, 2, 96, T))
        anchors = batch[:, 0:1, :, :]
        postves = batch[:, 1:2, :, :]
        opt.zero_grad()
        anc_emb = m(anchors)
        pos_emb = m(postves)
        loss, cp, cn = crit(anc_emb, pos_emb)
        loss.backward()
        if it == 0:
            grad_record(m, out, "s0_")
            out["s0_anc_emb"] = anc_emb.detach().numpy()
            out["s0_pos_emb"] = pos_emb.detach().numpy()
        opt.step()
        losses.append([loss.item(), cp, cn])
        if it in (0, 2):
            param_record(m, out, f"s{it}_")
    out["losses"] = np.array(losses)
    # val_epoch's clmr branch (:567-583) on the UPDATED weights, two batches: the returned means
    vals = []
    with torch.no_grad():
        for it in range(2):
            batch = cnn_ref.synth_vqt(220 + it, (B, 2, 96, T))
            loss, cp, cn = crit(m(batch[:, 0:1, :, :]), m(batch[:, 1:2, :, :]))
            vals.append([loss.item(), cp, cn])
    out["val_after"] = np.array(vals)
    print("clmr", losses, vals)
    np.savez(os.path.join(HERE, "clmr_eval.npz"), **out)


def make_finetune():
    T = 96
    seed, gain = 7, 2.0
    out = {"T": T, "seed": seed, "gain": gain}
    sd = cnn_ref.synth_state_dict(seed, gain)
    vqt = cnn_ref.synth_vqt(300, (2, 96, T))
    msk = np.zeros(T, dtype=np.float32)
    for c in range(5, T, 31):  # ballroom.py:219-221 style pulses: 1 at the beat, .5 neighbours
        msk[c] = 1.0
        if c - 1 >= 0:
            msk[c - 1] = 0.5
        if c + 1 < T:
            msk[c + 1] = 0.5
    out["msk"] = msk
    for red in ("max", "mean"):
        d = Down_CNN(reduction=red)
        d.pretext.load_state_dict(sd)
        d.eval()
        crit = torch.nn.BCELoss()
        opt = torch.optim.Adam(d.parameters(), lr=0.5 * 1e-5 * 10e-2, betas=(0.9, 0.999))  # loader.py:43
        v1 = vqt[0].reshape(1, 1, 96, T)
        v2 = vqt[1].reshape(1, 1, 96, T)
        m = torch.from_numpy(msk).reshape(1, T)
        opt.zero_grad()
        o = d(v1, v2)
        loss = crit(o, m)
        loss.backward()
        out[f"{red}_out"] = o.detach().numpy()
        out[f"{red}_loss"] = np.float32(loss.item())
        grad_record(d, out, f"{red}_")
        opt.step()
        param_record(d, out, f"{red}_s0_")
        print("finetune", red, loss.item())
    np.savez(os.path.join(HERE, "finetune.npz"), **out)


def make_manifest():
    m = Pretext_CNN()
    keys = list(m.state_dict().keys())
    shapes = [tuple(v.shape) for v in m.state_dict().values()]
    d = Down_CNN()
    dkeys = list(d.state_dict().keys())
    np.savez(os.path.join(HERE, "manifest.npz"), keys=np.array(keys), shapes=np.array([str(s) for s in shapes]),
             down_keys=np.array(dkeys), n_params=sum(int(np.prod(s)) for s in shapes))
    print("manifest", len(keys), sum(int(np.prod(s)) for s in shapes))


if __name__ == "__main__":
    make_manifest()
    make_ntxent()
    make_cnn_eval()
    make_step_eval()
    make_clmr()
    make_finetune()
