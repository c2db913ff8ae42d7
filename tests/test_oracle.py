"""CPU: pin the oracle against the golden vectors captured from the real reference
(tests/golden/make_golden.py).  No GPU, no /root/reference at run time."""
import numpy as np
import pytest
import torch

from oracle import cnn_ref, dropout_ref, ntxent_ref, step_ref

torch.set_num_threads(8)


def sample_idx(n, k=16):
    return np.unique(np.linspace(0, n - 1, k).astype(np.int64))


def check_stats(t: torch.Tensor, stats, samples, rtol=2e-4, atol=1e-6):
    f = t.detach().double().flatten()
    got = np.array([f.sum().item(), (f * f).sum().sqrt().item()])
    l2 = max(stats[1], 1e-30)
    assert abs(got[1] - stats[1]) <= rtol * l2 + atol, (got, stats)
    assert abs(got[0] - stats[0]) <= rtol * l2 * np.sqrt(f.numel()) + atol, (got, stats)
    s = f[sample_idx(f.numel())].numpy()
    scale = l2 / np.sqrt(f.numel())
    np.testing.assert_allclose(s, samples, rtol=2e-3, atol=2e-3 * scale + atol)


@pytest.mark.parametrize("name", ["u313", "u512", "collinear", "b4", "short"])
def test_ntxent_matches_reference(golden, name):
    g = golden(f"ntxent_{name}.npz")
    a = torch.tensor(g["a"], requires_grad=True)
    p = torch.tensor(g["p"], requires_grad=True)
    loss, pos, neg = ntxent_ref.ntxent(a, p, int(g["batch_len"]), float(g["tau"]))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-6
    assert abs(pos - float(g["pos"])) < 1e-6
    assert abs(neg - float(g["neg"])) < 1e-6
    np.testing.assert_allclose(a.grad.numpy(), g["ga"], rtol=1e-4, atol=2e-8)
    np.testing.assert_allclose(p.grad.numpy(), g["gp"], rtol=1e-4, atol=2e-8)
    # independent float64 closed form + analytic gradients
    l64, p64, n64, ga, gp = ntxent_ref.ntxent_numpy(g["a"], g["p"], int(g["batch_len"]), float(g["tau"]))
    assert abs(l64 - float(g["loss"])) < 2e-6
    np.testing.assert_allclose(ga, g["ga"], rtol=2e-3, atol=5e-8)
    np.testing.assert_allclose(gp, g["gp"], rtol=2e-3, atol=5e-8)


def test_manifest(golden):
    g = golden("manifest.npz")
    shapes = cnn_ref.pretext_param_shapes()
    assert list(g["keys"]) == list(shapes.keys())
    assert [str(tuple(s)) for s in shapes.values()] == list(g["shapes"])
    assert int(g["n_params"]) == 26819202
    assert list(g["down_keys"]) == ["pretext." + k for k in shapes.keys()]


def test_cnn_eval_matches_reference(golden):
    g = golden("cnn_eval.npz")
    B, T = int(g["B"]), int(g["T"])
    sd = cnn_ref.synth_state_dict(int(g["seed"]), float(g["gain"]))
    anc = cnn_ref.synth_vqt(101, (B, 1, 96, T))
    pos = cnn_ref.synth_vqt(102, (B, 1, 96, T))
    with torch.no_grad():
        a, p = cnn_ref.pretext_forward(sd, anc, pos)
    np.testing.assert_allclose(a.numpy(), g["anc_emb"], atol=2e-6)
    np.testing.assert_allclose(p.numpy(), g["pos_emb"], atol=2e-6)
    dsd = {"pretext." + k: v for k, v in sd.items()}
    with torch.no_grad():
        for red in ("max", "mean"):
            np.testing.assert_allclose(cnn_ref.down_forward(dsd, anc, pos, red).numpy(), g[f"down_{red}"], atol=2e-6)
        anc2 = cnn_ref.synth_vqt(103, (1, 1, 96, 45))
        np.testing.assert_allclose(cnn_ref.ds_cnn_forward(sd, "anchor.", anc2).numpy(), g["anc_emb_T45"], atol=2e-6)


def test_pretext_step_matches_reference(golden):
    g = golden("step_eval.npz")
    B, T, tau = int(g["B"]), int(g["T"]), float(g["tau"])
    for lr_tag, lr in (("lr1e-6", 1e-6), ("lr3e-5", 3e-5)):
        sd = cnn_ref.synth_state_dict(int(g["seed"]), float(g["gain"]))
        op = step_ref.OracleParams(sd, lr)
        for it in range(3):
            batch = cnn_ref.synth_vqt(200 + it, (B, 2, 96, T))
            loss, cp, cn, a, p = step_ref.pretext_step(op, batch, B, tau)
            ref = g[f"{lr_tag}_losses"][it]
            tol = 5e-6 if (lr < 1e-5 or it == 0) else 2e-3  # early Adam steps are sign-like: rounding noise in g is amplified at the larger lr
            assert abs(loss - ref[0]) < tol and abs(cp - ref[1]) < tol and abs(cn - ref[2]) < tol
            if it == 0 and lr_tag == "lr1e-6":
                np.testing.assert_allclose(a.numpy(), g["s0_anc_emb"], atol=2e-6)
                for k, gr in op.grads().items():
                    check_stats(gr, g[f"s0_grad_{k}_stats"], g[f"s0_grad_{k}_samples"])
            if it in (0, 2):
                for k, v in op.values().items():
                    check_stats(v, g[f"{lr_tag}_s{it}_param_{k}_stats"], g[f"{lr_tag}_s{it}_param_{k}_samples"],
                                rtol=1e-5 if it == 0 else 1e-4)


def ds_state_dict(seed, gain):
    return {k[len("anchor."):]: v for k, v in cnn_ref.synth_state_dict(seed, gain).items() if k.startswith("anchor.")}


def test_clmr_step_matches_reference(golden):
    """pt_task "clmr" (pretext.py:494-511): the oracle's shared-DS_CNN step against vectors from the real reference."""
    g = golden("clmr_eval.npz")
    B, T, tau, lr = int(g["B"]), int(g["T"]), float(g["tau"]), float(g["lr"])
    op = step_ref.OracleParams(ds_state_dict(int(g["seed"]), float(g["gain"])), lr)
    for it in range(3):
        batch = cnn_ref.synth_vqt(210 + it, (B, 2, 96, T))
        loss, cp, cn, a, p = step_ref.clmr_step(op, batch, B, tau)
        ref = g["losses"][it]
        tol = 5e-6 if it == 0 else 5e-4   # sign-like early Adam steps at lr 1e-5 amplify rounding noise in g
        assert abs(loss - ref[0]) < tol and abs(cp - ref[1]) < tol and abs(cn - ref[2]) < tol
        if it == 0:
            np.testing.assert_allclose(a.numpy(), g["s0_anc_emb"], atol=2e-6)
            np.testing.assert_allclose(p.numpy(), g["s0_pos_emb"], atol=2e-6)
            for k, gr in op.grads().items():
                check_stats(gr, g[f"s0_grad_{k}_stats"], g[f"s0_grad_{k}_samples"])
        if it in (0, 2):
            for k, v in op.values().items():
                check_stats(v, g[f"s{it}_param_{k}_stats"], g[f"s{it}_param_{k}_samples"], rtol=1e-5 if it == 0 else 1e-4)
    with torch.no_grad():
        for it in range(2):
            batch = cnn_ref.synth_vqt(220 + it, (B, 2, 96, T))
            a = cnn_ref.ds_cnn_forward(op.sd, "", batch[:, 0:1])
            p = cnn_ref.ds_cnn_forward(op.sd, "", batch[:, 1:2])
            loss, cp, cn = ntxent_ref.ntxent(a, p, B, tau)
            ref = g["val_after"][it]
            assert abs(loss.item() - ref[0]) < 5e-4 and abs(cp - ref[1]) < 5e-4 and abs(cn - ref[2]) < 5e-4


def own_decisions(sd, prefix, x):
    """The max-pool / ReLU decisions a plain forward of the oracle takes, in the replay record's format."""
    import torch.nn.functional as F
    rec, out = {}, x
    for li, (name, cin, cout, kh, kw, pool) in enumerate(cnn_ref.LAYERS):
        z = F.conv2d(out, sd[f"{prefix}pretrained.{name}.weight"], sd[f"{prefix}pretrained.{name}.bias"],
                     padding=(kh // 2, kw // 2))
        am = None
        if pool > 1:
            Bn, C, H, Tt = z.shape
            am = z.reshape(Bn, C, H // pool, pool, Tt).argmax(dim=3).to(torch.uint8)
            z = F.max_pool2d(z, (pool, 1))
        out = F.relu(z)
        rec[li] = (out != 0, am)
    return rec


def test_mask_replay_of_own_decisions_is_the_identity():
    """Mask-replay mode (cnn_ref.cnn_forward(replay=...)) fed the oracle's OWN decisions reproduces the plain step bit
    for bit, and counts zero overridden decisions; fed ONE flipped pool decision it reports exactly that flip."""
    B, T, tau = 2, 40, 0.25
    sd = cnn_ref.synth_state_dict(5, 2.0)
    batch = cnn_ref.synth_vqt(200, (B, 2, 96, T))
    op = step_ref.OracleParams(sd, 1e-6)
    l0, _, _, a0, p0 = step_ref.pretext_step(op, batch, B, tau, step_opt=False)
    g0 = op.grads()
    rec = {"anchor": own_decisions(sd, "anchor.", batch[:, 0:1]), "postve": own_decisions(sd, "postve.", batch[:, 1:2])}
    op2 = step_ref.OracleParams(sd, 1e-6)
    flips = {}
    l1, _, _, a1, p1 = step_ref.pretext_step(op2, batch, B, tau, step_opt=False, replay=rec, flips=flips)
    assert l0 == l1 and torch.equal(a0, a1) and torch.equal(p0, p1)
    g1 = op2.grads()
    assert all(torch.equal(g0[k], g1[k]) for k in g0)
    assert all(n_am == 0 and n_mask == 0 for per in flips.values() for (n_am, n_mask, _n) in per.values())
    am = rec["anchor"][1][1]
    am[0, 0, 0, 0] = (am[0, 0, 0, 0] + 1) % 3
    flips = {}
    step_ref.pretext_step(step_ref.OracleParams(sd, 1e-6), batch, B, tau, step_opt=False, replay=rec, flips=flips)
    assert flips["anchor"][1][0] == 1 and flips["postve"][1][0] == 0


def test_finetune_step_matches_reference(golden):
    g = golden("finetune.npz")
    T = int(g["T"])
    vqt = cnn_ref.synth_vqt(300, (2, 96, T))
    msk = torch.from_numpy(g["msk"])
    for red in ("max", "mean"):
        sd = {"pretext." + k: v for k, v in cnn_ref.synth_state_dict(int(g["seed"]), float(g["gain"])).items()}
        op = step_ref.OracleParams(sd, 0.5 * 1e-5 * 10e-2)
        loss, out = step_ref.finetune_step(op, vqt, msk, reduction=red)
        assert abs(loss - float(g[f"{red}_loss"])) < 5e-6
        np.testing.assert_allclose(out.numpy(), g[f"{red}_out"], atol=2e-6)
        for k, gr in op.grads().items():
            check_stats(gr, g[f"{red}_grad_{k}_stats"], g[f"{red}_grad_{k}_samples"])
        for k, v in op.values().items():
            check_stats(v, g[f"{red}_s0_param_{k}_stats"], g[f"{red}_s0_param_{k}_samples"], rtol=1e-5)


def test_dropout_mask_statistics():
    m = dropout_ref.keep_mask(dropout_ref.layer_seed(1234, 3, 1), (4, 128, 8, 64), 0.1)
    rate = m.mean()
    assert abs(rate - 0.9) < 3e-3
    # different (layer, branch) streams are decorrelated
    m2 = dropout_ref.keep_mask(dropout_ref.layer_seed(1234, 3, 0), (4, 128, 8, 64), 0.1)
    assert abs((m & m2).mean() - 0.81) < 5e-3
    # train-mode forward uses exactly this mask and rescales by 1/(1-p)
    sd = cnn_ref.synth_state_dict(3, 2.0)
    x = cnn_ref.synth_vqt(9, (1, 1, 96, 40))
    taps = []
    with torch.no_grad():
        cnn_ref.cnn_forward(sd, "anchor.", x, drop_seed=77, branch=0, taps=taps)
    y1 = taps[0]
    k1 = dropout_ref.keep_mask(dropout_ref.layer_seed(77, 0, 0), tuple(y1.shape), 0.1)
    assert (y1.numpy()[~k1] == 0).all()


def test_adam_numpy_matches_torch():
    rng = np.random.default_rng(0)
    p = rng.standard_normal(1000).astype(np.float32)
    tp = torch.tensor(p.copy(), requires_grad=True)
    opt = torch.optim.Adam([tp], lr=1e-3)
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for step in range(1, 4):
        gr = rng.standard_normal(1000).astype(np.float32)
        tp.grad = torch.tensor(gr.copy())
        opt.step()
        p, m, v = step_ref.adam_numpy(p, gr, m, v, step, 1e-3)
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=2e-6, atol=1e-7)
