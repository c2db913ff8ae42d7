"""GPU: the fused PretextTrainer (flat buffers, no autograd) against the drop-in module path with a stock
torch.optim.Adam -- i.e. the reference's own step structure (pretext.py:475-492) -- plus the epoch functions,
the fused Adam and the on-device shift batching (pretext.py:308-318)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cnn_ref  # noqa: E402

DEV = "cuda:0"


def make_model(seed=5):
    from zeronotesamba_amd.models.models import Pretext_CNN
    m = Pretext_CNN()
    m.load_state_dict(cnn_ref.synth_state_dict(seed, 2.0))
    return m.to(DEV)


def test_trainer_step_equals_module_path():
    """Same kernels through two host paths: gradients must be bitwise equal, losses equal."""
    from zeronotesamba_amd.models.loss_functions import NTXent
    from zeronotesamba_amd.pretext import PretextTrainer
    B, T, tau, lr = 4, 64, 0.25, 1e-4
    ma, mb = make_model(), make_model()
    ma.eval()
    tr = PretextTrainer(mb, batch_len=B, temperature=tau, lr=lr, train=False)
    crit = NTXent(B, tau)
    batch = cnn_ref.synth_vqt(200, (B, 2, 96, T))
    a, p = ma(batch[:, 0:1].to(DEV), batch[:, 1:2].to(DEV))
    loss, cp, cn = crit(a, p)
    loss.backward()
    out3 = tr.step(tr.stack_inputs(batch), update=False).tolist()
    assert abs(out3[0] - loss.item()) < 1e-6 and abs(out3[1] - cp) < 1e-6 and abs(out3[2] - cn) < 1e-6
    for (ka, va), (kb, vb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert ka == kb
        assert torch.equal(va.grad, vb.grad), ka
    # the trainer keeps module parameters as views of its flat buffer: state_dict still round-trips
    sd = mb.state_dict()
    assert sd["anchor.pretrained.cv4.weight"].data_ptr() == mb.anchor.pretrained.cv4.weight.data_ptr()
    # and one update moves every parameter by at most lr (Adam's first step is sign-like)
    before = tr.flat_p.clone()
    tr.step(tr.stack_inputs(batch))
    d = (tr.flat_p - before).abs()
    assert d.max().item() <= lr * 1.001 and d.mean().item() > 0.5 * lr


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_a_non_finite_input_gives_a_non_finite_loss(bad):
    """pretext.py:486-492: the reference's loop would print a NaN loss the step after its input (or a weight) went bad.
    One poisoned spectrogram element, train mode (a dropped NaN must stay a NaN): the loss of the module path and of the
    fused trainer are NaN, every gradient tensor is poisoned, and the oracle agrees."""
    from oracle import step_ref
    from zeronotesamba_amd.models.loss_functions import NTXent
    from zeronotesamba_amd.pretext import PretextTrainer
    B, T, tau = 4, 64, 0.25
    batch = cnn_ref.synth_vqt(200, (B, 2, 96, T))
    batch[1, 0, 40, 33] = bad
    ma, mb = make_model(), make_model()
    ma.train()
    a, p = ma(batch[:, 0:1].to(DEV), batch[:, 1:2].to(DEV))
    loss, _, _ = NTXent(B, tau)(a, p)
    assert torch.isnan(loss)
    tr = PretextTrainer(mb, batch_len=B, temperature=tau, lr=1e-4, train=True)
    out3 = tr.step(tr.stack_inputs(batch), update=False)
    assert torch.isnan(out3[0])
    assert not torch.isfinite(tr.flat_g).all()
    op = step_ref.OracleParams(cnn_ref.synth_state_dict(5, 2.0), 1e-4)
    rl = step_ref.pretext_step(op, batch, B, tau)[0]
    assert np.isnan(rl)


def test_gradient_accumulation_through_step_accumulate():
    """step(update=False) leaves ONE batch's gradient in flat_g; accumulate=True adds the next one in place (the
    backward-weights kernels accumulate), apply_update(1 / k) then applies the mean -- and every micro-batch of one
    accumulation draws its own dropout masks (the seed folds in a micro-batch counter that the update resets)."""
    from zeronotesamba_amd.pretext import PretextTrainer
    B, T = 4, 64
    tr = PretextTrainer(make_model(), batch_len=B, lr=1e-4, train=False)
    x1 = tr.stack_inputs(cnn_ref.synth_vqt(201, (B, 2, 96, T)))
    x2 = tr.stack_inputs(cnn_ref.synth_vqt(202, (B, 2, 96, T)))
    tr.step(x1, update=False)
    g1 = tr.flat_g.clone()
    tr.step(x2, update=False)
    g2 = tr.flat_g.clone()
    assert not torch.equal(g1, g2)
    tr.step(x1, update=False)
    tr.step(x2, update=False, accumulate=True)
    ref = g1 + g2
    assert (tr.flat_g - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()
    before, steps = tr.flat_p.clone(), tr.step_count
    tr.apply_update(0.5)
    assert tr.step_count == steps + 1 and tr.micro_count == 0 and not torch.equal(before, tr.flat_p)
    # dropout on: two micro-batches of the SAME input inside one accumulation must not share their masks
    tt = PretextTrainer(make_model(), batch_len=B, lr=1e-4, train=True, seed=11)
    tt.step(x1, update=False)
    ga = tt.flat_g.clone()
    tt.step(x1, update=False)
    assert not torch.equal(ga, tt.flat_g)


def test_fused_adam_matches_torch_adam():
    """zns_adam_step_f32 against torch.optim.Adam on well-conditioned gradients (|g| >> eps)."""
    from zeronotesamba_amd.optim import Adam
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(100003, generator=g)
    pa = p0.clone().to(DEV).requires_grad_(True)
    pb = p0.clone().to(DEV).requires_grad_(True)
    oa, ob = torch.optim.Adam([pa], lr=1e-3), Adam([pb], lr=1e-3)
    for _ in range(5):
        gr = torch.randn(100003, generator=g).to(DEV)
        pa.grad, pb.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    assert (pa - pb).abs().max().item() < 2e-6
    with pytest.raises(ValueError):
        Adam([pb], lr=1e-3, weight_decay=0.1)


def test_epoch_functions_and_fused_adam():
    from zeronotesamba_amd.models.loss_functions import NTXent
    from zeronotesamba_amd.optim import Adam
    from zeronotesamba_amd.pretext import PretextTrainer, train_epoch, val_epoch
    B, T = 4, 40
    loader = [[cnn_ref.synth_vqt(300 + i, (B, 2, 96, T))] for i in range(2)]
    m1, m2 = make_model(7), make_model(7)
    crit = NTXent(B, 0.25)
    torch.manual_seed(3)
    m1, l1, p1, n1 = train_epoch(m1, loader, crit, torch.optim.Adam(m1.parameters(), lr=1e-5))
    torch.manual_seed(3)
    m2, l2, p2, n2 = train_epoch(m2, loader, crit, Adam(m2.parameters(), lr=1e-5))
    assert abs(l1 - l2) < 1e-5 and abs(p1 - p2) < 1e-5 and abs(n1 - n2) < 1e-5
    for va, vb in zip(m1.parameters(), m2.parameters()):
        assert (va - vb).abs().max().item() <= 2.1 * 1e-5   # two sign-like steps of lr each, |g| ~ eps elements may differ
    vl, vp, vn = val_epoch(m1, loader, crit, None)
    assert np.isfinite([vl, vp, vn]).all() and 0 < vp <= 1.0
    with pytest.raises(ValueError, match="Which pretext task"):
        train_epoch(m1, loader, crit, None, pt_task="nope")
    tr = PretextTrainer(make_model(8), batch_len=B, lr=1e-6)
    tl, tp, tn = tr.train_epoch(loader)
    assert np.isfinite([tl, tp, tn]).all()
    with pytest.raises(IndexError):
        tr.step(torch.zeros(2 * (B + 1), 1, 96, T, device=DEV))


def test_shift_gather_matches_reference_indexing():
    """pretext.py:312-318: 16 distinct starts in range(313), 313-frame crops of one 626-frame bank row."""
    from zeronotesamba_amd import _lib, _ops
    rng = np.random.default_rng(0)
    bank = torch.from_numpy(rng.standard_normal((3, 2, 96, 626)).astype(np.float32)).to(DEV)
    starts = rng.choice(313, size=16, replace=False).astype(np.int32)
    clip = np.full(16, 1, dtype=np.int32)
    out = torch.empty((2, 16, 96, 313), dtype=torch.float32, device=DEV)
    clip_d, starts_d = torch.from_numpy(clip).to(DEV), torch.from_numpy(starts).to(DEV)   # keep alive across the call
    rc = _lib.lib().zns_shift_gather_f32(bank.data_ptr(), 626, clip_d.data_ptr(), starts_d.data_ptr(), 16, 313,
                                         out.data_ptr(), _ops.stream_ptr())
    _lib.check(rc, "shift_gather")
    ref = torch.stack([bank[1, :, :, s:s + 313] for s in starts], dim=0)      # [16, 2, 96, 313] as new_train_bank rows
    assert torch.equal(out, ref.transpose(0, 1).contiguous())


def test_shift_bank_sampler_and_fused_epoch():
    """Resident bank + on-device crops follow the reference's sampling law and batch grouping (pretext.py:308-321)."""
    import random
    from zeronotesamba_amd.pretext import PretextTrainer, ShiftBank
    rng = np.random.default_rng(1)
    bank_np = rng.standard_normal((3, 2, 96, 626)).astype(np.float32)
    sb = ShiftBank(bank_np, batch_len=16, crop=313, device=DEV)
    random.seed(5)
    expect = random.sample(range(0, 313), 16)                       # what the reference would draw
    random.seed(5)
    x = sb.batch(2)
    assert tuple(x.shape) == (32, 1, 96, 313)
    ref = np.stack([bank_np[2, :, :, s:s + 313] for s in expect])   # rows of new_train_bank for clip 2
    np.testing.assert_array_equal(x[:16, 0].cpu().numpy(), ref[:, 0])
    np.testing.assert_array_equal(x[16:, 0].cpu().numpy(), ref[:, 1])
    assert len(set(expect)) == 16 and max(expect) < 313
    with pytest.raises(IndexError):
        sb.batch(3)
    # np.random.shuffle(train_bank) (pretext.py:282): same clip order as the reference under np.random.seed, no bank copy
    big = rng.standard_normal((7, 2, 96, 330)).astype(np.float32)
    sb7 = ShiftBank(big, batch_len=4, crop=313, device=DEV)
    ptr = sb7.bank.data_ptr()
    ref_bank = big.copy()
    np.random.seed(11)
    np.random.shuffle(ref_bank)
    np.random.shuffle(ref_bank)                      # every epoch shuffles again (pretext.py:282)
    np.random.seed(11)
    sb7.shuffle_()
    sb7.shuffle_()
    assert sb7.bank.data_ptr() == ptr                # the 13.8 GB bank is never copied
    for c in range(7):
        got = sb7.batch(c, starts=[0, 5, 9, 16])
        np.testing.assert_array_equal(got[0, 0].cpu().numpy(), ref_bank[c, 0, :, 0:313])
        np.testing.assert_array_equal(got[4 + 2, 0].cpu().numpy(), ref_bank[c, 1, :, 9:322])
    sb7.shuffle_(np.random.RandomState(3))
    assert sorted(sb7.order.tolist()) == list(range(7))
    with pytest.raises(ValueError):
        sb.batch(0, starts=[0] * 15)
    tr = PretextTrainer(make_model(9), batch_len=16, lr=1e-6)
    l, p, n = tr.train_epoch(sb.batches([0, 1], rng=random.Random(3)))
    assert np.isfinite([l, p, n]).all()


def test_load_models_rules():
    """loader.py:8-69: checkpoint loading, freezing and learning-rate rules."""
    from zeronotesamba_amd.loader import load_models
    sd = cnn_ref.synth_state_dict(7, 2.0)
    crit, opt, model = load_models("pretrained", "frozen", 1e-5, state_dict=sd)
    assert not any(p.requires_grad for p in model.pretext.anchor.pretrained.parameters())
    assert all(p.requires_grad for p in model.pretext.anchor.fc1.parameters())
    assert sum(len(g["params"]) for g in opt.param_groups) == 4 and opt.param_groups[0]["lr"] == 1e-5
    crit, opt, model = load_models("pretrained", "finetune", 1e-5, state_dict=sd)
    assert abs(opt.param_groups[0]["lr"] - 0.5 * 1e-5 * 10e-2) < 1e-18 and len(opt.param_groups[0]["params"]) == 36
    x = cnn_ref.synth_vqt(300, (2, 96, 40))
    out = model(x[0].reshape(1, 1, 96, 40).to(DEV), x[1].reshape(1, 1, 96, 40).to(DEV))
    loss = crit(out, torch.full_like(out, 0.5))
    loss.backward()
    opt.step()
    crit, opt, model = load_models("vanilla", "", 1e-4)
    assert type(model).__name__ == "DS_CNN" and opt.param_groups[0]["lr"] == 1e-4


def test_clmr_trainer_matches_two_pass_module_path():
    """pt_task "clmr" (pretext.py:494-511): ONE DS_CNN on both views.  The fused trainer runs it as a single stack
    over the 2B rows; the reference structure is two autograd passes whose weight gradients add up."""
    from zeronotesamba_amd.models.loss_functions import NTXent
    from zeronotesamba_amd.models.models import DS_CNN, Pretext_CNN
    from zeronotesamba_amd.pretext import PretextTrainer
    B, T, tau = 4, 64, 0.25
    sd = {k[len("anchor."):]: v for k, v in cnn_ref.synth_state_dict(9, 2.0).items() if k.startswith("anchor.")}
    ma, mb = DS_CNN(), DS_CNN()
    ma.load_state_dict(sd)
    mb.load_state_dict(sd)
    ma, mb = ma.to(DEV).eval(), mb.to(DEV)
    batch = cnn_ref.synth_vqt(210, (B, 2, 96, T))
    crit = NTXent(B, tau)
    loss, cp, cn = crit(ma(batch[:, 0:1].to(DEV)), ma(batch[:, 1:2].to(DEV)))
    loss.backward()
    tr = PretextTrainer(mb, batch_len=B, temperature=tau, lr=1e-4, train=False)
    assert tr.pt_task == "clmr" and tr.nbr == 1
    out3 = tr.step(tr.stack_inputs(batch), update=False).tolist()
    assert abs(out3[0] - loss.item()) < 2e-6 and abs(out3[1] - cp) < 2e-6 and abs(out3[2] - cn) < 2e-6
    for (ka, va), (kb, vb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert ka == kb
        ref = va.grad
        err = (vb.grad - ref).norm().item() / max(ref.norm().item(), 1e-30)
        assert err < 1e-2, (ka, err)     # split-K order differs between B and 2B rows; see DESIGN.md section 6
    ev = tr.evaluate(tr.stack_inputs(batch)).tolist()
    assert abs(ev[0] - loss.item()) < 2e-6
    with pytest.raises(TypeError):
        PretextTrainer(mb, pt_task="zerons")
    with pytest.raises(ValueError):
        PretextTrainer(mb, pt_task="byol")
    with pytest.raises(TypeError):
        PretextTrainer(Pretext_CNN().to(DEV), pt_task="clmr")


def _check_stats(t, stats, samples, rtol, srtol=3e-3, atol=1e-7):
    f = t.detach().cpu().double().flatten()
    idx = np.unique(np.linspace(0, f.numel() - 1, 16).astype(np.int64))
    l2 = max(stats[1], 1e-30)
    assert abs((f * f).sum().sqrt().item() - stats[1]) <= rtol * l2 + atol
    assert abs(f.sum().item() - stats[0]) <= rtol * l2 * np.sqrt(f.numel()) + atol
    np.testing.assert_allclose(f[idx].numpy(), samples, rtol=srtol, atol=srtol * l2 / np.sqrt(f.numel()) + atol)


def test_epoch_returns_match_the_reference(golden):
    """pretext.train_epoch / val_epoch RETURN VALUES against vectors from the real reference (golden step_eval.npz):
    val_epoch on the first golden batch returns that batch's (loss, pos, neg); train_epoch over the three golden batches
    (dropout probability set to 0 on the modules' own `dp`, as the fixture was captured in eval mode: there is no
    BatchNorm, so train and eval differ only by dropout) returns the mean of the three per-step triples, with a stock
    torch Adam, with the fused Adam, and through PretextTrainer.train_epoch.  pretext.py:453-592."""
    from zeronotesamba_amd.models.loss_functions import NTXent
    from zeronotesamba_amd.optim import Adam
    from zeronotesamba_amd.pretext import PretextTrainer, train_epoch, val_epoch
    g = golden("step_eval.npz")
    B, T, tau = int(g["B"]), int(g["T"]), float(g["tau"])
    ref = g["lr1e-6_losses"]
    loader = [[cnn_ref.synth_vqt(200 + it, (B, 2, 96, T))] for it in range(3)]
    crit = NTXent(B, tau)
    m = make_model(int(g["seed"]))
    opt = torch.optim.Adam(m.parameters(), lr=1e-6)
    m.anchor.pretrained.cv1.weight.grad = torch.ones_like(m.anchor.pretrained.cv1.weight)
    got = val_epoch(m, loader[:1], crit, opt)
    np.testing.assert_allclose(got, ref[0], atol=2e-5)
    assert m.anchor.pretrained.cv1.weight.grad is None or not m.anchor.pretrained.cv1.weight.grad.any()   # :552 zero_grad
    assert not m.training
    for make_opt in (lambda ps: torch.optim.Adam(ps, lr=1e-6), lambda ps: Adam(ps, lr=1e-6)):
        m = make_model(int(g["seed"]))
        for br in (m.anchor, m.postve):
            br.pretrained.dp.p = 0.0
        m2, l, p, n = train_epoch(m, loader, crit, make_opt(m.parameters()))
        assert m2 is m and m.training
        np.testing.assert_allclose([l, p, n], ref.mean(axis=0), atol=2e-5)
        for k, v in m.named_parameters():
            _check_stats(v, g[f"lr1e-6_s2_param_{k}_stats"], g[f"lr1e-6_s2_param_{k}_samples"], rtol=1e-5)
    m = make_model(int(g["seed"]))
    for br in (m.anchor, m.postve):
        br.pretrained.dp.p = 0.0
    tr = PretextTrainer(m, batch_len=B, temperature=tau, lr=1e-6)
    np.testing.assert_allclose(tr.train_epoch(loader), ref.mean(axis=0), atol=2e-5)
    for k, v in m.named_parameters():
        _check_stats(v, g[f"lr1e-6_s2_param_{k}_stats"], g[f"lr1e-6_s2_param_{k}_samples"], rtol=1e-5)
    # and with dropout really on the epoch still runs and differs
    m = make_model(int(g["seed"]))
    _, l_do, _, _ = train_epoch(m, loader, crit, torch.optim.Adam(m.parameters(), lr=1e-6))
    assert np.isfinite(l_do) and abs(l_do - ref[:, 0].mean()) > 1e-6


def _ds_model(seed):
    from zeronotesamba_amd.models.models import DS_CNN
    sd = {k[len("anchor."):]: v for k, v in cnn_ref.synth_state_dict(seed, 2.0).items() if k.startswith("anchor.")}
    m = DS_CNN()
    m.load_state_dict(sd)
    return m.to(DEV), sd


def test_clmr_against_the_reference(golden):
    """pt_task "clmr" (pretext.py:494-511, 567-583) against vectors captured from the REAL reference's DS_CNN + NTXent +
    Adam(lr=1e-5) (tests/golden/clmr_eval.npz): the fused PretextTrainer (one stack over the 2B rows) step by step --
    losses, embeddings, gradients, parameters after steps 1 and 3 -- then the module-level train_epoch / val_epoch
    return values.  Gradients additionally at 1e-4 against the oracle with the device's decisions replayed."""
    from oracle import step_ref
    from replay_util import device_decisions, grad_err
    from zeronotesamba_amd.models.loss_functions import NTXent
    from zeronotesamba_amd.pretext import PretextTrainer, train_epoch, val_epoch
    g = golden("clmr_eval.npz")
    B, T, tau, lr = int(g["B"]), int(g["T"]), float(g["tau"]), float(g["lr"])
    ref = g["losses"]
    m, sd = _ds_model(int(g["seed"]))
    tr = PretextTrainer(m, batch_len=B, temperature=tau, lr=lr, train=False)
    assert tr.pt_task == "clmr"
    for it in range(3):
        batch = cnn_ref.synth_vqt(210 + it, (B, 2, 96, T))
        x = tr.stack_inputs(batch)
        if it == 0:
            replay, emb = device_decisions(x, tr.P, B, stack_B=2 * B)
            np.testing.assert_allclose(emb[:B].cpu().numpy(), g["s0_anc_emb"], atol=1e-4)
            np.testing.assert_allclose(emb[B:].cpu().numpy(), g["s0_pos_emb"], atol=1e-4)
        out3 = tr.step(x).tolist()
        tol = 2e-5 if it == 0 else 5e-4        # sign-like early Adam steps at lr 1e-5 amplify rounding noise in g
        np.testing.assert_allclose(out3, ref[it], atol=tol)
        if it == 0:
            op = step_ref.OracleParams(sd, lr)
            step_ref.clmr_step(op, batch, B, tau, step_opt=False, replay=replay)
            og = op.grads()
            for k, v in m.named_parameters():
                _check_stats(v.grad, g[f"s0_grad_{k}_stats"], g[f"s0_grad_{k}_samples"], rtol=1e-2, srtol=0.1)
                assert grad_err(k, v.grad, og) <= 1e-4, (k, grad_err(k, v.grad, og))
        if it in (0, 2):
            for k, v in m.named_parameters():
                _check_stats(v, g[f"s{it}_param_{k}_stats"], g[f"s{it}_param_{k}_samples"], rtol=1e-5 if it == 0 else 1e-4)
    vals = [tr.evaluate(tr.stack_inputs(cnn_ref.synth_vqt(220 + it, (B, 2, 96, T)))).tolist() for it in range(2)]
    np.testing.assert_allclose(vals, g["val_after"], atol=5e-4)
    # module-level epoch functions, the reference's own call structure
    loader = [[cnn_ref.synth_vqt(210 + it, (B, 2, 96, T))] for it in range(3)]
    vloader = [[cnn_ref.synth_vqt(220 + it, (B, 2, 96, T))] for it in range(2)]
    crit = NTXent(B, tau)
    m, _ = _ds_model(int(g["seed"]))
    m.pretrained.dp.p = 0.0
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    _, l, p, n = train_epoch(m, loader, crit, opt, pt_task="clmr")
    np.testing.assert_allclose([l, p, n], ref.mean(axis=0), atol=5e-4)
    np.testing.assert_allclose(val_epoch(m, vloader, crit, opt, pt_task="clmr"), g["val_after"].mean(axis=0), atol=5e-4)


def test_rccl_bucketed_allreduce_on_the_launch_streams():
    """The data-parallel step on a REAL RCCL group (one rank: this box has one GPU).  The bucketed async all-reduces
    run on RCCL's stream between the backward kernels; with one rank they must leave the step bitwise unchanged,
    which checks the stream ordering (reduce after the layer's gradient kernels, Adam after the reduce)."""
    import socket
    import torch.distributed as dist
    from zeronotesamba_amd.distributed import GradBucketReducer
    from zeronotesamba_amd.pretext import PretextTrainer
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        B, T = 4, 64
        ta = PretextTrainer(make_model(3), batch_len=B, lr=1e-4, seed=3)
        tb = PretextTrainer(make_model(3), batch_len=B, lr=1e-4, seed=3)
        tb.reducer = GradBucketReducer(tb.flat_g, tb.buckets, None, always=True)
        assert len(tb.buckets) >= 3 and tb.buckets[-1][2] == tb.n_flat
        x = ta.stack_inputs(cnn_ref.synth_vqt(77, (B, 2, 96, T)))
        for _ in range(3):
            oa, ob = ta.step(x), tb.step(x)
        assert torch.equal(oa, ob) and torch.equal(ta.flat_g, tb.flat_g) and torch.equal(ta.flat_p, tb.flat_p)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("train", [False, True])
def test_two_rank_trainer_replicas_stay_identical_and_equal_the_accumulated_step(tmp_path, train):
    """SURVEY 8e on the REAL trainer: two rank processes (fresh interpreters, tests/dp_worker.py) that start from
    DIFFERENT weights and step three times on DIFFERENT batches must (a) hold bit-identical parameters and Adam moments
    afterwards (construction-time broadcast + every step's all-reduce), and (b) equal a one-process run that starts
    from rank 0's weights, accumulates the two ranks' batch gradients and applies the fused Adam with grad_scale = 1/2
    -- the semantics pretext.py:475-492 has per rank with the mean gradient (global batch = 2 x 16 shifts, local
    NT-Xent per clip).  train=True: dropout on, each rank draws its own masks (the seed folds the rank in)."""
    import socket
    import subprocess
    import sys
    import os
    import dp_worker as W
    from zeronotesamba_amd.models.models import Pretext_CNN
    from zeronotesamba_amd.pretext import PretextTrainer
    steps = 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), str(tmp_path), str(int(train)), str(steps)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    assert r0["world"] == r1["world"] == 2 and r0["step_count"] == r1["step_count"] == steps and r0["buckets"] >= 3
    # replicas start from rank 0's weights whatever they were built with ...
    m = Pretext_CNN()
    m.load_state_dict(W.rank_state_dict(0))
    tr = PretextTrainer(m.to(DEV), batch_len=W.B, temperature=W.TAU, lr=W.LR, seed=W.SEED, train=train)
    assert torch.equal(r0["p0"], tr.flat_p.cpu()) and torch.equal(r1["p0"], r0["p0"])
    # ... (a) and end bit-identical
    for k in ("p", "m", "v"):
        assert torch.equal(r0[k], r1[k]), k
    assert not torch.equal(r0["out"], r1["out"])                  # they did see different batches
    # (b) one process, the two batches accumulated, Adam on half the sum
    for st in range(steps):
        tr.rank = 0
        o0 = tr.step(tr.stack_inputs(W.rank_batch(st, 0)), update=False)
        g0 = tr.flat_g.clone()
        tr.rank = 1
        tr.micro_count = 0                                         # rank 1's first (only) micro-batch of this step, not rank 0's second
        o1 = tr.step(tr.stack_inputs(W.rank_batch(st, 1)), update=False)
        tr.flat_g += g0
        tr.rank = 0
        tr.apply_update(0.5)
        assert torch.equal(o0.cpu(), r0["out"][st]) and torch.equal(o1.cpu(), r1["out"][st]), st
    for k, mine in (("p", tr.flat_p), ("m", tr.flat_m), ("v", tr.flat_v)):
        d = (mine.cpu() - r0[k]).abs().max().item()
        print(f"two ranks vs one accumulated process, flat_{k}: max |diff| {d:.3e}, bitwise {torch.equal(mine.cpu(), r0[k])}")
        assert d <= 1e-6, (k, d)
    assert (tr.flat_p.cpu() - r0["p0"]).abs().max().item() > 0.5 * W.LR      # and the weights did move


def test_checkpoint_round_trip_through_the_fused_trainer(tmp_path):
    """pretext.py:409-412 / loader.py:25-27 / sample_script.py:42: `torch.save(model.state_dict())` of a model whose
    parameters are views into the trainer's flat buffer, loaded into a fresh Pretext_CNN, into a model that already
    HAS a trainer, and into Down_CNN().pretext."""
    import os
    from zeronotesamba_amd.models.models import Down_CNN, Pretext_CNN
    from zeronotesamba_amd.pretext import PretextTrainer
    B, T = 2, 72
    m = make_model(5)
    tr = PretextTrainer(m, batch_len=B, lr=1e-3, train=False)
    x = tr.stack_inputs(cnn_ref.synth_vqt(210, (B, 2, 96, T)))
    tr.step(x)
    tr.step(x)
    path = tmp_path / "shift_pret_cnn_16.pth"
    torch.save(m.state_dict(), path)                                           # pretext.py:410
    n_par = sum(p.numel() for p in m.parameters())
    size = os.path.getsize(path)
    assert n_par == 26819202 and n_par * 4 <= size < n_par * 4 * 1.01, size    # 107 MB, not 36 copies of the flat buffer
    e_tr = tr.evaluate(x)
    with torch.no_grad():
        a0, p0 = m.eval()(x[:B], x[B:])
    # (a) fresh model
    f = Pretext_CNN()
    f.load_state_dict(torch.load(path, map_location="cpu"))
    f.to(DEV).eval()
    with torch.no_grad():
        a1, p1 = f(x[:B], x[B:])
    assert torch.equal(a0, a1) and torch.equal(p0, p1)
    # (b) a model that already has a trainer: the views stay views of the flat buffer, cached packed operands follow
    m2 = make_model(9)
    tr2 = PretextTrainer(m2, batch_len=B, lr=1e-3, train=False)
    from zeronotesamba_amd import _ops
    with _ops.frozen_weights():                    # operands cached across the two evaluations: the load must invalidate them
        e_before = tr2.evaluate(x)
        m2.load_state_dict(torch.load(path, map_location=DEV))
        tr2._check_flat()
        lo, hi = tr2.flat_p.data_ptr(), tr2.flat_p.data_ptr() + tr2.n_flat * 4
        assert all(lo <= p.data_ptr() < hi for p in m2.parameters())
        e_after = tr2.evaluate(x)
    assert not torch.equal(e_before, e_after) and torch.equal(e_after, e_tr)
    o_a, o_b = tr.step(x), tr2.step(x)                                         # and training continues from there
    assert torch.equal(o_a, o_b)
    # (c) Down_CNN().pretext.load_state_dict (loader.py:27, sample_script.py:42)
    d = Down_CNN()
    d.pretext.load_state_dict(torch.load(path, map_location="cpu"))
    d.to(DEV).eval()
    with torch.no_grad():
        comb = d(x[:B], x[B:])
    assert torch.equal(comb, torch.maximum(a0, p0))


def test_shift_bank_batches_use_one_index_table():
    """ShiftBank.batches draws every clip's starts up front (as pretext.py:308-318 does before the first step) and
    uploads them as one table: same crops, same generator state as calling batch() clip by clip."""
    import random
    from zeronotesamba_amd.pretext import ShiftBank
    g = torch.Generator().manual_seed(3)
    bank = torch.randn((5, 2, 96, 120), generator=g)
    sb = ShiftBank(bank, batch_len=4, crop=60, device=DEV)
    sb.shuffle_(np.random.RandomState(1))
    r1, r2 = random.Random(9), random.Random(9)
    got = list(sb.batches([3, 0, 4], rng=r1))
    want = [sb.batch(c, rng=r2) for c in (3, 0, 4)]
    assert r1.getstate() == r2.getstate() and len(got) == 3
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert list(sb.batches([], rng=r1)) == []
    with pytest.raises(IndexError):
        list(sb.batches([5]))
