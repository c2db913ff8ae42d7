"""Full BASELINE sizes on the device (B=16 per branch, T in {512, 313, 626}): the CPU oracle would take minutes
here, so the check is between INDEPENDENT device implementations of the same step -- the frequency-domain MFMA
path (default), the direct MFMA path and (forward only, T=313) the naive one-thread-per-output kernels -- plus
size-independent properties: dropout keep-rate, gradient accumulation = 2x, determinism (bitwise repeatability)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def build(seed=0):
    from zeronotesamba_amd.models.models import Pretext_CNN
    torch.manual_seed(seed)
    m = Pretext_CNN()
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("weight"):
                p.mul_(2.0)
    return m.to(DEV)


def synth(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((2 * B, 1, 96, T), generator=g) * 3.0 - 6.0
    return x.clamp_(-20.72, 3.0).to(DEV)


def run(impl, model, x, B, drop_p=0.0, seed=0, backward=True, replay=None, want_saved=False):
    """One forward (+ backward) on kernel family `impl`.  replay = (acts, argm, emb) of ANOTHER family's forward: the backward
    then takes that forward's decisions (arg-max maps, zero patterns) and saved activations, so the two families' gradients
    differ by their convolution arithmetic alone."""
    from zeronotesamba_amd import _lib, _ops
    P = _ops.StackParams.from_flat(model.anchor.stack_params() + model.postve.stack_params(), 2)
    prev = _lib.lib().zns_set_impl(impl)
    try:
        with torch.no_grad():
            emb, acts, argm = _ops.stack_forward(x, P, B, drop_p, seed, save=True)
            saved = (acts, argm, emb)
            if replay is not None:
                acts, argm, emb = replay
                argm = {k: v for k, v in argm.items() if k < 100}     # arg-max maps only: the other family's saved spectra stay out
            grads = None
            if backward:
                demb = torch.cos(torch.arange(emb.numel(), device=DEV, dtype=torch.float32) * 0.37).reshape(emb.shape)
                grads = [torch.zeros_like(p) for p in P.flat()]
                G = _ops.StackParams.from_flat(grads, 2)
                _ops.stack_backward(x, P, B, acts, argm, emb, demb, None, 1.0 / (1.0 - drop_p), G)
        torch.cuda.synchronize()
    finally:
        _lib.lib().zns_set_impl(prev)
    if want_saved:
        return emb, acts, grads, saved
    return emb, acts, grads


def default_against_direct(model, x, B, T):
    """The default (frequency-domain) path against the direct MFMA kernels: embeddings and every layer's activations at the
    north_star tolerance; gradients twice -- each family on its OWN decisions (flip tolerant: max-pool / ReLU near-ties,
    DESIGN.md section 6: one flipped element moves a weight gradient by ~1e-2 of its norm), and the strict one: the direct
    kernels' backward on the default forward's decisions and activations, where only the two families' convolution
    arithmetic is left (1e-4 of every parameter gradient's norm)."""
    from zeronotesamba_amd import _lib
    e_f, a_f, g_f, saved_f = run(_lib.ZNS_IMPL_AUTO, model, x, B, want_saved=True)
    e_d, a_d, g_d = run(_lib.ZNS_IMPL_DIRECT, model, x, B)
    assert tuple(e_f.shape) == (2 * B, T)
    assert (e_f - e_d).abs().max().item() < 1e-4
    for li in range(8):
        scale = a_d[li].abs().max().item()
        assert (a_f[li] - a_d[li]).abs().max().item() <= 1e-4 * scale, f"layer {li + 1} activations"
    for gf, gd in zip(g_f, g_d):
        rel = ((gf - gd).norm() / gd.norm().clamp_min(1e-30)).item()
        assert rel < 3e-2, rel
    _, _, g_r = run(_lib.ZNS_IMPL_DIRECT, model, x, B, replay=saved_f)
    worst = 0.0
    for gf, gr in zip(g_f, g_r):
        worst = max(worst, ((gf - gr).norm() / gr.norm().clamp_min(1e-30)).item())
    print("B=%d T=%d: default vs direct backward on the same decisions, worst per-parameter gradient rel L2 %.2e" % (B, T, worst))
    assert worst <= 1e-4, worst
    return e_f, a_f, g_f


@pytest.mark.parametrize("T", [512, 313, 626])
def test_spectral_vs_direct_full_size(T):
    from zeronotesamba_amd import _lib
    B = 16
    model = build()
    x = synth(B, T, 1)
    e_f, a_f, g_f = default_against_direct(model, x, B, T)
    # bitwise repeatability (split-K slabs and spectra are reduced in a fixed order, no float atomics)
    e2, _, g2 = run(_lib.ZNS_IMPL_AUTO, model, x, B)
    assert torch.equal(e_f, e2)
    assert all(torch.equal(u, v) for u, v in zip(g_f, g2))


def test_naive_forward_agrees_at_reference_tile():
    from zeronotesamba_amd import _lib
    B, T = 16, 313                                                     # pretext.py:285
    model = build(1)
    x = synth(B, T, 2)
    e_f, _, _ = run(_lib.ZNS_IMPL_AUTO, model, x, B, backward=False)
    e_n, _, _ = run(_lib.ZNS_IMPL_NAIVE, model, x, B, backward=False)
    assert (e_f - e_n).abs().max().item() < 1e-4


def test_dropout_rate_and_accumulation_full_size():
    from zeronotesamba_amd import _lib, _ops
    B, T = 16, 512
    model = build(2)
    x = synth(B, T, 3)
    _, a_eval, _ = run(_lib.ZNS_IMPL_AUTO, model, x, B, backward=False)
    _, a_train, _ = run(_lib.ZNS_IMPL_AUTO, model, x, B, drop_p=0.1, seed=77, backward=False)
    y_e, y_t = a_eval[0], a_train[0]                                   # cv1 output: same input, so the masks compare directly
    pos = y_e > 0
    kept = (y_t != 0) & pos
    rate = kept.sum().item() / pos.sum().item()
    assert abs(rate - 0.9) < 2e-3
    ratio = (y_t[kept] / y_e[kept])
    assert (ratio - 1.0 / 0.9).abs().max().item() < 1e-5               # 1/(1-p) rescale
    # gradient accumulation: accumulate=True twice == 2x
    P = _ops.StackParams.from_flat(model.anchor.stack_params() + model.postve.stack_params(), 2)
    with torch.no_grad():
        emb, acts, argm = _ops.stack_forward(x, P, B, 0.0, 0, save=True)
        demb = torch.ones_like(emb)
        g1 = [torch.zeros_like(p) for p in P.flat()]
        _ops.stack_backward(x, P, B, acts, argm, emb, demb, None, 1.0, _ops.StackParams.from_flat(g1, 2))
        g2 = [g.clone() for g in g1]
        _ops.stack_backward(x, P, B, acts, argm, emb, demb, None, 1.0, _ops.StackParams.from_flat(g2, 2), accumulate=True)
    for u, v in zip(g1, g2):
        assert ((v - 2 * u).norm() / u.norm().clamp_min(1e-30)).item() < 1e-6


@pytest.mark.parametrize("B,T", [(1, 1876), (2, 939)])
def test_finetune_shape_long_clip_segmented_rows(B, T):
    """BASELINE config 4 shape: batch of one 30 s clip (T=1876, epochs.py:48-63).  Rows longer than 12 windows are cut
    into segments that run as separate spectral samples; checked against the direct MFMA kernels (activations,
    gradients) and the naive kernels (embeddings) at a long, non-multiple-of-32 T."""
    from zeronotesamba_amd import _lib
    model = build(3)
    x = synth(B, T, 4)
    e_a, a_a, g_a = default_against_direct(model, x, B, T)
    e_n, _, _ = run(_lib.ZNS_IMPL_NAIVE, model, x, B, backward=False)
    assert (e_a - e_n).abs().max().item() < 1e-4
    # the segmented path is the frequency-domain one: a saved input spectrum exists for cv2..cv6
    assert _lib.lib().zns_conv2d_spectrum_bytes(2, B, T, 128, 128, 9, 17, 32) > 0


@pytest.mark.parametrize("B,T", [(1, 32), (3, 45), (2, 100), (5, 577), (1, 1000), (2, 529)])
def test_ragged_shapes_spectral_vs_direct(B, T):
    """Ragged batch sizes and row lengths: T below one window (32), odd T (no 16-byte alignment anywhere), T just past
    a segment boundary (cv2: 12 windows of 52 = 624 > 577 > 11*52; cv6: 577 = 13.1 windows -> 2 segments), T = 529 =
    one sample past 12 windows of cv6 (44*12 = 528) and a long row with 3 segments.  Default (frequency-domain) path
    against the direct MFMA kernels: activations of every layer, embeddings, every gradient."""
    model = build(11)
    x = synth(B, T, 100 + T)
    default_against_direct(model, x, B, T)


def test_clmr_style_shared_branch_module_api():
    """pretext.py:494-511 ("clmr"): ONE DS_CNN applied to two views; its weight gradient is the sum of both passes."""
    from zeronotesamba_amd.models.loss_functions import NTXent
    from zeronotesamba_amd.models.models import DS_CNN
    torch.manual_seed(0)
    m = DS_CNN().to(DEV).eval()
    B, T = 4, 64
    v = synth(B, T, 5)
    a = m(v[:B])
    p = m(v[B:])
    loss, _, _ = NTXent(B, 0.25)(a, p)
    loss.backward()
    g_both = m.pretrained.cv3.weight.grad.clone()
    m.zero_grad()
    a = m(v[:B])
    loss2, _, _ = NTXent(B, 0.25)(a, p.detach())
    loss2.backward()
    g_a = m.pretrained.cv3.weight.grad.clone()
    m.zero_grad()
    p = m(v[B:])
    loss3, _, _ = NTXent(B, 0.25)(a.detach(), p)
    loss3.backward()
    g_p = m.pretrained.cv3.weight.grad.clone()
    assert ((g_both - (g_a + g_p)).norm() / g_both.norm()).item() < 1e-5


@pytest.mark.parametrize("T", [512, 313, 626])
def test_full_size_step_against_the_cpu_oracle(T):
    """BASELINE config 3 at its full size (B=16 pairs; T=512 the benchmarked tile, T=313 the crop the reference trains
    on -- pretext.py:285 -- and T=626 a whole bank row; at 313 and 626 cv7 / cv8 take the K-split kernels), dropout
    off: the fused trainer's loss, cosine statistics,
    embeddings and every parameter gradient against the CPU oracle (stock torch ops, numerically the reference:
    tests/test_oracle.py).  One oracle step takes ~12 s on the box's 16 host cores; two are run:
      (1) mask replay -- the oracle takes the device's max-pool / ReLU decisions (tests/replay_util.py): every
          per-parameter gradient must agree to 1e-4 of its L2 norm (north_star tolerance);
      (2) the oracle on its own decisions: flip-tolerant 3e-2 (one flipped element reroutes a gradient column)."""
    from conftest import host_threads
    from oracle import step_ref
    from replay_util import device_decisions, flip_summary, grad_err, rel_l2, total_flips
    from zeronotesamba_amd import _ops
    from zeronotesamba_amd.pretext import PretextTrainer
    B, tau = 16, 0.25
    torch.set_num_threads(host_threads())
    model = build(2)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    batch = (torch.randn((B, 2, 96, T), generator=g) * 3.0 - 6.0).clamp_(-20.72, 3.0)
    tr = PretextTrainer(model, batch_len=B, temperature=tau, lr=1e-6, train=False)
    x = tr.stack_inputs(batch)
    out3 = tr.step(x, update=False).tolist()
    replay, emb = device_decisions(x, tr.P, B)            # same kernels, bitwise repeatable: the decisions step() took
    # ---- (1) mask replay: 1e-4 on every parameter gradient
    op = step_ref.OracleParams(sd, 1e-6)
    flips = {}
    loss, cpos, cneg, ra, rp = step_ref.pretext_step(op, batch, B, tau, drop_seed=None, step_opt=False, replay=replay,
                                                     flips=flips)
    assert abs(out3[0] - loss) < 2e-5 and abs(out3[1] - cpos) < 2e-5 and abs(out3[2] - cneg) < 2e-5
    assert (emb.cpu() - torch.cat([ra, rp])).abs().max().item() < 1e-4
    ref = op.grads()
    print("full-size step T=%d vs CPU oracle with the device's decisions replayed: loss %.6f/%.6f" % (T, out3[0], loss))
    print("  decisions overridden:", flip_summary(flips), "(total %d)" % total_flips(flips))
    worst = 0.0
    for name, p in model.named_parameters():
        rel = grad_err(name, p.grad, ref)
        worst = max(worst, rel)
        print(f"  {name:36s} rel {rel:.2e}  |ref| {ref[name].norm().item():.3e}")
        assert rel <= 1e-4, (name, rel)
    print("  worst per-parameter gradient rel L2: %.2e" % worst)
    # ---- (2) the oracle's own decisions: flip tolerant
    op2 = step_ref.OracleParams(sd, 1e-6)
    loss2, cpos2, cneg2, ra2, rp2 = step_ref.pretext_step(op2, batch, B, tau, drop_seed=None, step_opt=False)
    assert abs(out3[0] - loss2) < 2e-5 and abs(out3[1] - cpos2) < 2e-5 and abs(out3[2] - cneg2) < 2e-5
    assert (emb.cpu() - torch.cat([ra2, rp2])).abs().max().item() < 1e-4
    ref2 = op2.grads()
    # Gradients: ReLU / max-pool decisions on values within fp32 summation noise of a tie differ between ANY two fp32
    # implementations.  ONE flipped ReLU element at cv8 moves that branch's cv8 weight gradient by ~1e-2 of its L2 norm
    # and everything below it by 4e-3..9e-3, while tensors above the flip (fc1) agree to 4e-5.  Hence 3e-2 here.
    for name, p in model.named_parameters():
        assert rel_l2(p.grad, ref2[name]) < 3e-2, name
    assert rel_l2(model.anchor.fc1.weight.grad, ref2["anchor.fc1.weight"]) < 2e-4


@pytest.mark.parametrize("red", ["max", "mean"])
def test_finetune_step_full_size_against_the_cpu_oracle(red):
    """BASELINE config 4 at its full size (epochs.py:48-63 through loader.load_models("pretrained", "finetune"),
    loader.py:43): Down_CNN(max / mean), batch of ONE 30 s clip (T = 1876: every spectral layer runs NS > 1 row
    segments), the 120-BPM {0, 0.5, 1} mask bench.py uses, BCELoss, Adam(lr = 0.5 * 1e-5 * 10e-2).  Whole step against
    the CPU oracle (one oracle step is ~3 s on the box's host cores) with the device's max-pool / ReLU decisions
    replayed: loss 2e-5, output 1e-4, every parameter gradient <= 1e-4 of its L2 norm, post-Adam parameters 1e-5 --
    and, because an update of lr = 5e-7 hides inside that 1e-5, the UPDATE itself: Adam's first step moves a parameter
    by lr * g / (|g| + eps), so device and oracle may only differ where a gradient element is within their 1e-4
    disagreement of zero."""
    import contextlib
    import io
    from conftest import host_threads
    from oracle import cnn_ref, step_ref
    from replay_util import device_decisions, flip_summary, grad_err, total_flips
    from zeronotesamba_amd import _ops
    from zeronotesamba_amd.loader import load_models
    T, lr = 1876, 0.5 * 1e-5 * 10e-2
    torch.set_num_threads(host_threads())
    sd = cnn_ref.synth_state_dict(5, 2.0)      # with these weights the anchor branch wins a quarter of the frames of max()
    with contextlib.redirect_stdout(io.StringIO()):
        crit, opt, d = load_models("pretrained", "finetune", 1e-5, state_dict=sd, device=DEV)
    d.reduction = red
    d.eval()                                            # dropout off: the oracle's masks are then the ReLU decisions only
    assert abs(opt.param_groups[0]["lr"] - lr) < 1e-18
    vqt = cnn_ref.synth_vqt(300 + T, (2, 96, T))
    msk = torch.zeros(T)
    msk[5::31] = 1.0
    msk[4::31] = 0.5
    msk[6::31] = 0.5
    v1, v2 = vqt[0].reshape(1, 1, 96, T).to(DEV), vqt[1].reshape(1, 1, 96, T).to(DEV)
    before = {k: v.detach().cpu().clone() for k, v in d.named_parameters()}
    opt.zero_grad()
    o = d(v1, v2)
    loss = crit(o, msk.reshape(1, T).to(DEV))
    loss.backward()
    grads = {k: v.grad.detach().cpu().clone() for k, v in d.named_parameters()}
    opt.step()
    torch.cuda.synchronize()
    P = _ops.StackParams.from_flat(d.pretext.anchor.stack_params() + d.pretext.postve.stack_params(), 2)
    # the decisions the step took: same kernels on the pre-update weights are bitwise repeatable, so re-run on a copy
    d0 = type(d)(reduction=red)
    d0.pretext.load_state_dict(sd)
    d0.to(DEV).eval()
    P0 = _ops.StackParams.from_flat(d0.pretext.anchor.stack_params() + d0.pretext.postve.stack_params(), 2)
    replay, emb0 = device_decisions(torch.cat([v1, v2]), P0, 1)
    op = step_ref.OracleParams({"pretext." + k: v for k, v in sd.items()}, lr)
    flips = {}
    rl, ro = step_ref.finetune_step(op, vqt, msk, reduction=red, step_opt=True, replay=replay, flips=flips)
    print(f"fine-tune step T={T} ({red}) vs CPU oracle, decisions replayed: loss {loss.item():.6f}/{rl:.6f};",
          flip_summary(flips), "(total %d)" % total_flips(flips))
    assert abs(loss.item() - rl) < 2e-5
    assert (o.detach().cpu() - ro).abs().max().item() < 1e-4
    ref = op.grads()
    assert all(float(g.abs().max()) > 0 for g in grads.values())          # both branches receive gradient
    worst = 0.0
    for k in grads:
        e = grad_err(k, grads[k], ref)
        worst = max(worst, e)
        assert e <= 1e-4, (k, e)
    print("  worst per-parameter gradient rel L2: %.2e" % worst)
    after_ref = op.values()
    n_all = n_bad = 0
    for k, v in d.named_parameters():
        got, want = v.detach().cpu(), after_ref[k]
        # 1e-5 of the tensor's scale -- but never tighter than one sign flip of a near-zero gradient element (2 lr: Adam's
        # first step is lr * sign(g)), which the count below bounds to 0.1 % of the elements
        worst_p = (got - want).abs().max().item()
        assert worst_p <= max(1e-5 * max(want.abs().max().item(), 1e-3), 2.2 * lr), (k, worst_p)
        # the update: both sides moved by lr * g / (|g| + eps); elements whose gradient is within the 1e-4 disagreement
        # of zero may take the other sign (2 * lr apart), everything else must agree to a tenth of lr
        dd, dr = (got - before[k]).double(), (want - before[k]).double()
        bad = ((dd - dr).abs() > 0.1 * lr)
        n_all += bad.numel()
        n_bad += int(bad.sum())
        assert 0.5 * lr < float(dd.abs().max()) <= 1.1 * lr, k          # fp32 rounding of p - step: one ulp of p
    print("  Adam update: %d of %d elements differ by more than lr/10" % (n_bad, n_all))
    assert n_bad <= 1e-3 * n_all


@pytest.mark.parametrize("B_big,T", [(80, 313), (48, 512)])
def test_a_large_batch_equals_its_chunks_of_sixteen(B_big, T):
    """Batch sizes beyond BASELINE's 16 (spectra of several GiB: 64-bit offsets, descriptors rebuilt per sample): with
    dropout off a clip's activations do not depend on its batch, so the embeddings must be bit for bit those of the
    same clips in chunks of 16, and the parameter gradients the sum over the chunks up to fp32 reassociation
    (tools/diag_large.py runs the same check up to 400 clips per branch)."""
    from zeronotesamba_amd import _ops
    model = build()
    x = synth(B_big, T, 4)
    P = _ops.StackParams.from_flat(model.anchor.stack_params() + model.postve.stack_params(), 2)
    demb_all = torch.randn((2 * B_big, T), generator=torch.Generator().manual_seed(9)).to(DEV)

    def fb(xx, B, demb):
        with torch.no_grad():
            emb, acts, argm = _ops.stack_forward(xx, P, B, 0.0, 0, save=True)
            grads = [torch.zeros_like(p) for p in P.flat()]
            _ops.stack_backward(xx, P, B, acts, argm, emb, demb.contiguous(), None, 1.0, _ops.StackParams.from_flat(grads, 2))
        torch.cuda.synchronize()
        return emb, grads

    e_big, g_big = fb(x, B_big, demb_all)
    g_sum, ea, ep = None, [], []
    for i in range(0, B_big, 16):
        xc = torch.cat([x[i:i + 16], x[B_big + i:B_big + i + 16]]).contiguous()
        dc = torch.cat([demb_all[i:i + 16], demb_all[B_big + i:B_big + i + 16]])
        e, g = fb(xc, 16, dc)
        ea.append(e[:16]); ep.append(e[16:])
        g_sum = g if g_sum is None else [a + b for a, b in zip(g_sum, g)]
    assert torch.equal(e_big, torch.cat(ea + ep))
    for a, b in zip(g_big, g_sum):
        assert ((a - b).norm() / b.norm().clamp_min(1e-30)).item() < 1e-4
