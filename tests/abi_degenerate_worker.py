"""Child process of tests/test_gpu_layers.py::test_compute_entry_points_refuse_degenerate_extents.

For every compute entry point of include/zns.h: one valid small call (rc 0), then the same call with each integer
argument replaced by 0 and by -1.  Extents (batch, time, channels, rows, counts ...) must be REFUSED with an error
code and a message; flags (layer index, accumulate, transposed ...) must still succeed.  Nothing may fault: the process
exits 0 and prints "ok" only if every call returned."""
import ctypes as C
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch

from zeronotesamba_amd import _lib

L = _lib.lib()
DEV = torch.device("cuda", 0)
PP = C.c_void_p * 2
calls = 0


def dp(t):
    return C.c_void_p(t.data_ptr() if t is not None else None)


def f32(*shape):
    return torch.randn(shape, device=DEV, dtype=torch.float32) * 0.1


def sweep(name, make, extents, flags):
    """make(ov) -> rc with integer arguments overridden by the dict ov"""
    global calls
    rc = make({})
    torch.cuda.synchronize()
    assert rc == 0, (name, "valid call", rc, L.zns_last_error())
    for key in list(extents) + list(flags):
        for bad in (0, -1):
            if key in flags and bad in flags[key]:
                want_ok = True
            elif key in flags:
                continue
            else:
                want_ok = False
            rc = make({key: bad})
            torch.cuda.synchronize()
            calls += 1
            if want_ok:
                assert rc == 0, (name, key, bad, rc, L.zns_last_error())
            else:
                assert rc != 0 and len(L.zns_last_error()) > 0, (name, key, bad, "was accepted")


def conv():
    nbr, B, T, cin, cout, kh, kw, H, pool = 2, 2, 40, 64, 64, 7, 13, 96, 3
    x = f32(nbr * B, cin, H, T)
    w = [f32(cout, cin, kh, kw) for _ in range(nbr)]
    b = [f32(cout) for _ in range(nbr)]
    y = torch.empty((nbr * B, cout, H // pool, T), device=DEV)
    am = torch.empty((nbr * B, cout, H // pool, T), device=DEV, dtype=torch.uint8)
    wsb = L.zns_conv2d_workspace_bytes(nbr, B, T, cin, cout, kh, kw, H)
    ws = torch.empty(wsb, device=DEV, dtype=torch.uint8)
    dz = f32(nbr * B, cout, H, T)
    dx = torch.empty((nbr * B, cin, H, T), device=DEV)
    dw = [torch.zeros_like(t) for t in w]
    db = [torch.zeros_like(t) for t in b]
    g = dict(nbr=nbr, B=B, T=T, cin=cin, cout=cout, kh=kh, kw=kw, H=H)
    geo = list(g)

    def fwd(ov):
        a = {**g, "pool": pool, "layer": 1, **ov}
        return L.zns_conv2d_fwd_f32(*[a[k] for k in geo], a["pool"], dp(x), None, PP(*[t.data_ptr() for t in w]),
                                    PP(*[t.data_ptr() for t in b]), dp(y), dp(am), C.c_float(0.1), 5, a["layer"],
                                    None, None, dp(ws), ws.numel(), None)

    def dgrad(ov):
        a = {**g, "prev_pool": 1, **ov}
        return L.zns_conv2d_dgrad_f32(*[a[k] for k in geo], dp(dz), PP(*[t.data_ptr() for t in w]), dp(dx), dp(x),
                                      C.c_float(1.0), None, a["prev_pool"], None, None, dp(ws), ws.numel(), None)

    def wgrad(ov):
        a = {**g, "accumulate": 1, **ov}
        return L.zns_conv2d_wgrad_f32(*[a[k] for k in geo], dp(dz), dp(x), None, PP(*[t.data_ptr() for t in dw]),
                                      PP(*[t.data_ptr() for t in db]), a["accumulate"], None, None, dp(ws), ws.numel(),
                                      None)

    lid = C.c_int(0)
    nb = L.zns_conv2d_packed_bytes(nbr, B, T, cin, cout, kh, kw, H, 0, C.byref(lid))
    assert nb > 0 and lid.value != 0
    dst = torch.empty(nb, device=DEV, dtype=torch.uint8)

    def pack(ov):
        a = {**g, "transposed": 0, **ov}
        return L.zns_conv2d_pack_f32(*[a[k] for k in geo], a["transposed"], PP(*[t.data_ptr() for t in w]), dp(dst),
                                     dst.numel(), None)

    for impl in (_lib.ZNS_IMPL_AUTO, _lib.ZNS_IMPL_DIRECT, _lib.ZNS_IMPL_NAIVE):
        prev = L.zns_set_impl(impl)
        try:
            sweep(f"conv2d_fwd[{impl}]", fwd, geo + ["pool"], {"layer": (0, -1)})
            sweep(f"conv2d_dgrad[{impl}]", dgrad, geo + ["prev_pool"], {})
            sweep(f"conv2d_wgrad[{impl}]", wgrad, geo, {"accumulate": (0, -1)})
        finally:
            L.zns_set_impl(prev)
    sweep("conv2d_pack", pack, geo, {"transposed": (0,)})
    # a workspace one byte short is an error, not an overrun
    rc = L.zns_conv2d_fwd_f32(nbr, B, T, cin, cout, kh, kw, H, pool, dp(x), None, PP(*[t.data_ptr() for t in w]),
                              PP(*[t.data_ptr() for t in b]), dp(y), dp(am), C.c_float(0.1), 5, 1, None, None, dp(ws), 16,
                              None)
    assert rc != 0, "16-byte workspace accepted"


def small():
    nbr, B, T, Cc = 2, 3, 50, 128
    h = f32(nbr * B, Cc, 1, T)
    w = [f32(1, Cc) for _ in range(nbr)]
    b = [f32(1) for _ in range(nbr)]
    emb = torch.empty((nbr * B, T), device=DEV)
    demb = f32(nbr * B, T)
    dz = torch.empty_like(h)
    dw = [torch.zeros_like(t) for t in w]
    db = [torch.zeros_like(t) for t in b]
    ws = torch.empty(1 << 20, device=DEV, dtype=torch.uint8)
    g = dict(nbr=nbr, B=B, T=T, C=Cc)

    def head_fwd(ov):
        a = {**g, **ov}
        return L.zns_head_fwd_f32(a["nbr"], a["B"], a["T"], a["C"], dp(h), PP(*[t.data_ptr() for t in w]),
                                  PP(*[t.data_ptr() for t in b]), dp(emb), None)

    def head_bwd(ov):
        a = {**g, "accumulate": 0, **ov}
        return L.zns_head_bwd_f32(a["nbr"], a["B"], a["T"], a["C"], dp(h), dp(emb), dp(demb), None,
                                  PP(*[t.data_ptr() for t in w]), C.c_float(1.0), dp(dz), PP(*[t.data_ptr() for t in dw]),
                                  PP(*[t.data_ptr() for t in db]), a["accumulate"], dp(ws), ws.numel(), None)

    sweep("head_fwd", head_fwd, list(g), {})
    sweep("head_bwd", head_bwd, list(g), {"accumulate": (0, -1)})

    a_, p_ = f32(4, 99), f32(4, 99)
    out = torch.empty_like(a_)

    def down_fwd(ov):
        a = {"reduction": 0, **ov}
        return L.zns_down_fwd_f32(dp(a_), dp(p_), a_.numel(), a["reduction"], dp(out), None)

    def down_bwd(ov):
        a = {"reduction": 1, **ov}
        return L.zns_down_bwd_f32(dp(a_), dp(p_), dp(out), a_.numel(), a["reduction"], dp(a_.clone()), dp(p_.clone()), None)

    sweep("down_fwd", down_fwd, [], {"reduction": (0,)})
    sweep("down_bwd", down_bwd, [], {"reduction": (0,)})
    assert L.zns_down_fwd_f32(dp(a_), dp(p_), 0, 0, dp(out), None) == 0          # an empty tensor is a no-op
    assert L.zns_down_fwd_f32(dp(a_), dp(p_), a_.numel(), 7, dp(out), None) != 0

    rows, D = 16, 512
    ea, ep = f32(rows, D), f32(rows, D)
    out3 = torch.empty(3, device=DEV)
    ga, gp = torch.empty_like(ea), torch.empty_like(ep)
    nws = torch.empty(L.zns_ntxent_workspace_bytes(rows), device=DEV, dtype=torch.uint8)

    def ntx(ov):
        a = {"rows": rows, "D": D, "batch_len": 16, **ov}
        return L.zns_ntxent_f32(dp(ea), dp(ep), a["rows"], a["D"], a["batch_len"], C.c_float(0.1), dp(out3), None, dp(ga), dp(gp),
                                C.c_float(1.0), dp(nws), nws.numel(), None)

    sweep("ntxent", ntx, ["rows", "D", "batch_len"], {})
    assert L.zns_ntxent_f32(dp(ea), dp(ep), rows, D, 16, C.c_float(0.1), dp(out3), None, dp(ga), dp(gp), C.c_float(1.0), dp(nws), 64,
                            None) != 0
    assert L.zns_ntxent_f32(dp(ea), dp(ep), 2049, D, 4096, C.c_float(0.1), dp(out3), None, None, None, C.c_float(1.0), dp(nws),
                            nws.numel(), None) != 0

    o, t = torch.rand(1, 77, device=DEV), (torch.rand(1, 77, device=DEV) > 0.5).float()
    loss, dout = torch.empty(1, device=DEV), torch.empty_like(o)
    assert L.zns_bce_f32(dp(o), dp(t), 77, dp(loss), dp(dout), C.c_float(1.0), None) == 0
    assert L.zns_bce_f32(dp(o), dp(t), 0, dp(loss), dp(dout), C.c_float(1.0), None) != 0
    # zns_scale_f32: null pointers with a non-zero count are refused, empty arrays are a no-op
    assert L.zns_scale_f32(dp(loss), dp(o), dp(dout), 77, None, None, 0, None) == 0
    assert L.zns_scale_f32(dp(loss), dp(o), dp(dout), 0, None, None, 0, None) == 0
    assert L.zns_scale_f32(None, dp(o), dp(dout), 77, None, None, 0, None) != 0
    assert L.zns_scale_f32(dp(loss), None, dp(dout), 77, None, None, 0, None) != 0
    assert L.zns_scale_f32(dp(loss), dp(o), dp(dout), 77, None, None, 5, None) != 0

    p, gr, m, v = f32(1001), f32(1001), torch.zeros(1001, device=DEV), torch.zeros(1001, device=DEV)

    def adam(ov):
        a = {"step": 1, **ov}
        return L.zns_adam_step_f32(dp(p), dp(gr), dp(m), dp(v), 1001, C.c_float(1e-3), C.c_float(0.9), C.c_float(0.999),
                                   C.c_float(1e-8), a["step"], C.c_float(1.0), None)

    sweep("adam_step", adam, ["step"], {})
    assert L.zns_adam_step_f32(dp(p), dp(gr), dp(m), dp(v), 0, C.c_float(1e-3), C.c_float(0.9), C.c_float(0.999),
                               C.c_float(1e-8), 1, C.c_float(1.0), None) == 0   # zero parameters: nothing to do
    table = torch.tensor([[p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), 1001]], dtype=torch.int64, device=DEV)

    def adam_multi(ov):
        a = {"n_tensors": 1, "step": 2, **ov}
        return L.zns_adam_multi_f32(dp(table), a["n_tensors"], 1001, C.c_float(1e-3), C.c_float(0.9), C.c_float(0.999),
                                    C.c_float(1e-8), a["step"], C.c_float(1.0), None)

    sweep("adam_multi", adam_multi, ["n_tensors", "step"], {})
    assert L.zns_adam_multi_f32(None, 1, 1001, C.c_float(1e-3), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), 1,
                                C.c_float(1.0), None) != 0


def online():
    bank = f32(3, 2, 96, 64)
    clip = torch.tensor([0, 2, 1, 1], dtype=torch.int32, device=DEV)
    start = torch.tensor([0, 5, 32, 17], dtype=torch.int32, device=DEV)
    out = torch.empty((2, 4, 96, 32), device=DEV)

    def gather(ov):
        a = {"bank_T": 64, "n_items": 4, "T": 32, **ov}
        return L.zns_shift_gather_f32(dp(bank), a["bank_T"], dp(clip), dp(start), a["n_items"], a["T"], dp(out), None)

    sweep("shift_gather", gather, ["bank_T", "n_items", "T"], {})

    n = 40000
    stems = [f32(n) for _ in range(3)]
    mix = torch.empty(n, device=DEV)
    SP = C.c_void_p * 3

    def stem_mix(ov):
        a = {"n_stems": 3, "n": n, **ov}
        return L.zns_stem_mix_f32(SP(*[t.data_ptr() for t in stems]), a["n_stems"], a["n"], dp(mix), None)

    sweep("stem_mix", stem_mix, ["n_stems", "n"], {})
    assert L.zns_stem_mix_f32(SP(*[t.data_ptr() for t in stems]), 9, n, dp(mix), None) != 0

    ns = 16000
    table = torch.tensor([[2, 100, stems[0].data_ptr(), stems[1].data_ptr(), stems[2].data_ptr()],
                          [1, 0, stems[1].data_ptr(), stems[2].data_ptr(), 0]], dtype=torch.int64, device=DEV)
    staged = torch.empty((2, 2, ns), device=DEV)

    def stage(ov):
        a = {"n_files": 2, "max_other": 2, "n_samples": ns, **ov}
        return L.zns_stage_crops_f32(dp(table), a["n_files"], a["max_other"], a["n_samples"], dp(staged), None)

    sweep("stage_crops", stage, ["n_files", "max_other", "n_samples"], {})

    cnt = torch.zeros(2, dtype=torch.int32, device=DEV)
    offs = torch.tensor([0, 2 * ns], dtype=torch.int64, device=DEV)

    def gate(ov):
        a = {"n_cand": 2, "n_samples": ns, **ov}
        return L.zns_rms_gate_f32(dp(staged), C.c_void_p(staged.data_ptr() + 4 * ns), dp(offs), a["n_cand"],
                                  a["n_samples"], None, None, dp(cnt), None)

    sweep("rms_gate", gate, ["n_cand", "n_samples"], {})
    assert L.zns_rms_gate_f32(dp(staged), dp(staged), dp(offs), 2, ns, None, None, None, None) != 0


def xqt():
    n_clips, n = 3, 33333
    wav = f32(n_clips, n)
    out = torch.empty((n_clips, 96, L.zns_xqt_frames(n)), device=DEV)
    ws = torch.empty(L.zns_xqt_workspace_bytes(n_clips, n, 0), device=DEV, dtype=torch.uint8)

    def run(ov):
        a = {"n_clips": n_clips, "n_samples": n, "sample_rate": 16000, "stages": 3, **ov}
        return L.zns_xqt_stages_f32(dp(wav), a["n_clips"], a["n_samples"], a["sample_rate"], a.get("mode", 0), dp(out), dp(ws),
                                    ws.numel(), a["stages"], None)

    sweep("xqt_stages", run, ["n_clips", "n_samples", "sample_rate", "stages"], {"mode": (0,)})
    assert run({"mode": 2}) == _lib.ZNS_ERR_MODE and run({"mode": -1}) == _lib.ZNS_ERR_MODE
    assert run({"n_samples": 40}) != 0                                      # shorter than the reflect padding
    assert run({"stages": 16}) != 0 and run({"stages": 3 | 4 | 8}) != 0
    assert L.zns_xqt_f32(dp(wav), n_clips, n, 16000, 0, dp(out), dp(ws), 1024, None) != 0


if __name__ == "__main__":
    conv()
    small()
    online()
    xqt()
    torch.cuda.synchronize()
    print(f"{calls} degenerate calls refused or served without a fault")
    print("ok")
