"""CPU: the C-ABI library loads and exports every symbol include/zns.h (product) and include/zns_test.h (test / bench only)
declare; host-only entry points behave (no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def header_symbols(name="zns.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zns_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from zeronotesamba_amd import _lib
    L = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"libzns_hip.so does not export {s}"
    # and the ctypes table binds exactly the header's surface
    assert sorted(_lib.SIGNATURES.keys()) == syms
    # the test / bench surface is a separate header and a separate table; nothing of it leaks into the product header
    tsyms = header_symbols("zns_test.h")
    assert sorted(_lib.TEST_SIGNATURES.keys()) == tsyms and not set(tsyms) & set(syms)
    for s in tsyms:
        assert hasattr(L, s), f"libzns_hip.so does not export {s}"
    for word in ("zns_set_impl", "zns_probe", "selftest", "host_bank", "debug", "tuning", "ZNS_TUNE"):
        assert word not in open(os.path.join(ROOT, "include", "zns.h")).read()


def test_library_reads_no_environment_variable():
    """The product library takes its tuning knobs through include/zns_test.h only: no getenv, no ZNS_* variable name in the
    binary (flag names inside error messages -- ZNS_XQT_SEPARATE ... -- are C macros of zns.h, not variables)."""
    import subprocess
    from zeronotesamba_amd import _lib
    nm = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True)
    assert nm.returncode == 0
    assert not re.search(r"\b(secure_)?getenv\b", nm.stdout), "libzns_hip.so imports getenv"
    for f in os.listdir(os.path.join(ROOT, "zeronotesamba_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(ROOT, "zeronotesamba_amd", "csrc", f)).read(), f
    L = _lib.lib()
    assert L.zns_test_get_tuning(_lib.ZNS_TUNE_KSPLIT) == -1 and L.zns_test_get_tuning(_lib.ZNS_TUNE_XQT_DBG) == 0
    assert L.zns_test_set_tuning(_lib.ZNS_TUNE_COUNT, 1) != 0 and L.zns_test_set_tuning(-1, 1) != 0
    assert L.zns_test_set_tuning(_lib.ZNS_TUNE_IPW, 3) == 0 and L.zns_test_get_tuning(_lib.ZNS_TUNE_IPW) == 3
    assert L.zns_test_set_tuning(_lib.ZNS_TUNE_IPW, 0) == 0


def test_host_only_entry_points():
    from zeronotesamba_amd import _lib
    L = _lib.lib()
    assert L.zns_version() >= 100
    assert L.zns_xqt_frames(160000) == 626 and L.zns_xqt_frames(80000) == 313   # pretext.py:255,285
    assert L.zns_xqt_frames(130816) == 512
    assert L.zns_conv2d_workspace_bytes(2, 16, 512, 128, 128, 9, 17, 32) > 2 * 128 * 128 * 153 * 4
    assert L.zns_set_impl(_lib.ZNS_IMPL_AUTO) in (0, 1)
    rc = L.zns_xqt_host_bank(7, None, None, 0, None)
    assert rc == _lib.ZNS_ERR_MODE and b"Mode can only be vqt or cqt!" in L.zns_last_error()


def test_size_queries_survive_degenerate_arguments():
    """Every *_bytes / *_frames query with an empty or negative extent answers (0 or the minimum) instead of dividing
    by it: a zero batch once raised SIGFPE inside zns_conv2d_workspace_bytes.  In a child process, so that a fault is a
    test failure and not the end of the test run."""
    import subprocess
    import sys
    code = r"""
import ctypes as C, itertools, sys
sys.path.insert(0, %r)
from zeronotesamba_amd import _lib
L = _lib.lib()
good = [2, 16, 512, 128, 128, 9, 17, 32]
lid = C.c_int(7)
for pos in range(8):
    for bad in (0, -1, -2147483648):
        a = list(good); a[pos] = bad
        assert L.zns_conv2d_workspace_bytes(*a) in (0, 256), (a,)
        assert L.zns_conv2d_spectrum_bytes(*a) == 0, (a,)
        assert L.zns_conv2d_dz_spectrum_bytes(*a) == 0, (a,)
        for tr in (0, 1):
            assert L.zns_conv2d_packed_bytes(*a, tr, C.byref(lid)) == 0 and lid.value == 0, (a,)
# geometries no kernel instance exists for: a size, never a fault
for cin, cout, kh, kw, H in itertools.product((1, 3, 64), (5, 64), (1, 2, 3), (1, 4, 11), (1, 7, 96)):
    for B, T in ((1, 1), (3, 31), (16, 512)):
        L.zns_conv2d_workspace_bytes(2, B, T, cin, cout, kh, kw, H)
        L.zns_conv2d_spectrum_bytes(2, B, T, cin, cout, kh, kw, H)
        L.zns_conv2d_dz_spectrum_bytes(2, B, T, cin, cout, kh, kw, H)
        L.zns_conv2d_packed_bytes(2, B, T, cin, cout, kh, kw, H, 0, None)
for v in (0, -1, -2147483648):
    assert L.zns_ntxent_workspace_bytes(v) == 256
    assert L.zns_xqt_workspace_bytes(v, 160000, 0) == 256 and L.zns_xqt_workspace_bytes(4, v, 0) == 256
    assert L.zns_xqt_frames(v) in (0, 1) and L.zns_rms_frames(v) in (0, 1)
assert L.zns_xqt_workspace_bytes(1, 1, 0) >= 4 and L.zns_xqt_workspace_bytes(1, 2147483647, 1) > 2**32
print("ok")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout[-500:], r.stderr[-2000:])


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from zeronotesamba_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="only implementation"):
        _lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "zeronotesamba_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports oracle"
