"""ctypes binding of libzns_hip.so (C ABI declared in include/zns.h).

There is deliberately NO fallback: if the shared library is missing or a call
fails, a RuntimeError is raised.  Build it with ``python -m zeronotesamba_amd.build``
(hipcc --offload-arch=gfx950), which is what ``__graft_entry__.build()`` does.
"""
import ctypes as C  # noqa: N812 (also re-exported as _lib.C for byref/c_int users)
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzns_hip.so")

_lib: Optional[C.CDLL] = None

c_fp = C.c_void_p          # device pointers travel as integers
c_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/zns.h (the product surface) one to one
SIGNATURES = {
    "zns_last_error": (C.c_char_p, []),
    "zns_version": (C.c_int, []),
    "zns_conv2d_workspace_bytes": (C.c_size_t, [C.c_int] * 8),
    "zns_conv2d_spectrum_bytes": (C.c_size_t, [C.c_int] * 8),
    "zns_conv2d_dz_spectrum_bytes": (C.c_size_t, [C.c_int] * 8),
    "zns_conv2d_packed_bytes": (C.c_size_t, [C.c_int] * 9 + [C.POINTER(C.c_int)]),
    "zns_conv2d_pack_f32": (C.c_int, [C.c_int] * 9 + [c_pp, c_fp, C.c_size_t, c_fp]),
    "zns_conv2d_fwd_f32": (C.c_int, [C.c_int] * 9 + [c_fp, c_fp, c_pp, c_pp, c_fp, c_fp, C.c_float, C.c_uint32, C.c_int,
                                                     c_fp, c_fp, c_fp, C.c_size_t, c_fp]),
    "zns_conv2d_dgrad_f32": (C.c_int, [C.c_int] * 8 + [c_fp, c_pp, c_fp, c_fp, C.c_float, c_fp, C.c_int,
                                                       c_fp, c_fp, c_fp, C.c_size_t, c_fp]),
    "zns_conv2d_wgrad_f32": (C.c_int, [C.c_int] * 8 + [c_fp, c_fp, c_fp, c_pp, c_pp, C.c_int, c_fp, c_fp, c_fp,
                                                       C.c_size_t, c_fp]),
    "zns_head_fwd_f32": (C.c_int, [C.c_int] * 4 + [c_fp, c_pp, c_pp, c_fp, c_fp]),
    "zns_head_bwd_f32": (C.c_int, [C.c_int] * 4 + [c_fp, c_fp, c_fp, c_fp, c_pp, C.c_float, c_fp, c_pp, c_pp, C.c_int,
                                                   c_fp, C.c_size_t, c_fp]),
    "zns_down_fwd_f32": (C.c_int, [c_fp, c_fp, C.c_size_t, C.c_int, c_fp, c_fp]),
    "zns_down_bwd_f32": (C.c_int, [c_fp, c_fp, c_fp, C.c_size_t, C.c_int, c_fp, c_fp, c_fp]),
    "zns_ntxent_workspace_bytes": (C.c_size_t, [C.c_int]),
    "zns_ntxent_f32": (C.c_int, [c_fp, c_fp, C.c_int, C.c_int, C.c_int, C.c_float, c_fp, c_fp, c_fp, c_fp, C.c_float, c_fp,
                                 C.c_size_t, c_fp]),
    "zns_scale_f32": (C.c_int, [c_fp, c_fp, c_fp, C.c_size_t, c_fp, c_fp, C.c_size_t, c_fp]),
    "zns_bce_f32": (C.c_int, [c_fp, c_fp, C.c_size_t, c_fp, c_fp, C.c_float, c_fp]),
    "zns_adam_step_f32": (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_int, C.c_float, c_fp]),
    "zns_adam_multi_f32": (C.c_int, [c_fp, C.c_int, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                     C.c_float, c_fp]),
    "zns_xqt_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "zns_xqt_frames": (C.c_int, [C.c_int]),
    "zns_xqt_f32": (C.c_int, [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, c_fp, C.c_size_t, c_fp]),
    "zns_xqt_stages_f32": (C.c_int, [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, c_fp, C.c_size_t, C.c_int, c_fp]),
    "zns_shift_gather_f32": (C.c_int, [c_fp, C.c_int, c_fp, c_fp, C.c_int, C.c_int, c_fp, c_fp]),
    "zns_stem_mix_f32": (C.c_int, [c_fp, C.c_int, C.c_longlong, c_fp, c_fp]),
    "zns_stage_crops_f32": (C.c_int, [c_fp, C.c_int, C.c_int, C.c_int, c_fp, c_fp]),
    "zns_rms_frames": (C.c_int, [C.c_int]),
    "zns_rms_gate_f32": (C.c_int, [c_fp, c_fp, c_fp, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp]),
}

# include/zns_test.h: test / bench only (kernel-family switch, measurement probe, self-tests, host tables)
TEST_SIGNATURES = {
    "zns_set_impl": (C.c_int, [C.c_int]),
    "zns_test_set_tuning": (C.c_int, [C.c_int, C.c_int]),
    "zns_test_get_tuning": (C.c_int, [C.c_int]),
    "zns_probe_arm": (C.c_int, [C.c_int] * 4 + [c_pp, c_pp, C.c_int]),
    "zns_probe_count": (C.c_int, []),
    "zns_xqt_host_bank": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_size_t,
                                    C.POINTER(C.c_double)]),
    "zns_selftest_div_sqrt_half": (C.c_int, [c_fp, c_fp, c_fp]),
    "zns_xqt_debug_stamps": (C.c_int, [C.POINTER(C.c_ulonglong), C.c_size_t]),
}

ZNS_IMPL_AUTO, ZNS_IMPL_NAIVE, ZNS_IMPL_DIRECT = 0, 1, 2
ZNS_REDUCE_MAX, ZNS_REDUCE_MEAN = 0, 1
ZNS_MODE_VQT, ZNS_MODE_CQT = 0, 1
ZNS_XQT_SEPARATE, ZNS_XQT_FUSED, ZNS_XQT_EXACT = 4, 8, 16
ZNS_ERR_MODE = -3
# include/zns_test.h: keys of zns_test_set_tuning (tests / tools only; the library reads no environment variable)
(ZNS_TUNE_KSPLIT, ZNS_TUNE_IPW, ZNS_TUNE_XQT_PLAN, ZNS_TUNE_XQT_LDS_PAD, ZNS_TUNE_XQT_TPW, ZNS_TUNE_XQT_CHUNKS,
 ZNS_TUNE_XQT_DBG, ZNS_TUNE_XQT_STAMPS, ZNS_TUNE_WIN128, ZNS_TUNE_XQT_WAVES16, ZNS_TUNE_ONE_ROW_DIRECT, ZNS_TUNE_COUNT) = range(12)


def lib() -> C.CDLL:
    """Load the HIP library (once).  Raises RuntimeError when it is absent: no CPU path exists."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"zeronotesamba_amd: {LIB_PATH} not found. The MI355X kernels are the only implementation; "
                "build them with `python -m zeronotesamba_amd.build` (needs hipcc).")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(TEST_SIGNATURES.items()):
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared entry point
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().zns_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")


def ptr_array(ptrs):
    """Host array of device pointers (the `const float* const*` parameters)."""
    arr = (C.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr
