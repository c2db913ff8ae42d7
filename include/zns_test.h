/*
 * zns_test.h -- TEST / BENCH ONLY entry points of libzns_hip.so.
 *
 * Not part of the drop-in boundary (include/zns.h is): process-global kernel-family switches, the measurement probe
 * bench.py arms around the dominant kernel, device self-tests and host-side tables the CPU tests compare with the oracle.
 * Nothing here is called by zeronotesamba_amd's product modules except through bench.py and tests/.
 */
#ifndef ZNS_TEST_H
#define ZNS_TEST_H

#include "zns.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Never call it from a serving or training process.  Selects the kernel family for every later conv
 * fwd/dgrad/wgrad call of the PROCESS (process-wide on purpose: autograd runs backward() on its own thread, which must
 * launch the family a validation test selected on the Python thread).  Returns the previous value. */
int zns_set_impl(int impl);

/* Tuning and diagnostic knobs of tests/ and tools/ (process-global, like zns_set_impl).  The library reads NO environment
 * variable: a stray variable in a training job cannot change a summation order or skip a store.  Every key's default (0,
 * KSPLIT: -1) is the product behaviour.  zns_test_set_tuning returns ZNS_OK or ZNS_ERR_ARG (unknown key). */
enum {
    ZNS_TUNE_KSPLIT = 0,      /* cv7 / cv8 input-channel split: -1 by the rule of T (default), 0 never, 2 / 4 / 8 forced */
    ZNS_TUNE_IPW,             /* two-kind spectral layout: (sample, slot) items per workgroup, 0 = chosen by the launcher;
                               * -2: spectral GEMM launches walk their items in item order instead of slot-major (A/B timing) */
    ZNS_TUNE_XQT_PLAN,        /* per-stage pyramid: chain lengths as decimal digits (223 = chains of 2, 2, 3 levels), 0 = default */
    ZNS_TUNE_XQT_LDS_PAD,     /* exact pyramid: pad the chain kernel's LDS to this many KiB (caps occupancy) */
    ZNS_TUNE_XQT_TPW,         /* per-stage filter bank: tiles per wave, 0 = default */
    ZNS_TUNE_XQT_CHUNKS,      /* fused generate_XQT kernel: runs a clip is cut into, 0 = default */
    ZNS_TUNE_XQT_DBG,         /* fused kernel: phase-skipping bits (WRONG RESULTS: timing diagnosis only) */
    ZNS_TUNE_XQT_STAMPS,      /* fused kernel: record s_memtime stamps of workgroup 0 (zns_xqt_debug_stamps) */
    ZNS_TUNE_WIN128,          /* cv2 / cv3 / cv4 spectral window: 0 by the launcher's rule, 1 force 128-point, -1 force 64-point */
    ZNS_TUNE_XQT_WAVES16,     /* fused generate_XQT kernel: 0 = 16 waves (two filter waves per SIMD, default), -1 = the 12-wave layout */
    ZNS_TUNE_ONE_ROW_DIRECT,  /* cv7 / cv8: 0 = frequency domain (one GEMM per branch and slot, default), 1 = the direct kernels */
    ZNS_TUNE_COUNT
};
int zns_test_set_tuning(int key, int value);
int zns_test_get_tuning(int key);

/* One probe per process: arming while a probe is armed, or disarming from a thread other than the
 * one that armed it, returns ZNS_ERR_ARG.
 * Measurement hook for bench.py: record hipEvents immediately around the MAIN kernel of the next
 * conv launches that match (kind, cin, cout, kh).  kind: 0 forward, 1 backward-data, 2 backward-weights.
 * ev_start/ev_stop: host arrays of n hipEvent_t (created with timing enabled); launch i uses slot
 * i % n.  n == 0 disarms.  Returns the number of matching launches recorded so far via zns_probe_count(). */
int zns_probe_arm(int kind, int cin, int cout, int kh, void* const* ev_start, void* const* ev_stop, int n);
int zns_probe_count(void);

/* HOST-only helper (no GPU touched): the folded per-octave FIR bank the kernel applies, as interleaved
 * re/im floats [octave][12][n_fft[octave]], and the 32+31 decimation taps.  Used by the CPU tests. */
int zns_xqt_host_bank(int mode, int* n_fft8, float* fir, size_t fir_floats, double* taps63);

/* The pyramid divides by float32(sqrt(0.5)) as `(float)((double)x * (1 / c))`; this runs that against the
 * float32 division over ALL 2^32 float32 bit patterns.  mismatches (device, 8 bytes) receives the count of differing
 * results, first_bad_bits (device, 4 bytes) the smallest differing input's bits (0xFFFFFFFF: none). */
int zns_selftest_div_sqrt_half(unsigned long long* mismatches, unsigned* first_bad_bits, void* stream);

/* Tuning aid: with ZNS_TUNE_XQT_STAMPS set (zns_test_set_tuning) the fused generate_XQT kernel records s_memtime stamps of its
 * workgroup 0 (steps 64..95 x 12 waves x 8 stamps); this copies the last launch's stamps to `host_out` (n <= 3072 values)
 * after synchronising the device.  Returns ZNS_ERR_ARG when no stamped launch has run. */
int zns_xqt_debug_stamps(unsigned long long* host_out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* ZNS_TEST_H */
