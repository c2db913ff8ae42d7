// Shared declarations of the generate_XQT kernels (zns_xqt.hip: per-stage launches; zns_xqt_stream.hip: the fused kernel).
#pragma once
#include "zns_common.h"

namespace zns {

constexpr int XQ_SR = 16000, XQ_HOP = 256, XQ_BINS = 96, XQ_BPO = 12, XQ_OCT = 8;
constexpr int NL = 32, NR = 31;   // decimation taps touched for the exact 2:1 ratio

struct DecTaps {
    double l[NL];   // taps applied to x[2t - i]
    double r[NR];   // taps applied to x[2t + 1 + k]
};
struct DecTapsF {
    float l[NL];    // the same rounded to float32; right wing: r[k] = l[k + 1]
};

struct Bank {
    int n_fft[XQ_OCT];
    int hop[XQ_OCT];
    int off[XQ_OCT];        // float2 offset of octave i inside `fir`
    float* fira = nullptr;  // device: MFMA A operand of the filter bank, [octave][n_fft / 2 k-steps][64 lanes]
    int offa[XQ_OCT];       // float offset of octave i inside `fira`
    float* fira16 = nullptr;   // device: A operand for v_mfma_f32_16x16x4_f32, [octave][row tile 0..1][n_fft / 4 k-steps][64 lanes]
    int offa16[XQ_OCT];     // float offset of octave i inside `fira16`
    DecTaps* taps = nullptr;   // device: float64 decimation taps (the exact kernels)
    float tapsf[NL];           // host: the 32 distinct taps rounded to float32 (r[k] = l[k + 1]); kernel argument of the default kernels
    bool ready = false;
};

int ensure_bank(int mode, Bank** out);

// resampy's `y_hat /= sqrt(0.5)`-in-float32 (librosa.resample: y_hat /= np.sqrt(ratio), float32 array / float32 scalar):
// the correctly rounded float32 quotient x / c, c = float32(sqrt(0.5)).  v_div_* is a ten-instruction sequence per
// output; instead: x * RN64(1 / c) in float64, rounded once to float32.  That IS the correctly rounded quotient: the
// float64 product is within 2^-52 (relative) of x / c, while a quotient of two 24-bit significands, c not a power of
// two, is never closer than 2^-48 to a float32 rounding boundary (|X 2^k - (2M + 1) C| >= 1 with C < 2^24).  Checked
// exhaustively over all 2^32 float32 inputs on the device: zns_selftest_div_sqrt_half (tests/test_vqt.py).
constexpr float DC_SQRT_HALF = 0.70710677f;
constexpr double DC_INV_SQRT_HALF = 1.0 / (double)DC_SQRT_HALF;
__device__ __forceinline__ float dc_div_sqrt_half(double x_float32_valued) {
    return (float)(x_float32_valued * DC_INV_SQRT_HALF);
}

// Epilogue of both filter-bank kernels: log(|V| + 1e-9) (input_rep.py:51-52, eps = 10e-10) with the hardware sqrt / log2
// (1 ulp each: <= 2e-6 absolute in the log domain, a fiftieth of the parity tolerance).  ln(m) = log2(m) * ln 2 is formed
// exactly as the compiler's own expansion of logf forms it -- the product in extended precision, c + cc = ln 2 to 48 bits --
// minus the part of that expansion that rescales denormal arguments: the argument here is never below 1e-9.  The epilogue
// is most of the filter-bank waves' vector instructions in the fused kernel (ten of these per wave and step): 11
// instructions instead of 22.
__device__ __forceinline__ float xqt_log_magnitude(float re, float im) {
    const float m = __builtin_amdgcn_sqrtf(re * re + im * im) + 1e-9f;
    const float y = __builtin_amdgcn_logf(m);                       // v_log_f32: log2
    constexpr float c = 0x1.62e42ep-1f, cc = 0x1.efa39ep-25f;       // ln 2 = c + cc
    const float r = y * c;
    const float lo = __builtin_fmaf(y, cc, __builtin_fmaf(y, c, -r));
    const float ln = __builtin_fmaf(c, y, lo);
    return __builtin_fabsf(y) < __builtin_inff() ? ln : y;          // log(inf) = inf, log(NaN) = NaN
}

// ---------------------------------------------------------------------------------------------------------
// The default pyramid's 63-tap chain, packed.  Order per output, as resampy: left wing i = 0..31 (x[2t - i]), right wing
// r = 0..30 (x[2t + 1 + r], weight l[r + 1]), one fused multiply-add per tap.  A non-packed v_fma_f32 issues at 4.2 cycles
// per wave and SIMD like every vector instruction (tools/ubench_valu.hip: 75 TFLOP/s; the 157 TFLOP/s vector peak is
// v_pk_fma_f32), so the taps are PACKED without changing any output's own order: outputs k (even) and k + 1 share a
// register pair, and tap i of output k reads the same sample as tap i + 2 of output k + 1 (x[2k - i] = x[2(k + 1) - (i + 2)]):
// one v_pk_fma_f32 with the tap pair (l[i], l[i + 2]) from scalar registers and that sample broadcast to both halves
// (op_sel).  Output k + 1 simply runs two taps ahead in the left wing and two behind in the right one; the four taps at
// each wing's ends that have no partner are single v_fma_f32.  67 instructions per output pair instead of 126, every
// rounding where it was (the results equal the plain v_fma_f32 chain bit for bit).
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct TapPairs {
    f32x2 p[30];                 // p[i] = (l[i], l[i + 2])
};
// acc.lo += t.lo * x[H], acc.hi += t.hi * x[H]
template <int H>
__device__ __forceinline__ void pk_left(f32x2& acc, const f32x2 t, const f32x2 x) {
    if (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(t), "v"(x));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(t), "v"(x));
}
// acc.lo += t.hi * x[H], acc.hi += t.lo * x[H]
template <int H>
__device__ __forceinline__ void pk_right(f32x2& acc, const f32x2 t, const f32x2 x) {
    if (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "+v"(acc) : "s"(t), "v"(x));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(t), "v"(x));
}
// 2 NP consecutive outputs from the thread's span of 64 + 4 NP samples: xp[q] = samples (2 q, 2 q + 1), sample n belongs to
// input index 2 t0 - 32 + n of the first output t0.  The chain comes in phases so that a caller can request its samples in
// the order of first use and start on the first while the last are still in flight:
//   open        the odd outputs' taps 0, 1                                        samples 33 .. 46
//   left<I0,I1> packed left-wing taps i = I0 .. I1 - 1                             samples 32 - (I1 - 1) .. 44 - I0   (+ 4 per pair)
//   turn        the even outputs' taps 30, 31 and their right-wing taps 0, 1       samples 1 .. 14 and 33 .. 46
//   right<R0,R1> packed right-wing taps r = R0 .. R1 - 1 (2 <= R0)                 samples 33 + R0 .. 45 + (R1 - 1)
//   close       the odd outputs' taps r = 29, 30; results out                      samples 64 .. 77
template <int NP>
struct Taps63 {
    f32x2 acc[NP];
    template <class XP>
    __device__ __forceinline__ void open(const XP& xp, const TapPairs& w) {
        const float l0 = w.p[0].x, l1 = w.p[1].x;
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            float hi = __builtin_fmaf(l0, xp[(34 + 4 * m) >> 1][0], 0.f);
            hi = __builtin_fmaf(l1, xp[(33 + 4 * m) >> 1][1], hi);
            acc[m] = f32x2{0.f, hi};
        }
    }
    template <int I0, int I1, class XP>
    __device__ __forceinline__ void left(const XP& xp, const TapPairs& w) {
#pragma unroll
        for (int i = I0; i < I1; ++i)
#pragma unroll
            for (int m = 0; m < NP; ++m) {
                const int n = 32 + 4 * m - i;                              // x[2k - i], k = 2 m
                if (n & 1) pk_left<1>(acc[m], w.p[i], xp[n >> 1]); else pk_left<0>(acc[m], w.p[i], xp[n >> 1]);
            }
    }
    template <class XP>
    __device__ __forceinline__ void turn(const XP& xp, const TapPairs& w) {
        const float l1 = w.p[1].x, l2 = w.p[0].y, l30 = w.p[28].y, l31 = w.p[29].y;
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            float lo = acc[m].x;
            lo = __builtin_fmaf(l30, xp[(2 + 4 * m) >> 1][0], lo);
            lo = __builtin_fmaf(l31, xp[(1 + 4 * m) >> 1][1], lo);
            lo = __builtin_fmaf(l1, xp[(33 + 4 * m) >> 1][1], lo);
            lo = __builtin_fmaf(l2, xp[(34 + 4 * m) >> 1][0], lo);
            acc[m].x = lo;
        }
    }
    template <int R0, int R1, class XP>
    __device__ __forceinline__ void right(const XP& xp, const TapPairs& w) {
#pragma unroll
        for (int r = R0; r < R1; ++r)
#pragma unroll
            for (int m = 0; m < NP; ++m) {
                const int n = 33 + 4 * m + r;                              // x[2k + 1 + r]: tap r of output k, tap r - 2 of output k + 1
                if (n & 1) pk_right<1>(acc[m], w.p[r - 1], xp[n >> 1]); else pk_right<0>(acc[m], w.p[r - 1], xp[n >> 1]);
            }
    }
    template <class XP>
    __device__ __forceinline__ void close(const XP& xp, const TapPairs& w, float (&out)[2 * NP]) {
        const float l30 = w.p[28].y, l31 = w.p[29].y;
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            float hi = acc[m].y;
            hi = __builtin_fmaf(l30, xp[(64 + 4 * m) >> 1][0], hi);
            hi = __builtin_fmaf(l31, xp[(65 + 4 * m) >> 1][1], hi);
            out[2 * m] = acc[m].x;
            out[2 * m + 1] = hi;
        }
        // unpinned, the optimiser sinks each output's chain into the branch that later keeps or zeroes it (serial chains)
#pragma unroll
        for (int k = 0; k < 2 * NP; ++k) asm volatile("" : "+v"(out[k]));
    }
};
template <int NP>
__device__ __forceinline__ void taps63_pairs(const f32x2 (&xp)[32 + 2 * NP], const TapPairs& w, float (&out)[2 * NP]) {
    Taps63<NP> t;
    t.open(xp, w);
    t.template left<0, 30>(xp, w);
    t.turn(xp, w);
    t.template right<2, NR>(xp, w);
    t.close(xp, w, out);
}

// The fused kernel (zns_xqt_stream.hip): both stages in ONE launch, pyramid levels never leave LDS.  vqt bank only.
bool xqt_stream_supported(const Bank* b, int n_clips, int n_samples);
unsigned long long* xqt_stream_stamps();   // tuning aid, see ZNS_XQT_STAMPS
int xqt_stream_launch(const Bank* b, const float* wav, int n_clips, int n_samples, float* out, hipStream_t st);

}  // namespace zns
