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
// (1 ulp each: <= 2e-6 absolute in the log domain, a fiftieth of the parity tolerance, a fifth of the instructions of the
// correctly rounded library forms)
__device__ __forceinline__ float xqt_log_magnitude(float re, float im) {
    return __logf(__builtin_amdgcn_sqrtf(re * re + im * im) + 1e-9f);
}

// The fused kernel (zns_xqt_stream.hip): both stages in ONE launch, pyramid levels never leave LDS.  vqt bank only.
bool xqt_stream_supported(const Bank* b, int n_clips, int n_samples);
unsigned long long* xqt_stream_stamps();   // tuning aid, see ZNS_XQT_STAMPS
int xqt_stream_launch(const Bank* b, const float* wav, int n_clips, int n_samples, float* out, hipStream_t st);

}  // namespace zns
