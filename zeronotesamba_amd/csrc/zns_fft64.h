// Real 64-point FFT / inverse FFT in registers (one transform per lane), for the frequency-domain conv path.
// Half-size complex radix-2 DIF (output bit-reversed, resolved at compile time) + real split post-pass.
// Validated against numpy in tests (through spec_fwd_kernel / spec_inv_kernel) and prototyped in float64.
#pragma once
#include <hip/hip_runtime.h>

namespace zns {

// cos / sin (2 pi k / 64)
__device__ constexpr float FFT_C[64] = {1.0f, 0.99518472f, 0.980785251f, 0.956940353f, 0.923879504f, 0.881921291f, 0.831469595f, 0.773010433f, 0.707106769f, 0.634393275f, 0.555570245f, 0.471396744f, 0.382683426f, 0.290284663f, 0.195090324f, 0.0980171412f, 0.0f, -0.0980171412f, -0.195090324f, -0.290284663f, -0.382683426f, -0.471396744f, -0.555570245f, -0.634393275f, -0.707106769f, -0.773010433f, -0.831469595f, -0.881921291f, -0.923879504f, -0.956940353f, -0.980785251f, -0.99518472f, -1.0f, -0.99518472f, -0.980785251f, -0.956940353f, -0.923879504f, -0.881921291f, -0.831469595f, -0.773010433f, -0.707106769f, -0.634393275f, -0.555570245f, -0.471396744f, -0.382683426f, -0.290284663f, -0.195090324f, -0.0980171412f, 0.0f, 0.0980171412f, 0.195090324f, 0.290284663f, 0.382683426f, 0.471396744f, 0.555570245f, 0.634393275f, 0.707106769f, 0.773010433f, 0.831469595f, 0.881921291f, 0.923879504f, 0.956940353f, 0.980785251f, 0.99518472f};
__device__ constexpr float FFT_S[64] = {0.0f, 0.0980171412f, 0.195090324f, 0.290284663f, 0.382683426f, 0.471396744f, 0.555570245f, 0.634393275f, 0.707106769f, 0.773010433f, 0.831469595f, 0.881921291f, 0.923879504f, 0.956940353f, 0.980785251f, 0.99518472f, 1.0f, 0.99518472f, 0.980785251f, 0.956940353f, 0.923879504f, 0.881921291f, 0.831469595f, 0.773010433f, 0.707106769f, 0.634393275f, 0.555570245f, 0.471396744f, 0.382683426f, 0.290284663f, 0.195090324f, 0.0980171412f, 0.0f, -0.0980171412f, -0.195090324f, -0.290284663f, -0.382683426f, -0.471396744f, -0.555570245f, -0.634393275f, -0.707106769f, -0.773010433f, -0.831469595f, -0.881921291f, -0.923879504f, -0.956940353f, -0.980785251f, -0.99518472f, -1.0f, -0.99518472f, -0.980785251f, -0.956940353f, -0.923879504f, -0.881921291f, -0.831469595f, -0.773010433f, -0.707106769f, -0.634393275f, -0.555570245f, -0.471396744f, -0.382683426f, -0.290284663f, -0.195090324f, -0.0980171412f};

__host__ __device__ constexpr int bitrev5(int k) {
    return ((k & 1) << 4) | ((k & 2) << 2) | (k & 4) | ((k & 8) >> 2) | ((k & 16) >> 4);
}

// in-place radix-2 DIF on 32 complex points, natural order in, bit-reversed order out; SIGN -1 forward, +1 inverse
template <int SIGN>
__device__ __forceinline__ void cfft32(float (&re)[32], float (&im)[32]) {
#pragma unroll
    for (int lg = 4; lg >= 0; --lg) {
        const int s = 1 << lg;
#pragma unroll
        for (int g = 0; g < 32; g += 2 * s) {
#pragma unroll
            for (int j = 0; j < s; ++j) {
                const int i = g + j, t = j * (32 / s);   // twiddle angle 2 pi j / (2 s) = 2 pi t / 64
                const float ar = re[i], ai = im[i], br = re[i + s], bi = im[i + s];
                re[i] = ar + br;
                im[i] = ai + bi;
                const float dr = ar - br, di = ai - bi;
                if (t == 0) {
                    re[i + s] = dr;
                    im[i + s] = di;
                } else {
                    const float c = FFT_C[t], sn = SIGN * FFT_S[t];
                    re[i + s] = dr * c - di * sn;
                    im[i + s] = dr * sn + di * c;
                }
            }
        }
    }
}

// x[64] real -> slots: sr[0] = X[0], si[0] = X[32] (both real), (sr[k], si[k]) = X[k], k = 1..31 (unnormalised DFT)
__device__ __forceinline__ void rfft64(const float (&x)[64], float (&sr)[32], float (&si)[32]) {
    float re[32], im[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        re[m] = x[2 * m];
        im[m] = x[2 * m + 1];
    }
    cfft32<-1>(re, im);
    sr[0] = re[0] + im[0];
    si[0] = re[0] - im[0];
#pragma unroll
    for (int k = 1; k < 32; ++k) {
        const float a = re[bitrev5(k)], b = im[bitrev5(k)], c = re[bitrev5(32 - k)], d = im[bitrev5(32 - k)];
        const float er = 0.5f * (a + c), ei = 0.5f * (b - d);
        const float orr = 0.5f * (b + d), oi = -0.5f * (a - c);
        const float wr = FFT_C[k], wi = -FFT_S[k];
        sr[k] = er + (wr * orr - wi * oi);
        si[k] = ei + (wr * oi + wi * orr);
    }
}

// inverse of rfft64 (exact inverse, includes the 1/64): slots -> x[64]
__device__ __forceinline__ void irfft64(const float (&sr)[32], const float (&si)[32], float (&x)[64]) {
    float re[32], im[32];
    re[0] = 0.5f * (sr[0] + si[0]);
    im[0] = 0.5f * (sr[0] - si[0]);
#pragma unroll
    for (int k = 1; k < 32; ++k) {
        const float p = sr[k], q = si[k], r = sr[32 - k], s = si[32 - k];
        const float er = 0.5f * (p + r), ei = 0.5f * (q - s);
        const float dr = 0.5f * (p - r), di = 0.5f * (q + s);
        const float c = FFT_C[k], sn = FFT_S[k];
        const float orr = dr * c - di * sn, oi = dr * sn + di * c;
        re[k] = er - oi;
        im[k] = ei + orr;
    }
    cfft32<1>(re, im);
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        x[2 * m] = re[bitrev5(m)] * (1.0f / 32.0f);
        x[2 * m + 1] = im[bitrev5(m)] * (1.0f / 32.0f);
    }
}

}  // namespace zns
