// Real 64- and 128-point FFT / inverse FFT in registers (one transform per lane), for the frequency-domain conv path.
// Half-size complex radix-2 DIF (output bit-reversed, resolved at compile time) + real split post-pass.
// Validated against numpy in tests (through spec_fwd_kernel / spec_inv_kernel) and prototyped in float64.
#pragma once
#include <hip/hip_runtime.h>

namespace zns {

// cos / sin (2 pi k / 128), float32 roundings of the exact values.  The 64-point transform reads every second entry
// (the same float values its own 64-entry table held: results are bit for bit those of the first version).
__device__ constexpr float FFT_C128[128] = {1.0f, 0.9987954497337341f, 0.9951847195625305f, 0.9891765117645264f, 0.9807852506637573f, 0.9700312614440918f, 0.9569403529167175f, 0.9415440559387207f, 0.9238795042037964f, 0.903989315032959f, 0.8819212913513184f, 0.8577286005020142f, 0.8314695954322815f, 0.803207516670227f, 0.7730104327201843f, 0.7409511208534241f, 0.7071067690849304f, 0.6715589761734009f, 0.6343932747840881f, 0.5956993103027344f, 0.5555702447891235f, 0.5141027569770813f, 0.4713967442512512f, 0.4275550842285156f, 0.3826834261417389f, 0.3368898630142212f, 0.290284663438797f, 0.24298018217086792f, 0.19509032368659973f, 0.1467304676771164f, 0.0980171412229538f, 0.049067676067352295f, 0.0f, -0.049067676067352295f, -0.0980171412229538f, -0.1467304676771164f, -0.19509032368659973f, -0.24298018217086792f, -0.290284663438797f, -0.3368898630142212f, -0.3826834261417389f, -0.4275550842285156f, -0.4713967442512512f, -0.5141027569770813f, -0.5555702447891235f, -0.5956993103027344f, -0.6343932747840881f, -0.6715589761734009f, -0.7071067690849304f, -0.7409511208534241f, -0.7730104327201843f, -0.803207516670227f, -0.8314695954322815f, -0.8577286005020142f, -0.8819212913513184f, -0.903989315032959f, -0.9238795042037964f, -0.9415440559387207f, -0.9569403529167175f, -0.9700312614440918f, -0.9807852506637573f, -0.9891765117645264f, -0.9951847195625305f, -0.9987954497337341f, -1.0f, -0.9987954497337341f, -0.9951847195625305f, -0.9891765117645264f, -0.9807852506637573f, -0.9700312614440918f, -0.9569403529167175f, -0.9415440559387207f, -0.9238795042037964f, -0.903989315032959f, -0.8819212913513184f, -0.8577286005020142f, -0.8314695954322815f, -0.803207516670227f, -0.7730104327201843f, -0.7409511208534241f, -0.7071067690849304f, -0.6715589761734009f, -0.6343932747840881f, -0.5956993103027344f, -0.5555702447891235f, -0.5141027569770813f, -0.4713967442512512f, -0.4275550842285156f, -0.3826834261417389f, -0.3368898630142212f, -0.290284663438797f, -0.24298018217086792f, -0.19509032368659973f, -0.1467304676771164f, -0.0980171412229538f, -0.049067676067352295f, 0.0f, 0.049067676067352295f, 0.0980171412229538f, 0.1467304676771164f, 0.19509032368659973f, 0.24298018217086792f, 0.290284663438797f, 0.3368898630142212f, 0.3826834261417389f, 0.4275550842285156f, 0.4713967442512512f, 0.5141027569770813f, 0.5555702447891235f, 0.5956993103027344f, 0.6343932747840881f, 0.6715589761734009f, 0.7071067690849304f, 0.7409511208534241f, 0.7730104327201843f, 0.803207516670227f, 0.8314695954322815f, 0.8577286005020142f, 0.8819212913513184f, 0.903989315032959f, 0.9238795042037964f, 0.9415440559387207f, 0.9569403529167175f, 0.9700312614440918f, 0.9807852506637573f, 0.9891765117645264f, 0.9951847195625305f, 0.9987954497337341f};
__device__ constexpr float FFT_S128[128] = {0.0f, 0.049067676067352295f, 0.0980171412229538f, 0.1467304676771164f, 0.19509032368659973f, 0.24298018217086792f, 0.290284663438797f, 0.3368898630142212f, 0.3826834261417389f, 0.4275550842285156f, 0.4713967442512512f, 0.5141027569770813f, 0.5555702447891235f, 0.5956993103027344f, 0.6343932747840881f, 0.6715589761734009f, 0.7071067690849304f, 0.7409511208534241f, 0.7730104327201843f, 0.803207516670227f, 0.8314695954322815f, 0.8577286005020142f, 0.8819212913513184f, 0.903989315032959f, 0.9238795042037964f, 0.9415440559387207f, 0.9569403529167175f, 0.9700312614440918f, 0.9807852506637573f, 0.9891765117645264f, 0.9951847195625305f, 0.9987954497337341f, 1.0f, 0.9987954497337341f, 0.9951847195625305f, 0.9891765117645264f, 0.9807852506637573f, 0.9700312614440918f, 0.9569403529167175f, 0.9415440559387207f, 0.9238795042037964f, 0.903989315032959f, 0.8819212913513184f, 0.8577286005020142f, 0.8314695954322815f, 0.803207516670227f, 0.7730104327201843f, 0.7409511208534241f, 0.7071067690849304f, 0.6715589761734009f, 0.6343932747840881f, 0.5956993103027344f, 0.5555702447891235f, 0.5141027569770813f, 0.4713967442512512f, 0.4275550842285156f, 0.3826834261417389f, 0.3368898630142212f, 0.290284663438797f, 0.24298018217086792f, 0.19509032368659973f, 0.1467304676771164f, 0.0980171412229538f, 0.049067676067352295f, 0.0f, -0.049067676067352295f, -0.0980171412229538f, -0.1467304676771164f, -0.19509032368659973f, -0.24298018217086792f, -0.290284663438797f, -0.3368898630142212f, -0.3826834261417389f, -0.4275550842285156f, -0.4713967442512512f, -0.5141027569770813f, -0.5555702447891235f, -0.5956993103027344f, -0.6343932747840881f, -0.6715589761734009f, -0.7071067690849304f, -0.7409511208534241f, -0.7730104327201843f, -0.803207516670227f, -0.8314695954322815f, -0.8577286005020142f, -0.8819212913513184f, -0.903989315032959f, -0.9238795042037964f, -0.9415440559387207f, -0.9569403529167175f, -0.9700312614440918f, -0.9807852506637573f, -0.9891765117645264f, -0.9951847195625305f, -0.9987954497337341f, -1.0f, -0.9987954497337341f, -0.9951847195625305f, -0.9891765117645264f, -0.9807852506637573f, -0.9700312614440918f, -0.9569403529167175f, -0.9415440559387207f, -0.9238795042037964f, -0.903989315032959f, -0.8819212913513184f, -0.8577286005020142f, -0.8314695954322815f, -0.803207516670227f, -0.7730104327201843f, -0.7409511208534241f, -0.7071067690849304f, -0.6715589761734009f, -0.6343932747840881f, -0.5956993103027344f, -0.5555702447891235f, -0.5141027569770813f, -0.4713967442512512f, -0.4275550842285156f, -0.3826834261417389f, -0.3368898630142212f, -0.290284663438797f, -0.24298018217086792f, -0.19509032368659973f, -0.1467304676771164f, -0.0980171412229538f, -0.049067676067352295f};

// bit reversal of k over log2(H) bits, H = 32 or 64
template <int H>
__host__ __device__ constexpr int bitrev(int k) {
    int r = 0;
    for (int b = 1, t = H >> 1; b < H; b <<= 1, t >>= 1)
        if (k & b) r |= t;
    return r;
}
__host__ __device__ constexpr int bitrev5(int k) { return bitrev<32>(k); }

// in-place radix-2 DIF on H complex points, natural order in, bit-reversed order out; SIGN -1 forward, +1 inverse
template <int H, int SIGN>
__device__ __forceinline__ void cfft(float (&re)[H], float (&im)[H]) {
#pragma unroll
    for (int s = H / 2; s >= 1; s >>= 1) {
#pragma unroll
        for (int g = 0; g < H; g += 2 * s) {
#pragma unroll
            for (int j = 0; j < s; ++j) {
                const int i = g + j, t = j * (64 / s);   // twiddle angle 2 pi j / (2 s) = 2 pi t / 128
                const float ar = re[i], ai = im[i], br = re[i + s], bi = im[i + s];
                re[i] = ar + br;
                im[i] = ai + bi;
                const float dr = ar - br, di = ai - bi;
                if (t == 0) {
                    re[i + s] = dr;
                    im[i + s] = di;
                } else {
                    const float c = FFT_C128[t], sn = SIGN * FFT_S128[t];
                    re[i + s] = dr * c - di * sn;
                    im[i + s] = dr * sn + di * c;
                }
            }
        }
    }
}

// x[FN] real -> slots: sr[0] = X[0], si[0] = X[FN/2] (both real), (sr[k], si[k]) = X[k], k = 1..FN/2-1 (unnormalised DFT)
template <int FN>
__device__ __forceinline__ void rfft(const float (&x)[FN], float (&sr)[FN / 2], float (&si)[FN / 2]) {
    constexpr int H = FN / 2, TW = 128 / FN;      // table stride: angle 2 pi k / FN = 2 pi (k TW) / 128
    float re[H], im[H];
#pragma unroll
    for (int m = 0; m < H; ++m) {
        re[m] = x[2 * m];
        im[m] = x[2 * m + 1];
    }
    cfft<H, -1>(re, im);
    sr[0] = re[0] + im[0];
    si[0] = re[0] - im[0];
#pragma unroll
    for (int k = 1; k < H; ++k) {
        const float a = re[bitrev<H>(k)], b = im[bitrev<H>(k)], c = re[bitrev<H>(H - k)], d = im[bitrev<H>(H - k)];
        const float er = 0.5f * (a + c), ei = 0.5f * (b - d);
        const float orr = 0.5f * (b + d), oi = -0.5f * (a - c);
        const float wr = FFT_C128[k * TW], wi = -FFT_S128[k * TW];
        sr[k] = er + (wr * orr - wi * oi);
        si[k] = ei + (wr * oi + wi * orr);
    }
}

// inverse of rfft (exact inverse, includes the 1/FN): slots -> x[FN]
template <int FN>
__device__ __forceinline__ void irfft(const float (&sr)[FN / 2], const float (&si)[FN / 2], float (&x)[FN]) {
    constexpr int H = FN / 2, TW = 128 / FN;
    float re[H], im[H];
    re[0] = 0.5f * (sr[0] + si[0]);
    im[0] = 0.5f * (sr[0] - si[0]);
#pragma unroll
    for (int k = 1; k < H; ++k) {
        const float p = sr[k], q = si[k], r = sr[H - k], s = si[H - k];
        const float er = 0.5f * (p + r), ei = 0.5f * (q - s);
        const float dr = 0.5f * (p - r), di = 0.5f * (q + s);
        const float c = FFT_C128[k * TW], sn = FFT_S128[k * TW];
        const float orr = dr * c - di * sn, oi = dr * sn + di * c;
        re[k] = er - oi;
        im[k] = ei + orr;
    }
    cfft<H, 1>(re, im);
#pragma unroll
    for (int m = 0; m < H; ++m) {
        x[2 * m] = re[bitrev<H>(m)] * (1.0f / H);
        x[2 * m + 1] = im[bitrev<H>(m)] * (1.0f / H);
    }
}

__device__ __forceinline__ void rfft64(const float (&x)[64], float (&sr)[32], float (&si)[32]) { rfft<64>(x, sr, si); }
__device__ __forceinline__ void irfft64(const float (&sr)[32], const float (&si)[32], float (&x)[64]) { irfft<64>(sr, si, x); }

// ------------------------------------------------------------------------------------------------------------------
// 128-point real transforms on TWO LANES per window (lane pair 2w, 2w + 1; par = lane & 1): a 128-point transform in one
// lane needs ~300 registers (one wave per SIMD: the inverse transform kernel ran at half the 64-point kernel's rate);
// split over a lane pair each lane runs the SAME 32-point complex FFT as the 64-point transform, plus two exchange steps
// through DPP (quad_perm 1,0,3,2: no LDS).  Both exchanges are symmetric -- each lane wants the neighbour's register with
// the same compile-time index -- which is what decides the split:
//   forward   lane par holds samples 4m + 2par + {0, 1} (z[2m + par] of the packed half-size sequence): decimation in time,
//             Z[k'] = A + w^-k' B on lane 0, Z[32 + k'] = A - w^-k' B on lane 1, and the real split pairs Z[k] with
//             Z[64 - k], i.e. lane 0's register k' with lane 1's register 32 - k' and vice versa;
//   inverse   the mirror image: lane par holds X[32 par + j], pairs with the neighbour's register 32 - j, then decimation
//             in frequency: lane 0 transforms Z[k] + Z[k + 32] (even samples z[2m]), lane 1 (Z[k] - Z[k + 32]) w^k.
// Lane 1's twiddles are lane 0's rotated by a quarter turn (k = 32 + j), so both lanes use the same constants.
__device__ __forceinline__ float lane_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}

// xs[2m + e] = x[4m + 2par + e]  ->  (sr[j], si[j]) = X[32 par + j], j = 0..31; lane 0's j = 0 carries (X[0], X[64]), both real
__device__ __forceinline__ void rfft128_pair(const float (&xs)[64], int par, float (&sr)[32], float (&si)[32]) {
    float re[32], im[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        re[m] = xs[2 * m];
        im[m] = xs[2 * m + 1];
    }
    cfft<32, -1>(re, im);                                   // A (lane 0) / B (lane 1), bit-reversed
    if (par) {                                              // T[k'] = w^-k' B[k'], w^-k' = exp(-2 pi i k' / 64)
#pragma unroll
        for (int k = 1; k < 32; ++k) {
            const int i = bitrev<32>(k);
            const float a = re[i], b = im[i], c = FFT_C128[2 * k], s = -FFT_S128[2 * k];
            re[i] = a * c - b * s;
            im[i] = a * s + b * c;
        }
    }
    const float sgn = par ? -1.f : 1.f;                     // lane 0: mine + other, lane 1: other - mine
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float orr = lane_xor1(re[i]), oi = lane_xor1(im[i]);
        re[i] = sgn * re[i] + orr;
        im[i] = sgn * im[i] + oi;
    }
    {
        const float a = re[0], b = im[0];
        sr[0] = par ? a : a + b;                            // lane 1: X[32] = conj(Z[32]) (self-paired)
        si[0] = par ? -b : a - b;
    }
#pragma unroll
    for (int k = 1; k < 32; ++k) {
        const float a = re[bitrev<32>(k)], b = im[bitrev<32>(k)];
        const float c = lane_xor1(re[bitrev<32>(32 - k)]), d = lane_xor1(im[bitrev<32>(32 - k)]);
        const float er = 0.5f * (a + c), ei = 0.5f * (b - d);
        const float orr = 0.5f * (b + d), oi = -0.5f * (a - c);
        const float o1 = par ? oi : orr, o2 = par ? -orr : oi;     // lane 1: twiddle of k = 32 + j = -i x twiddle of j
        const float wr = FFT_C128[k], wi = -FFT_S128[k];
        sr[k] = er + (wr * o1 - wi * o2);
        si[k] = ei + (wr * o2 + wi * o1);
    }
}

// inverse of rfft128_pair (includes the 1/128): (sr[j], si[j]) = X[32 par + j] -> xs[2m + e] = x[4m + 2par + e]
__device__ __forceinline__ void irfft128_pair(const float (&sr)[32], const float (&si)[32], int par, float (&xs)[64]) {
    float re[32], im[32];
    re[0] = par ? sr[0] : 0.5f * (sr[0] + si[0]);           // lane 0: (DC, Nyquist); lane 1: Z[32] = conj(X[32])
    im[0] = par ? -si[0] : 0.5f * (sr[0] - si[0]);
#pragma unroll
    for (int k = 1; k < 32; ++k) {
        const float p = sr[k], q = si[k], r = lane_xor1(sr[32 - k]), s = lane_xor1(si[32 - k]);
        const float er = 0.5f * (p + r), ei = 0.5f * (q - s);
        const float dr = 0.5f * (p - r), di = 0.5f * (q + s);
        const float c = FFT_C128[k], sn = FFT_S128[k];
        const float orr = dr * c - di * sn, oi = dr * sn + di * c;
        const float o1 = par ? -oi : orr, o2 = par ? orr : oi;    // lane 1: twiddle of k = 32 + j = +i x twiddle of j
        re[k] = er - o2;
        im[k] = ei + o1;
    }
    const float sgn = par ? -1.f : 1.f;                     // lane 0: Z[k] + Z[k + 32]; lane 1: Z[k] - Z[k + 32] = other - mine
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const float orr = lane_xor1(re[k]), oi = lane_xor1(im[k]);
        re[k] = sgn * re[k] + orr;
        im[k] = sgn * im[k] + oi;
    }
    if (par) {                                              // (Z[k] - Z[k + 32]) w^k, w^k = exp(+2 pi i k / 64)
#pragma unroll
        for (int k = 1; k < 32; ++k) {
            const float a = re[k], b = im[k], c = FFT_C128[2 * k], s = FFT_S128[2 * k];
            re[k] = a * c - b * s;
            im[k] = a * s + b * c;
        }
    }
    cfft<32, 1>(re, im);
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        xs[2 * m] = re[bitrev<32>(m)] * (1.0f / 64.0f);
        xs[2 * m + 1] = im[bitrev<32>(m)] * (1.0f / 64.0f);
    }
}

}  // namespace zns
