// Frequency-domain (overlap-save) form of the cv2..cv8 convolutions: same fp32 results, ~6x fewer MFMA FLOPs (cv7 / cv8, the
// one-row layers: ~10x; they run with the batch as the rows of one spectral sample per branch, see geom()).
//
// The kernels are wide along time (kw = 13..21 taps) and time is long (T = 313..626), so the time axis is cut
// into 64-sample windows that overlap by kw-1 (V = 65-kw valid outputs per window).  A 64-point real DFT of each
// window turns the kw-tap correlation along time into ONE complex multiply per frequency; the correlation over
// input channels and kernel ROWS stays a dense contraction, done on the fp32 matrix cores by the SAME conv_core /
// wgrad_core templates as the direct path, instantiated as a 1-D correlation with kh taps spaced NB positions
// apart along the flattened (row, window) axis, with 2x real-embedded channels (re, im) and one weight set per
// frequency.  Per output window the direct form costs V*kw MACs per (ci, co, r), this form 4*32 (32 complex
// bins, 4 real MACs each): 5.1x (cv2) .. 7.0x (cv6) fewer.  Everything stays fp32 (no reduced-precision
// shortcut): error vs the direct kernels ~1e-6 of the tensor max.
//
//   forward        y  = IDFT( G  . DFT(x windows) ),   G [f] = sum_s W[s]        e^{+i th f s}
//   backward-data  dx = IDFT( G' . DFT(dz windows) ),  G'[f] = sum_s W[kw-1-s]   e^{+i th f s}, roles of ci/co swapped
//   backward-wts   dW[s] = IDFT_s( sum_windows conj(DFT(dz V-block, zero padded)) . DFT(x window) )
//
// Complex products use the 3-multiplication (Karatsuba) form, so each frequency is THREE independent real GEMMs
//   Q1 = Gr Xr,  Q2 = Gi Xi,  Q3 = (Gr+Gi)(Xr+Xi);   Yr = Q1 - Q2,  Yi = Q3 - Q1 - Q2
// (backward-weights: R1 = dzr (x) xr, R2 = dzi (x) xi, R3 = (dzr-dzi) (x) (xr+xi); re = R1+R2, im = R3-R1+R2),
// i.e. 3*32 real MACs per window position instead of V*kw: 6.8x (cv2) .. 9.4x (cv6) fewer than the direct form.
// Spectra live in the workspace as S[sample][slot = 3 f + t][C][P], P = H*NB, planes t = (re, im, re +/- im);
// frequency slot 0 carries the two real bins (DC in t = 0, Nyquist in t = 1, t = 2 unused).
#include <cstdlib>
#include <type_traits>

#include "zns_conv_core.h"
#include "zns_fft64.h"

namespace zns {

// Window length FN: 64 samples for cv2 / cv5 / cv6 and for short rows, 128 for cv3 / cv4 where that needs fewer
// frequency-windows (use_win128).  FF = FN / 2 frequencies, FS = 3 FF weight / spectrum slots per sample: (frequency f,
// Karatsuba part t) -> 3 f + t.  Every transform kernel is instantiated per window length; the GEMM kernels only see more
// slots (run-time F) and fewer positions.
constexpr int FDIL = 12;     // 64-point windows: at most 12 windows per row segment (the row taps are NB positions apart)
constexpr int FDIL128 = 6;   // 128-point windows: at most 6
constexpr int FDILW = 12;    // backward-weights tiles keep two input-channel tiles in LDS: window count per row <= 12
// Transform blocks are 256 threads: one lane per 64-point window, a lane PAIR per 128-point window (zns_fft64.h)
template <int FN> constexpr int win_per_block() { return FN == 64 ? 256 : 128; }
// the inverse transform keeps a block's outputs in LDS (V floats per window): 128-thread blocks for 128-point windows
// (64 windows, 29 KB: five blocks per CU; with 256 threads it was two)
template <int FN> constexpr int inv_threads() { return FN == 64 ? 256 : 128; }
template <int FN> constexpr int inv_win_per_block() { return FN == 64 ? 256 : 64; }

// Spectrum planes are addressed as  (per-lane byte offset, wave-uniform plane offset)  through a buffer
// descriptor: the 96 plane accesses of a window then share ONE address VGPR (the plane offset goes in an SGPR).
__device__ __forceinline__ float spec_load(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ void spec_store(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, 0);
}

// ---------------------------------------------------------------------------------------------------------
// Windows are addressed by ONE flattened index per sample, q = (c * H + h) * NB + b = c * P + p: the spectrum planes
// S[n][slot][C][P] and the activation rows x[n][C][H][T] are both contiguous in it, so a block takes 256 consecutive
// q regardless of how small a single channel's P is (cv5/cv6: P = 96) and every lane of every wave has a window.
//
// x [nb, C, H, T] -> S [nb][FS][C][P].  Window b of a row starts at column b*V - off and has `valid` live samples
// (64 for the overlapped correlation windows, V for the zero-padded blocks of backward-weights' dZ).
//
// Long rows are cut into NS segments of NB <= 12 windows each; a segment is a separate spectral "sample"
// n' = n * NS + seg (the row taps of the GEMM stage are NB positions apart, so NB must stay small), NBT = windows
// of the whole row.
template <int FN>
__global__ void __launch_bounds__(256) spec_fwd_kernel(const float* __restrict__ x, float* __restrict__ S, int C,
                                                       int H, int T, int V, int NB, int off, int valid, int chunks,
                                                       float third_sign, int NS, int NBT, float* __restrict__ S2, int remap) {
    constexpr int FF = FN / 2, FS = 3 * FF, WINB = win_per_block<FN>(), K64 = FN / 64, WPW = WINB / 4;   // WPW: windows a wave fetches
    // Windows are staged through LDS so that global reads are coalesced: one wave-instruction fetches ONE window
    // (64 consecutive samples); all 64 fetches of a wave are issued before the first LDS write.  Each lane then
    // reads back its own window (row stride 65 floats: conflict free) and transforms it in registers.
    __shared__ float win[WINB * (FN + 1)];
    const int CP = C * H * NB;
    const int chunk = blockIdx.x % chunks;
    const int ns = blockIdx.x / chunks;             // spectral sample = (sample, segment)
    const int n = ns / NS, bseg = (ns - n * NS) * NB;
    const int lane = threadIdx.x & 63;
    const int wbase = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * WPW);
    const float* xn = x + (size_t)n * C * H * T;
    float xw[64];
    {
        // wave-uniform walk: (row R = c*H + h, window bb) of the 64 windows this wave fetches
        const int qw = chunk * WINB + wbase;
        int R = qw / NB, bb = qw - R * NB;
        int left = CP - qw;                           // windows that exist from here on
        // remap (the one-row layers, see geom()): the H "rows" of a channel are the samples of the batch, x is [H][C][T]:
        // row R = c * H + h lives at tensor row h * C + c (tracked by increments: no division per window)
        int rc = remap ? R / H : 0, rh = remap ? R - rc * H : 0;
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
#pragma unroll
            for (int h = 0; h < K64; ++h) {           // a 128-sample window is two wave-instructions
                const int smp = lane + 64 * h;
                const int col = (bseg + bb) * V - off + smp;
                const size_t ro = remap ? ((size_t)rh * C + rc) * T : (size_t)R * T;
                xw[i * K64 + h] = (i < left && bseg + bb < NBT && smp < valid && col >= 0 && col < T) ? xn[ro + col] : 0.f;
            }
            if (++bb == NB) {
                bb = 0; ++R;
                if (++rh == H) { rh = 0; ++rc; }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i)
#pragma unroll
        for (int h = 0; h < K64; ++h) win[(wbase + i) * (FN + 1) + lane + 64 * h] = xw[i * K64 + h];
    __syncthreads();
    const size_t tstride = (size_t)CP;          // between consecutive slots
    if constexpr (FN == 128) {
        // a lane pair per window (rfft128_pair): lane par holds samples 4m + 2par + {0, 1} and stores frequencies 32par + j
        const int w = threadIdx.x >> 1, par = threadIdx.x & 1;
        const int q = chunk * WINB + w;
        if (q >= CP) return;                    // (both lanes of a pair leave together)
        const float* wrow = win + w * (FN + 1) + 2 * par;
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            xw[2 * m] = wrow[4 * m];
            xw[2 * m + 1] = wrow[4 * m + 1];
        }
        float sr[32], si[32];
        rfft128_pair(xw, par, sr, si);
        float* out = S + (size_t)ns * FS * CP + q + (size_t)par * 96 * tstride;
        const bool dc = par == 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            out[(3 * j) * tstride] = sr[j];
            out[(3 * j + 1) * tstride] = si[j];
            out[(3 * j + 2) * tstride] = (j == 0 && dc) ? 0.f : sr[j] + third_sign * si[j];
        }
        if (S2 != nullptr) {
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                const int i0 = 4 * m + 2 * par;
                xw[2 * m] = (i0 < V) ? wrow[4 * m + off] : 0.f;
                xw[2 * m + 1] = (i0 + 1 < V) ? wrow[4 * m + 1 + off] : 0.f;
            }
            rfft128_pair(xw, par, sr, si);
            float* out2 = S2 + (size_t)ns * FS * CP + q + (size_t)par * 96 * tstride;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                out2[(3 * j) * tstride] = sr[j];
                out2[(3 * j + 1) * tstride] = si[j];
                out2[(3 * j + 2) * tstride] = (j == 0 && dc) ? 0.f : sr[j] - si[j];
            }
        }
        return;
    }
    const int q = chunk * WINB + threadIdx.x;
    if (q >= CP) return;
#pragma unroll
    for (int m = 0; m < 64; ++m) xw[m] = win[threadIdx.x * (FN + 1) + m];
    float* out = S + (size_t)ns * FS * CP + q;
    float sr[32], si[32];
    rfft64(xw, sr, si);
#pragma unroll
    for (int f = 0; f < 32; ++f) {
        out[(3 * f) * tstride] = sr[f];
        out[(3 * f + 1) * tstride] = si[f];
        out[(3 * f + 2) * tstride] = f ? sr[f] + third_sign * si[f] : 0.f;
    }
    if (S2 != nullptr) {
        // second spectrum from the same staged samples: the zero-padded V-block of backward-weights is samples
        // [off, off + V) of the overlapped window (both start V apart; off = kw/2), planes (re, im, re - im)
#pragma unroll
        for (int m = 0; m < 64; ++m) xw[m] = (m < V) ? win[threadIdx.x * (FN + 1) + m + off] : 0.f;
        rfft64(xw, sr, si);
        float* out2 = S2 + (size_t)ns * FS * CP + q;
#pragma unroll
        for (int f = 0; f < 32; ++f) {
            out2[(3 * f) * tstride] = sr[f];
            out2[(3 * f + 1) * tstride] = si[f];
            out2[(3 * f + 2) * tstride] = f ? sr[f] - si[f] : 0.f;
        }
    }
}

// S [nb][FS][C][P] -> inverse 64-point real FFT of each window, first V samples kept, with the layer epilogue fused:
//   MODE 0  z  = raw conv output                                         (plain store)
//   MODE 1  y  = dropout(relu(maxpool_rows(z) + bias)), arg-max rows     (forward)
//   MODE 2  dx = max-unpool(mask(z))                                      (backward-data epilogue, see zns.h)
// A block owns WPB consecutive windows of the flattened index, WPB a multiple of pool*NB so that the `pool` rows of
// every pooling group are inside the block.  Phase 1: every thread inverts ONE window into the LDS tile.  Phase 2:
// the block's outputs form (almost) one contiguous run of the result; they are written coalesced, and the row
// pooling (max + arg-max over the group's `pool` windows) happens on the way out of LDS.
struct InvEpi {
    Ptr2 bias;                 // MODE 1
    uint8_t* argmax;
    int pool, B;
    uint32_t thresh, seed;
    int layer;
    float inv_keep;            // MODE 1 and 2
    const float* xmask;        // MODE 2
    const uint8_t* prev_argmax;
    int prev_pool;
    uint32_t mV, mNB, mHq;     // ceil(2^32 / d) for the epilogue's index divisions (exact for x * d < 2^32)
    uint32_t mT, mPP;          // FLAT epilogue: divisions by T and by prev_pool
    int remap;                 // the one-row layers (geom()): rows are the samples of the batch, tensors are [H][C][T] (not ROW / FLAT)
};

template <int FN> constexpr int iv_max() { return FN - 12; }     // V <= FN - 12 (kw >= 13)

__device__ __forceinline__ int fastdiv(int x, int d, uint32_t magic) { return d == 1 ? x : (int)__umulhi((uint32_t)x, magic); }
static uint32_t div_magic(int d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d); }

// VW > 1: V and T are multiples of VW (4: cv2, cv4, cv6 at T = 512; 2: cv3, cv5): the epilogue moves VW samples per
// thread with 8- or 16-byte LDS reads, mask loads and stores; the LDS row stride is then V itself.
template <int N> struct VecOf;
template <> struct VecOf<2> { typedef f32x2 type; typedef uint16_t amtype; };
template <> struct VecOf<4> { typedef f32x4 type; typedef uint32_t amtype; };
template <> struct VecOf<1> { typedef float type; typedef uint8_t amtype; };

// ROW (VW = 4 only): V is even but not a multiple of 4 (cv3: 50, cv5: 46) and the row is ONE segment (NS = 1, T % 4 == 0):
// the epilogue then walks each output row in 16-byte chunks that ignore the window boundaries -- a row's windows are
// contiguous in the LDS tile -- reading the tile with two 8-byte loads per chunk.  With 8-byte stores (VW = 2) these two
// layers' inverse transforms ran at 4.2 TB/s against 5.3-5.6 for the others.
template <int VW, bool ROW>
__device__ __forceinline__ typename VecOf<VW>::type ld_stage(const float* sp) {
    if constexpr (ROW) {
        const f32x2 lo = *reinterpret_cast<const f32x2*>(sp), hi = *reinterpret_cast<const f32x2*>(sp + 2);
        f32x4 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
        return v;
    } else {
        return *reinterpret_cast<const typename VecOf<VW>::type*>(sp);
    }
}

template <int MODE, int VW, bool ROW = false, bool FLAT = false, int FN = 64>
__global__ void __launch_bounds__(inv_threads<FN>()) spec_inv_kernel(const float* __restrict__ S, float* __restrict__ out, int C, int H,
                                                       int T, int V, int NB, int chunks, int WPB, int NS, InvEpi E) {
    constexpr int FF = FN / 2, FS = 3 * FF, NTHR = inv_threads<FN>(), IVMAX = iv_max<FN>();
    extern __shared__ __attribute__((aligned(16))) float stage[];   // [inv_win_per_block<FN>()][VS], VS odd: per-lane rows hit distinct banks
    const int pool = MODE == 1 ? E.pool : 1;
    const int Hq = H / pool;                       // output rows per channel (pooled rows in MODE 1)
    const int CP = C * H * NB;
    const int VS = VW > 1 ? V : (V | 1);
    const int chunk = blockIdx.x % chunks;
    const int ns = blockIdx.x / chunks;             // spectral sample = (sample, segment)
    const int n = ns / NS, bseg = (ns - n * NS) * NB;
    const int q0 = chunk * WPB;
    const int nwin = min(WPB, CP - q0);            // a multiple of pool * NB
    const unsigned tstride = (unsigned)CP * 4u;    // bytes between consecutive slots
    float* sbase = const_cast<float*>(S) + (size_t)ns * FS * CP;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(sbase, 0, (int)(tstride * FS), 0x00020000);
    if constexpr (FN == 128) {
        // a lane pair per window (irfft128_pair): lane par loads frequencies 32par + j and ends up with samples 4m + 2par + {0, 1}
        const int w = threadIdx.x >> 1, par = threadIdx.x & 1;
        if (w < nwin) {
            const unsigned voff = (unsigned)(q0 + w) * 4u + (unsigned)par * 96u * tstride;
            float yr[32], yi[32];
            {
                const float q1 = spec_load(rs, voff, 0), q2 = spec_load(rs, voff, tstride), q3 = spec_load(rs, voff, 2 * tstride);
                yr[0] = par ? q1 - q2 : q1;              // lane 0: DC and Nyquist (both real); lane 1: frequency 32
                yi[0] = par ? q3 - q1 - q2 : q2;
            }
#pragma unroll
            for (int j = 1; j < 32; ++j) {
                const float q1 = spec_load(rs, voff, (3 * j) * tstride), q2 = spec_load(rs, voff, (3 * j + 1) * tstride),
                            q3 = spec_load(rs, voff, (3 * j + 2) * tstride);
                yr[j] = q1 - q2;
                yi[j] = q3 - q1 - q2;
            }
            float xs[64];
            irfft128_pair(yr, yi, par, xs);
            float* mine = stage + w * VS + 2 * par;
#pragma unroll
            for (int m = 0; m < (IVMAX + 3) / 4; ++m) {
                if (4 * m + 2 * par < V) mine[4 * m] = xs[2 * m];
                if (4 * m + 2 * par + 1 < V) mine[4 * m + 1] = xs[2 * m + 1];
            }
        }
    } else if ((int)threadIdx.x < nwin) {
        const unsigned voff = (unsigned)(q0 + threadIdx.x) * 4u;
        float yr[FF], yi[FF];
        yr[0] = spec_load(rs, voff, 0);                  // DC
        yi[0] = spec_load(rs, voff, tstride);            // Nyquist
#pragma unroll
        for (int f = 1; f < FF; ++f) {
            const float q1 = spec_load(rs, voff, (3 * f) * tstride), q2 = spec_load(rs, voff, (3 * f + 1) * tstride),
                        q3 = spec_load(rs, voff, (3 * f + 2) * tstride);
            yr[f] = q1 - q2;
            yi[f] = q3 - q1 - q2;
        }
        float xo[FN];
        irfft<FN>(yr, yi, xo);
        float* mine = stage + threadIdx.x * VS;
#pragma unroll
        for (int v = 0; v < IVMAX; ++v)
            if (v < V) mine[v] = xo[v];
    }
    __syncthreads();
    const int group = pool * NB;
    const int Rq0 = q0 / group;                    // first (pooled) output row of this block, counted over (c, hq)
    const int nout = (nwin / pool) * V;
    const int br = n / (MODE == 1 ? E.B : 1), bb = n - br * (MODE == 1 ? E.B : 1);
    const uint32_t lseed = MODE == 1 ? layer_seed(E.seed, E.layer, br) : 0u;
    if constexpr (FLAT) {
        // Rows whose length is not a multiple of 4 (T = 313, the reference's tile): the block's outputs are still ONE
        // contiguous run of memory -- (pooled) rows x [prev_pool rows] x T -- so it is cut into 16-byte chunks aligned in
        // GLOBAL memory; a chunk's four elements may sit in two rows, each is located (one division per chunk, then
        // increments) and read from the LDS tile on its own.  One-segment rows only (NS = 1).  With 4-byte stores these
        // launches cost 1.45x their T = 512 time per byte.
        const int pp = MODE == 2 ? E.prev_pool : 1;
        const int nrows = nwin / group;
        const int count = nrows * pp * T;
        const size_t g0 = ((size_t)n * C * Hq + Rq0) * pp * T;
        const int shift = (int)(g0 & 3);
        const int nchunk = (count + shift + 3) >> 2;
        constexpr int EV = 2;                          // two chunks per thread and trip: all global loads before the stores
        for (int e0 = threadIdx.x; e0 < nchunk; e0 += NTHR * EV) {
            float r[EV][4], mk[EV][4];
            int am[EV][4], qq[EV][4], Rqv[EV][4], colv[EV][4];
            bool okv[EV][4];
#pragma unroll
            for (int u = 0; u < EV; ++u) {
                const int f0 = 4 * (e0 + u * NTHR) - shift;
                int rowo = fastdiv(f0 < 0 ? 0 : f0, T, E.mT);
                int col = (f0 < 0 ? 0 : f0) - rowo * T + (f0 < 0 ? f0 : 0);     // negative only in the block's first chunk
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = f0 + k;
                    okv[u][k] = e0 + u * NTHR < nchunk && f >= 0 && f < count;
                    r[u][k] = 0.f; mk[u][k] = 1.f; am[u][k] = 0; qq[u][k] = 0; Rqv[u][k] = 0; colv[u][k] = col;
                    if (okv[u][k]) {
                        const int rin = pp > 1 ? fastdiv(rowo, pp, E.mPP) : rowo;
                        qq[u][k] = rowo - rin * pp;
                        Rqv[u][k] = Rq0 + rin;
                        const float* sp = stage + rin * group * VS + col;
                        float rv = sp[0];
                        int av = 0;
                        for (int a = 1; a < pool; ++a) {
                            const float t = sp[a * NB * VS];
                            if (pool_takes(t, rv)) { rv = t; av = a; }
                        }
                        r[u][k] = rv;
                        am[u][k] = av;
                        if (MODE == 1) mk[u][k] = E.bias.p[br][fastdiv(Rqv[u][k], Hq, E.mHq)];
                        if (MODE == 2 && pp > 1) {
                            const size_t oin = ((size_t)n * C * Hq + Rqv[u][k]) * T + col;
                            if (E.xmask) mk[u][k] = E.xmask[oin];
                            am[u][k] = E.prev_argmax[oin];
                        }
                    }
                    if (++col >= T) { col = 0; ++rowo; }
                }
                if (MODE == 2 && pp == 1 && E.xmask) {      // the mask has the output's layout: one aligned 16-byte load
                    if (okv[u][0] && okv[u][3]) {
                        const f32x4 m4 = *reinterpret_cast<const f32x4*>(E.xmask + g0 + f0);
#pragma unroll
                        for (int k = 0; k < 4; ++k) mk[u][k] = m4[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (okv[u][k]) mk[u][k] = E.xmask[g0 + f0 + k];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < EV; ++u) {
                const int f0 = 4 * (e0 + u * NTHR) - shift;
                f32x4 val;
                uint32_t amw = 0u;
                bool any = false, all = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float t = r[u][k];
                    if (MODE == 1) {
                        t += mk[u][k];
                        t = relu_nan(t);
                        if (E.thresh) {
                            const uint64_t idx = ((uint64_t)bb * C * Hq + Rqv[u][k]) * T + colv[u][k];
                            t = drop_apply(drop_keep(lseed, idx, E.thresh), t, E.inv_keep);
                        }
                        amw |= (uint32_t)am[u][k] << (8 * k);
                    } else if (MODE == 2) {
                        if (E.xmask) t = (mk[u][k] != 0.f) ? t * E.inv_keep : 0.f;
                        if (pp > 1) t = (am[u][k] == qq[u][k]) ? t : 0.f;
                    }
                    val[k] = t;
                    any = any || okv[u][k];
                    all = all && okv[u][k];
                }
                if (!any) continue;
                float* d = out + g0 + f0;
                if (all) {
                    *reinterpret_cast<f32x4*>(d) = val;
                    if (MODE == 1 && E.argmax && pool > 1) *reinterpret_cast<uint32_t*>(E.argmax + g0 + f0) = amw;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (okv[u][k]) {
                            d[k] = val[k];
                            if (MODE == 1 && E.argmax && pool > 1) E.argmax[g0 + f0 + k] = (uint8_t)(amw >> (8 * k));
                        }
                }
            }
        }
        return;
    }
    if constexpr (VW > 1) {
        typedef typename VecOf<VW>::type vec_t;
        typedef typename VecOf<VW>::amtype am_t;
        const int V4n = ROW ? T / 4 : V / VW;                       // chunks per row / per window
        const int nout4 = (ROW ? nwin / group : nwin / pool) * V4n;
        constexpr int EV = 2;                      // 2 x 4 outputs per thread and trip, loads before stores
        for (int e0 = threadIdx.x; e0 < nout4; e0 += NTHR * EV) {
            vec_t r[EV], mk[EV];
            uint32_t amw[EV];
            size_t o[EV];
            int Rq[EV], col[EV], orow[EV];
            bool ok[EV];
#pragma unroll
            for (int u = 0; u < EV; ++u) {
                const int e = e0 + u * NTHR;
                const int w = fastdiv(e, V4n, E.mV), v = (e - w * V4n) * VW;   // ROW: w = row in the block, v = column
                const int lg = ROW ? w : fastdiv(w, NB, E.mNB), b = ROW ? 0 : w - lg * NB;
                col[u] = ROW ? v : (bseg + b) * V + v;
                ok[u] = e < nout4 && col[u] < T;
                Rq[u] = Rq0 + lg;
                if (E.remap) {                                         // tensor row of (c, hq): hq * C + c
                    const int c = fastdiv(Rq[u], Hq, E.mHq);
                    orow[u] = (Rq[u] - c * Hq) * C + c;
                } else {
                    orow[u] = Rq[u];
                }
                o[u] = ((size_t)n * C * Hq + orow[u]) * T + col[u];
#pragma unroll
                for (int k = 0; k < VW; ++k) { r[u][k] = 0.f; mk[u][k] = 1.f; }
                amw[u] = 0u;
                if (ok[u]) {
                    const float* sp = stage + (lg * group + b) * VS + v;
                    r[u] = ld_stage<VW, ROW>(sp);
                    for (int a = 1; a < pool; ++a) {
                        const vec_t t = ld_stage<VW, ROW>(sp + a * NB * VS);
#pragma unroll
                        for (int k = 0; k < VW; ++k)
                            if (pool_takes(t[k], r[u][k])) { r[u][k] = t[k]; amw[u] = (amw[u] & ~(0xffu << (8 * k))) | ((uint32_t)a << (8 * k)); }
                    }
                    if (MODE == 1) {
                        const float bvv = E.bias.p[br][fastdiv(Rq[u], Hq, E.mHq)];
#pragma unroll
                        for (int k = 0; k < VW; ++k) mk[u][k] = bvv;
                    }
                    if (MODE == 2) {
                        if (E.xmask) mk[u] = *reinterpret_cast<const vec_t*>(E.xmask + o[u]);
                        if (E.prev_pool > 1) amw[u] = *reinterpret_cast<const am_t*>(E.prev_argmax + o[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < EV; ++u) {
                if (!ok[u]) continue;
                vec_t rr = r[u];
                if (MODE == 0) {
                    *reinterpret_cast<vec_t*>(out + o[u]) = rr;
                } else if (MODE == 1) {
#pragma unroll
                    for (int k = 0; k < VW; ++k) {
                        float t = rr[k] + mk[u][k];
                        t = relu_nan(t);
                        if (E.thresh) {
                            const uint64_t idx = ((uint64_t)bb * C * Hq + orow[u]) * T + col[u] + k;
                            t = drop_apply(drop_keep(lseed, idx, E.thresh), t, E.inv_keep);
                        }
                        rr[k] = t;
                    }
                    *reinterpret_cast<vec_t*>(out + o[u]) = rr;
                    if (E.argmax && pool > 1) *reinterpret_cast<am_t*>(E.argmax + o[u]) = (am_t)amw[u];
                } else {
                    if (E.xmask) {
#pragma unroll
                        for (int k = 0; k < VW; ++k) rr[k] = (mk[u][k] != 0.f) ? rr[k] * E.inv_keep : 0.f;
                    }
                    const int pp = E.prev_pool;
                    if (pp > 1) {
                        float* d = out + (((size_t)n * C * Hq + orow[u]) * pp) * T + col[u];
                        for (int q = 0; q < pp; ++q) {
                            vec_t t;
#pragma unroll
                            for (int k = 0; k < VW; ++k) t[k] = ((int)((amw[u] >> (8 * k)) & 0xffu) == q) ? rr[k] : 0.f;
                            *reinterpret_cast<vec_t*>(d + (size_t)q * T) = t;
                        }
                    } else {
                        *reinterpret_cast<vec_t*>(out + o[u]) = rr;
                    }
                }
            }
        }
        return;
    }
    // four outputs per thread and trip: their global loads (mask, arg-max, bias) are all in flight before the first
    // store, instead of one exposed memory round trip per element
    constexpr int EU = 4;
    for (int e0 = threadIdx.x; e0 < nout; e0 += NTHR * EU) {
        float r[EU], mk[EU];
        size_t o[EU];
        int am[EU], Rq[EU], col[EU], orow[EU];
        bool ok[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int e = e0 + u * NTHR;
            const int w = fastdiv(e, V, E.mV), v = e - w * V;          // pooled window inside the block, sample
            const int lg = fastdiv(w, NB, E.mNB), b = w - lg * NB;     // pooling group inside the block, window in its row
            col[u] = (bseg + b) * V + v;
            ok[u] = e < nout && col[u] < T;
            Rq[u] = Rq0 + lg;                                          // = c * Hq + hq
            if (E.remap) {                                             // tensor row of (c, hq): hq * C + c
                const int c = fastdiv(Rq[u], Hq, E.mHq);
                orow[u] = (Rq[u] - c * Hq) * C + c;
            } else {
                orow[u] = Rq[u];
            }
            o[u] = ((size_t)n * C * Hq + orow[u]) * T + col[u];
            r[u] = 0.f;
            am[u] = 0;
            mk[u] = 1.f;
            if (ok[u]) {
                const float* sp = stage + (lg * group + b) * VS + v;
                r[u] = sp[0];
                for (int a = 1; a < pool; ++a) {
                    const float t = sp[a * NB * VS];
                    if (pool_takes(t, r[u])) { r[u] = t; am[u] = a; }   // first maximum wins, a NaN wins over all: ATen's max_pool2d
                }
                if (MODE == 1) mk[u] = E.bias.p[br][fastdiv(Rq[u], Hq, E.mHq)];
                if (MODE == 2) {
                    if (E.xmask) mk[u] = E.xmask[o[u]];
                    if (E.prev_pool > 1) am[u] = E.prev_argmax[o[u]];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            if (!ok[u]) continue;
            float rr = r[u];
            if (MODE == 0) {
                out[o[u]] = rr;
            } else if (MODE == 1) {
                rr += mk[u];
                rr = relu_nan(rr);
                if (E.thresh) {
                    const uint64_t idx = ((uint64_t)bb * C * Hq + orow[u]) * T + col[u];
                    rr = drop_apply(drop_keep(lseed, idx, E.thresh), rr, E.inv_keep);
                }
                out[o[u]] = rr;
                if (E.argmax && pool > 1) E.argmax[o[u]] = (uint8_t)am[u];
            } else {
                if (E.xmask) rr = (mk[u] != 0.f) ? rr * E.inv_keep : 0.f;
                const int pp = E.prev_pool;
                if (pp > 1) {
                    float* d = out + (((size_t)n * C * Hq + orow[u]) * pp) * T + col[u];
                    for (int q = 0; q < pp; ++q) d[(size_t)q * T] = (q == am[u]) ? rr : 0.f;
                } else {
                    out[o[u]] = rr;
                }
            }
        }
    }
}

// Frequency weights in conv_core's slab order: three REAL [cout_c x cin_c*kh] operators per frequency
//   slot 3f+0: Gr   slot 3f+1: Gi   slot 3f+2: Gr + Gi      (f = 0: G(0), G(32), unused)
// laid out [branch*FS + slot][cob][cc][cp][tap][wm][lane_h][i][m].  One thread owns one (co, ci, tap): it reads the
// kw time taps once and loops over the 32 frequencies, so every slab is written with coalesced stores.
template <int FN>
__global__ void __launch_bounds__(256) pack_freq_weights_kernel(Ptr2 w, float* __restrict__ dst, int cin_c, int cout_c,
                                                                int kh, int kw, int cob_sz, int ck, int mt, int wm_n,
                                                                int transposed, long long per_wf, int nbr) {
    constexpr int FF = FN / 2, FS = 3 * FF;
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= per_wf * nbr) return;
    const int br = (int)(o / per_wf);
    long long q = o % per_wf;                       // offset inside one slab = the slab layout itself
    const long long off = q;
    const int m = (int)(q % mt); q /= mt;
    const int i = (int)(q % 32); q /= 32;
    const int h = (int)(q % 2); q /= 2;
    const int wm = (int)(q % wm_n); q /= wm_n;
    const int tap = (int)(q % kh); q /= kh;
    const int cp = (int)(q % (ck / 2)); q /= (ck / 2);
    const int ncc = cin_c / ck;
    const int cc = (int)(q % ncc); q /= ncc;
    const int cob = (int)q;
    const int co = cob * cob_sz + (wm * mt + m) * 32 + i;
    const int ci = cc * ck + 2 * cp + h;
    // G[f] = sum_s W[s] e^{+2 pi i f s / FN} = conj(DFT_FN(W zero padded)[f]): one register FFT per thread (the first
    // version summed 32 x kw twiddle products through LDS: 1600 FMAs and 800 LDS reads per thread)
    float x[FN];
    // The kw taps of one (co, ci, row) are ONE contiguous run in the OIHW tensor (4-byte aligned only: kw is odd), and the
    // runs of neighbouring lanes are far apart: read it with 16-byte loads (7 instead of 25 requests of 64 different
    // lines each), then place / flip in registers.  The run may be read up to 3 floats past its end -- inside the tensor
    // except for its very last run, where the tail is fetched element-wise.
    struct __attribute__((packed, aligned(4))) U4 { float v[4]; };
    float t[28];
    {
        const size_t run = !transposed ? (((size_t)co * cin_c + ci) * kh + tap) * kw
                                       : (((size_t)ci * cout_c + co) * kh + (kh - 1 - tap)) * kw;
        const float* wr = w.p[br] + run;
        const bool last = run + 28 > (size_t)cout_c * cin_c * kh * kw;     // would run past the tensor
        if (!last) {
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                if (4 * q < kw) {
                    const U4 u = *reinterpret_cast<const U4*>(wr + 4 * q);
                    t[4 * q] = u.v[0]; t[4 * q + 1] = u.v[1]; t[4 * q + 2] = u.v[2]; t[4 * q + 3] = u.v[3];
                } else {
                    t[4 * q] = t[4 * q + 1] = t[4 * q + 2] = t[4 * q + 3] = 0.f;
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 28; ++s) t[s] = (s < kw) ? wr[s] : 0.f;
        }
    }
    if (!transposed) {
#pragma unroll
        for (int s = 0; s < FN; ++s) x[s] = (s < 25 && s < kw) ? t[s] : 0.f;
    } else {
        // original tensor [cout_o = cin_c][cin_o = cout_c][kh][kw]; flipped in both kernel axes
#pragma unroll
        for (int s = 0; s < FN; ++s) {
            float v = 0.f;
            if (s < 25) {
#pragma unroll
                for (int k = 13; k <= 25; k += 2)                           // kw is one of 13, 15, ... 25: a static index per case
                    if (kw == k && s < k) v = t[k - 1 - s];
            }
            x[s] = v;
        }
    }
    float sr[FF], si[FF];
    rfft<FN>(x, sr, si);
    float* d = dst + (size_t)br * FS * per_wf + off;
    d[0] = sr[0];                                   // G(0)
    d[per_wf] = si[0];                              // G(FN / 2), real
    d[2 * per_wf] = 0.f;
#pragma unroll
    for (int f = 1; f < FF; ++f) {
        float* df = d + (size_t)(3 * f) * per_wf;
        const float gr = sr[f], gi = -si[f];
        df[0] = gr;
        df[per_wf] = gi;
        df[2 * per_wf] = gr + gi;
    }
}

// slab [br*FS + 3f+t][ks][tap][cout][cin] (R1 = dzr(x)xr, R2 = dzi(x)xi, R3 = (dzr-dzi)(x)(xr+xi)) -> dW[co][ci][tap][s]
// Block = one output channel co of one branch x 32 input channels: thread (ci_l = tid & 31, g = tid >> 5) handles kernel
// rows g, g+4, ... (one 32-thread group per kernel row in blocks of 32 kh threads, a single trip each, measured slower: 0.30
// against 0.27 ms per step -- fewer blocks in flight); the kw lags of every (ci, row) go through an LDS tile [32][kh][kw] that is then written out as ONE
// contiguous run of dW (32 * kh * kw floats) -- the first version had each thread write its kw floats at a stride of
// kh * kw floats from its neighbour (64 partial lines per store instruction): 0.45 ms per step for 0.1 GB of results.
// The launch also carries the bias gradient: blocks past `rblocks` (one per (branch, output channel)) sum slot 0 "re" of
// the zero-padded dZ blocks' spectrum (`sz`: each block's sum) over samples and positions -- as its own launch this
// was 16 us of latency five times per step; here it rides beside the slab reduction.
template <int FN>
__global__ void __launch_bounds__(128) wgrad_spec_reduce_kernel(const float* __restrict__ slab, MPtr2 dw, int KS, int cout,
                                                                int cin, int kh, int kw, int accumulate, int nbr,
                                                                const float* __restrict__ sz, MPtr2 db, int Bs, int P,
                                                                int rblocks) {
    constexpr int FF = FN / 2, FS = 3 * FF;
    extern __shared__ float tile[];                 // [32][kh][kw]
    if ((int)blockIdx.x >= rblocks) {
        const int co = ((int)blockIdx.x - rblocks) % cout, br = ((int)blockIdx.x - rblocks) / cout;
        float acc = 0.f;
        for (int b = 0; b < Bs; ++b) {
            const float* sp = sz + (((size_t)(br * Bs + b) * FS) * cout + co) * P;
            for (int p = threadIdx.x; p < P; p += 128) acc += sp[p];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if ((threadIdx.x & 63) == 0) tile[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float s = tile[0] + tile[1];
            float* d = db.p[br] + co;
            *d = accumulate ? *d + s : s;
        }
        return;
    }
    const int cib = cin / 32;
    int bid = blockIdx.x;
    const int cb = bid % cib; bid /= cib;
    const int co = bid % cout;
    const int br = bid / cout;
    const int ci_l = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int ci = cb * 32 + ci_l;
    const size_t per_ks = (size_t)kh * cout * cin;
    for (int tap = g; tap < kh; tap += 4) {
        const unsigned eoff = (unsigned)((tap * cout + co) * cin + ci);      // < per_ks: the only per-lane address part
        // eight frequencies (24 independent loads) per trip of the k loop: a k loop inside a per-frequency loop leaves
        // three loads in flight per wave and the kernel latency-bound at 1.1 TB/s
        float cr[FF], cim[FF];
        const float* sp0 = slab + (size_t)br * FS * KS * per_ks;           // uniform: scalar base + 32-bit lane offset
        const size_t slot = (size_t)KS * per_ks;
        // Slot planes through a buffer descriptor per block of eight frequencies (base = the block's first plane, lane offset =
        // the element, scalar offset = plane and split): with plain pointers the 96 plane addresses of the unrolled loops are
        // loop invariants of the split loop, 192 scalar registers the kernel does not have -- it spilled 236 of them to vector
        // lanes and moved them back and forth with 237 v_readlane + 236 v_writelane per tap.
        const unsigned slot_b = (unsigned)(slot * sizeof(float)), ks_b = (unsigned)(per_ks * sizeof(float));
#pragma unroll
        for (int f0 = 0; f0 < FF; f0 += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) cr[f0 + j] = cim[f0 + j] = 0.f;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(sp0 + (size_t)(3 * f0) * slot), 0, 0x7fffffff, 0x00020000);
            for (int k = 0; k < KS; ++k) {
                unsigned sb = slot_b;
                asm volatile("" : "+s"(sb));                 // the plane offsets are computed per trip, not kept
                const unsigned kb = (unsigned)k * ks_b;
                float r[24];
#pragma unroll
                for (int j = 0; j < 24; ++j) r[j] = spec_load(rs, eoff * 4u, (unsigned)j * sb + kb);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float r1 = r[3 * j], r2 = r[3 * j + 1], r3 = r[3 * j + 2];
                    if (f0 + j == 0) { cr[0] += r1; cim[0] += r2; }                  // DC and Nyquist, both real
                    else { cr[f0 + j] += r1 + r2; cim[f0 + j] += r3 - r1 + r2; }    // conj(dz) * x
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the next trip's loads below this one's adds
            }
        }
        // dW[s] = (1/FN) sum_f C[f] e^{+2 pi i f s / FN} (C Hermitian): the inverse real FFT, lags 0 .. kw-1 kept
        float lag[FN];
        irfft<FN>(cr, cim, lag);
        float* t = tile + ((size_t)ci_l * kh + tap) * kw;
#pragma unroll
        for (int s = 0; s < 25; ++s)
            if (s < kw) t[s] = lag[s];
    }
    __syncthreads();
    const int n = 32 * kh * kw;
    float* d = dw.p[br] + ((size_t)co * cin + cb * 32) * kh * kw;
    for (int i = threadIdx.x; i < n; i += 128) d[i] = accumulate ? d[i] + tile[i] : tile[i];
}

// =====================================================================================================
//                                          instances + dispatch
// =====================================================================================================
//                  CIN COUT KH KW(taps) H  MT RT CT WM WR WC CK POOL EPI DILMAX HET
using FF2 = ConvCfg<64, 64, 1, 7, 1, 2, 1, 2, 1, 1, 4, 8, 1, 1, FDIL>;
using FF3 = ConvCfg<64, 128, 1, 5, 1, 4, 1, 1, 1, 1, 12, 8, 1, 1, FDIL, 8>;
using FF4 = ConvCfg<128, 128, 1, 9, 1, 4, 1, 1, 1, 1, 12, 8, 1, 1, FDIL, 8>;
using FF5 = ConvCfg<128, 256, 1, 3, 1, 2, 1, 1, 4, 1, 3, 8, 1, 1, FDIL>;
using FF6 = ConvCfg<256, 256, 1, 5, 1, 2, 1, 1, 4, 1, 3, 4, 1, 1, FDIL>;
// narrower variants for short flattened axes (e.g. T = 313 / 626: P = 224 for cv3/cv4, 56..64 for cv5/cv6)
// cv2 at P = 960 = 5 x 192 (T = 512) or 576 = 3 x 192 (T = 313): 192-column blocks of 4 waves x (1 channel tile x 3 column
// tiles) fill every SIMD in every block, where the 256-column blocks leave a wave idle in the last one: 1337 -> 1277 us
using FF2a = ConvCfg<64, 64, 1, 7, 1, 1, 1, 3, 2, 1, 2, 8, 1, 1, FDIL>;
using FD2a = FF2a;
// (measured and not kept, same step: 320-column blocks of 4 waves x (1 x 5) for cv2: 1304 us against 1265; 4-wave blocks
// of (2 x 3) / (1 x 3) tiles per wave for cv5 forward / backward-data: 449 / 450 us against 436 / 434; cv3 forward as
// 4 waves x (1 x 3) or (2 x 3): 749 / 769 us against the two-kind layout's 699; cv3 backward-data with 16 input channels per
// stage: 746 us against 683)
using FF3a = ConvCfg<64, 128, 1, 5, 1, 4, 1, 1, 1, 1, 8, 8, 1, 1, FDIL, 4>;    // 7 column tiles, 8 waves
using FF4a = ConvCfg<128, 128, 1, 9, 1, 4, 1, 1, 1, 1, 8, 8, 1, 1, FDIL, 4>;
using FF5a = ConvCfg<128, 256, 1, 3, 1, 2, 1, 1, 4, 1, 2, 16, 1, 1, FDIL>;
using FF6a = ConvCfg<256, 256, 1, 5, 1, 2, 1, 1, 4, 1, 2, 8, 1, 1, FDIL>;
using FD3a = ConvCfg<128, 64, 1, 5, 1, 2, 1, 1, 1, 1, 8, 8, 1, 1, FDIL>;
using FD4a = FF4a;
using FD5a = ConvCfg<256, 128, 1, 3, 1, 1, 1, 1, 4, 1, 2, 16, 1, 1, FDIL>;
using FD6a = FF6a;
using FD2 = FF2;
// cv3 backward-data (2 channel tiles x 11 column tiles): two blocks of 192 columns, 4 waves x (1 channel tile x 3 column
// tiles) like FF2a, the dead twelfth column tile skipped (conv_core's live-tile K loop): 752 -> 683 us against the 11-wave
// one-block layout, whose SIMDs carried 3 / 3 / 3 / 2 waves
using FD3 = ConvCfg<128, 64, 1, 5, 1, 1, 1, 3, 2, 1, 2, 8, 1, 1, FDIL>;
using FD4 = FF4;
using FD5 = ConvCfg<256, 128, 1, 3, 1, 1, 1, 1, 4, 1, 3, 16, 1, 1, FDIL>;
using FD6 = FF6;
// cv3 / cv4 on 128-point windows (use_win128): P = 32 NB with NB <= 6 windows per row segment -- 160 columns at T = 512,
// 96 at the reference's T = 313 -- and 192 slots per sample.  Four channel tiles x five column tiles = 20 tiles: four waves of
// (1 channel tile x 5 column tiles), five tiles on every SIMD, no two-kind layout, no dead twelfth tile; 4-channel stages keep
// the operands at 45 KB (three workgroups per CU).  `a`: three column tiles per wave for P = 96 / 192 / 288.
//                   CIN COUT KH KW(taps) H  MT RT CT WM WR WC CK POOL EPI DILMAX
using FF3w = ConvCfg<64, 128, 1, 5, 1, 1, 1, 5, 4, 1, 1, 8, 1, 1, FDIL128>;
using FF3wa = ConvCfg<64, 128, 1, 5, 1, 1, 1, 3, 4, 1, 1, 8, 1, 1, FDIL128>;
using FF4w = ConvCfg<128, 128, 1, 9, 1, 1, 1, 5, 4, 1, 1, 4, 1, 1, FDIL128>;
using FF4wa = ConvCfg<128, 128, 1, 9, 1, 1, 1, 3, 4, 1, 1, 4, 1, 1, FDIL128>;
// cv7 / cv8 (see geom(): remap): one GEMM per (branch, slot), 32-channel stages, a wave = one channel tile x up to seven column
// tiles (128 columns per block: T = 313 has P = 128, T = 512 P = 208 = 4 + 3 live tiles; two workgroups per CU; seven tiles per
// wave in one block: 190 workgroups of one wave per SIMD, 45 us against 30)
//                  CIN COUT KH KW(taps) H  MT RT CT WM WR WC CK POOL EPI DILMAX
using FF7 = ConvCfg<256, 128, 1, 1, 1, 1, 1, 4, 4, 1, 1, 32, 1, 1, FDIL>;
using FD7 = ConvCfg<128, 256, 1, 1, 1, 1, 1, 4, 8, 1, 1, 32, 1, 1, FDIL>;
using FF8 = ConvCfg<128, 128, 1, 1, 1, 1, 1, 4, 4, 1, 1, 32, 1, 1, FDIL>;
using FD8 = FF8;
using FD4w = FF4w;
using FD4wa = FF4wa;
// cv3 backward-data (two channel tiles x five column tiles): four waves of (1 x 3) over 192 columns, the sixth column tile
// dead (conv_core's live-tile K loop skips it).  Measured per step at T = 512, same box: 22.73 ms against 22.81 for two waves
// of (1 x 5), 22.79 for one wave of (2 x 5), 22.84 for two waves of (2 x 3); cv3 forward as two waves of (2 x 5): 22.88; cv4
// as two waves of (2 x 5) (352 registers, one wave per SIMD): 23.08.
using FD3w = ConvCfg<128, 64, 1, 5, 1, 1, 1, 3, 2, 1, 2, 8, 1, 1, FDIL128>;
// cv2 on 128-point windows, one-segment rows only (use_win128): T = 313 -> 3 x 64 frequency-windows against 7 x 32, T = 626
// -> 6 x 64 against 13 x 32 in two segments (the step 30.1 -> 28.7 ms); T = 512 is a tie and stays on 64.  P = 96 NB = 288 /
// 576: four waves of (1 x 3) over 192-column blocks (P = 288: the second block half dead, its tiles skipped).  Measured per
// step at T = 313: 15.89 ms, against 16.01 for two-wave 96-column blocks, 15.99 for six waves over 288 columns, 16.02 for
// three waves of (2 x 3); 64-point windows: 16.0.  Long rows (the 30 s fine-tune clips, batch of one) stay on 64-point
// windows: with two samples per launch the doubled operators (weight pack, slab reduce) cost more than the GEMMs gain
// (6.85 against 6.72 ms per step).
using FF2w = ConvCfg<64, 64, 1, 7, 1, 1, 1, 3, 2, 1, 2, 8, 1, 1, FDIL128>;
using FD2w = FF2w;
// P = 288 (T = 313): 2 x 9 tiles do not split evenly over four waves; 320-column blocks of four waves x (1 x 5) give the waves
// 5 / 5 / 4 / 4 live tiles (the tenth column tile is dead) instead of 3 / 3 / 3 / 3 + 3 / 3 / 0 / 0 in two 192-column blocks
using FF2wb = ConvCfg<64, 64, 1, 7, 1, 1, 1, 5, 2, 1, 2, 8, 1, 1, FDIL128>;
using FD2wb = FF2wb;
// P = 96 (T = 313: 2 x 3 tiles): four waves over 128 columns with 2 / 2 / 1 / 1 live tiles (fourth column tile dead) -- T = 313
// per step 15.40-15.44 ms against 15.47-15.49 for two-wave workgroups of (1 x 3)
using FD3wa = ConvCfg<128, 64, 1, 5, 1, 1, 1, 2, 2, 1, 2, 8, 1, 1, FDIL128>;
//                 CIN COUT KH KW(taps) H  MT NS WMW WQ R DILMAX WCI
using FW2 = WgCfg<64, 64, 1, 7, 1, 1, 7, 2, 1, 1, FDILW, 2, 32>;
using FW3 = WgCfg<64, 128, 1, 5, 1, 1, 5, 4, 1, 1, FDILW, 2, 32>;
using FW4 = WgCfg<128, 128, 1, 9, 1, 2, 3, 2, 3, 1, FDILW, 2, 32>;
using FW5 = WgCfg<128, 256, 1, 3, 1, 2, 3, 2, 1, 1, FDILW, 2, 32>;
using FW6 = WgCfg<256, 256, 1, 5, 1, 2, 5, 4, 1, 1, FDILW, 2, 32>;
// cv7 / cv8 as one GEMM per (branch, slot) (see geom(): remap): a single "tap"; 8 waves = 2 output-channel groups of 2 tiles x 4
// input-channel tiles, 128 x 128 of dW per workgroup
using FW7 = WgCfg<256, 128, 1, 1, 1, 2, 1, 2, 1, 1, FDILW, 4, 32>;
using FW8 = WgCfg<128, 128, 1, 1, 1, 2, 1, 2, 1, 1, FDILW, 4, 32>;

struct FftGeom {
    int FN;     // window length: 64 or 128 samples
    int FS;     // spectrum / weight slots per sample: 3 * FN / 2
    int V;      // valid outputs per window
    int NBT;    // windows per image row
    int NS;     // segments a row is cut into (each a separate spectral sample)
    int NB;     // windows per segment (<= FDILW): the row taps of the GEMM stage are NB positions apart
    int P;      // flattened (row, window) positions per channel of one spectral sample
    int nb;     // spectral samples = nbr * B * NS
    int Bs;     // spectral samples per branch = B * NS
    int remap;  // cv7 / cv8 (one image row, kh = 1): see geom()
    int H;      // rows of a spectral sample (remap: the samples of a branch)
};

static int fft_layer_of(const ConvArgs& a) {
    struct L { int cin, cout, kh, kw, H; };
    static const L tab[] = {{64, 64, 7, 13, 96}, {64, 128, 5, 15, 32}, {128, 128, 9, 17, 32}, {128, 256, 3, 19, 8},
                            {256, 256, 5, 21, 8}, {256, 128, 1, 23, 1}, {128, 128, 1, 25, 1}};
    for (int i = 0; i < 7; ++i)
        if (tab[i].cin == a.cin && tab[i].cout == a.cout && tab[i].kh == a.kh && tab[i].kw == a.kw && tab[i].H == a.H)
            return i + 2;
    return -1;
}

// 128-point windows for cv2 / cv3 / cv4 wherever they need fewer frequency-windows per row than 64-point ones (2 per window
// against 1: T = 512: 2 x 5 against 11; the reference's T = 313: 2 x 3 against 7; rows under ~100 frames stay on 64).  A
// function of (layer, T) alone, never of the batch: a clip's sums must not depend on the batch it is in.
// zns_test_set_tuning(ZNS_TUNE_WIN128, 1 / -1) forces / forbids it (tests compare the two).
static bool use_win128(const ConvArgs& a) {
    const int l = fft_layer_of(a);
    if (l != 2 && l != 3 && l != 4) return false;
    const int m = tuning(ZNS_TUNE_WIN128);
    if (m < 0) return false;
    if (m > 0) return true;
    const int n128 = ceil_div(a.T, 129 - a.kw);
    if (l == 2 && n128 > FDIL128) return false;          // cv2: one-segment rows only (see FF2w)
    return 2 * n128 < ceil_div(a.T, 65 - a.kw);
}

// cv7 / cv8 have ONE image row and kh = 1: per sample the flattened (row, window) axis would be 8 .. 13 positions, a sliver of
// a 32-column tile.  Without row taps nothing couples the positions of a launch, so the samples of a branch become the ROWS of
// one spectral sample per branch -- "H" = B, P = B * NBT (T = 512: 208, the reference's T = 313: 128) -- and every (branch,
// slot) is one plain GEMM  [cout x cin] x [cin x P]  on the matrix cores: 96 real MACs per 42- / 40-sample window instead of
// 23 / 25 per sample, 10x fewer than the direct form these layers ran in until round 5.  The transforms address the
// [B][C][T] tensors through a row remap (spec_fwd_kernel / spec_inv_kernel: remap); rows are never cut (no taps: any NB).
constexpr int ONE_ROW_MAX_NB = 256;     // the inverse transform's block holds a whole row of windows

static FftGeom geom(const ConvArgs& a) {
    FftGeom g;
    g.remap = fft_layer_of(a) >= 7 ? 1 : 0;
    if (g.remap) {
        g.FN = 64;
        g.FS = 96;
        g.V = g.FN - a.kw + 1;
        g.NBT = ceil_div(a.T, g.V);
        g.NS = 1;
        g.NB = g.NBT;
        g.H = a.B;
        g.P = a.B * g.NB;
        g.Bs = 1;
        g.nb = a.nbr;
        return g;
    }
    g.H = a.H;
    g.FN = use_win128(a) ? 128 : 64;
    g.FS = 3 * (g.FN / 2);
    g.V = g.FN - a.kw + 1;
    g.NBT = ceil_div(a.T, g.V);
    g.NS = ceil_div(g.NBT, g.FN == 128 ? FDIL128 : FDILW);
    g.NB = ceil_div(g.NBT, g.NS);
    g.P = a.H * g.NB;
    g.Bs = a.B * g.NS;
    g.nb = a.nbr * g.Bs;
    return g;
}

bool fft_supported(const ConvArgs& a) {
    const int l = fft_layer_of(a);
    if (l < 0 || a.nbr < 1 || a.nbr > 2) return false;
    if (l >= 7 && (tuning(ZNS_TUNE_ONE_ROW_DIRECT) > 0 || ceil_div(a.T, 65 - a.kw) > ONE_ROW_MAX_NB)) return false;
    return a.T >= 32 && a.B >= 1;
}

bool fft_wgrad_supported(const ConvArgs& a) { return fft_supported(a); }

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct FftWs {
    size_t s_in, s_out, wf, slab, total;
};

template <class W>
static int fwg_pick_ks(const ConvArgs& a, const FftGeom& g, int* tiles_total, int* tps) {
    ConvArgs t = a;
    t.B = g.Bs;          // every segment of every sample contributes K tiles
    t.T = g.P;           // K tiles run over the flattened axis
    t.nbr = a.nbr * g.FS;  // one accumulation per (branch, frequency, Karatsuba part)
    return wg_pick_ks<W>(t, tiles_total, tps);
}

template <class W>
static size_t fwg_slab_floats(const ConvArgs& a, const FftGeom& g) {
    int tt, tps;
    const int ks = fwg_pick_ks<W>(a, g, &tt, &tps);
    return (size_t)a.nbr * g.FS * ks * W::Q * W::COUT * W::CIN;
}

static FftWs fft_ws(const ConvArgs& a) {
    const FftGeom g = geom(a);
    FftWs w{};
    const size_t cmax = (size_t)(a.cin > a.cout ? a.cin : a.cout);
    const size_t spec = (size_t)g.nb * g.FS * cmax * g.P * sizeof(float);
    w.s_in = 0;
    w.s_out = align_up(spec);
    w.wf = w.s_out + align_up(spec);
    const size_t wfb = (size_t)a.nbr * g.FS * a.cout * a.cin * a.kh * sizeof(float);
    size_t slab = 0;
    switch (fft_layer_of(a)) {
        case 2: slab = fwg_slab_floats<FW2>(a, g); break;
        case 3: slab = fwg_slab_floats<FW3>(a, g); break;
        case 4: slab = fwg_slab_floats<FW4>(a, g); break;
        case 5: slab = fwg_slab_floats<FW5>(a, g); break;
        case 6: slab = fwg_slab_floats<FW6>(a, g); break;
        case 7: slab = fwg_slab_floats<FW7>(a, g); break;
        case 8: slab = fwg_slab_floats<FW8>(a, g); break;
    }
    slab *= sizeof(float);
    w.slab = w.wf;   // backward-weights does not need the packed frequency weights: same region
    w.total = w.wf + align_up(wfb > slab ? wfb : slab);
    return w;
}

size_t fft_workspace_bytes(const ConvArgs& a) { return fft_supported(a) ? fft_ws(a).total : 0; }
size_t fft_spectrum_bytes(const ConvArgs& a) {
    const FftGeom g = geom(a);
    return (size_t)g.nb * g.FS * a.cin * g.P * sizeof(float);
}
size_t fft_dz_spectrum_bytes(const ConvArgs& a) {
    const FftGeom g = geom(a);
    return (size_t)g.nb * g.FS * a.cout * g.P * sizeof(float);
}

static int run_spec_fwd(const float* x, float* S, int nb, int C, int H, int T, const FftGeom& g, int off, int valid,
                        float third_sign, hipStream_t st, float* S2 = nullptr) {
    const int remap = g.remap;
    if (g.FN == 128) {
        const int chunks = ceil_div(C * g.P, win_per_block<128>());
        hipLaunchKernelGGL(spec_fwd_kernel<128>, dim3((unsigned)((size_t)nb * chunks)), dim3(256), 0, st, x, S,
                           C, H, T, g.V, g.NB, off, valid, chunks, third_sign, g.NS, g.NBT, S2, remap);
    } else {
        const int chunks = ceil_div(C * g.P, 256);
        hipLaunchKernelGGL(spec_fwd_kernel<64>, dim3((unsigned)((size_t)nb * chunks)), dim3(256), 0, st, x, S, C, H, T, g.V,
                           g.NB, off, valid, chunks, third_sign, g.NS, g.NBT, S2, remap);
    }
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

template <int MODE, int FN>
static int run_spec_inv_n(const float* S, float* out, int nb, int C, int H, int T, const FftGeom& g, InvEpi E,
                          hipStream_t st) {
    constexpr int NTHR = inv_threads<FN>(), WINB = inv_win_per_block<FN>();
    const int pool = MODE == 1 ? E.pool : 1;
    const int group = pool * g.NB;
    const int WPB = (WINB / group) * group;
    const int chunks = ceil_div(C * g.P, WPB);
    auto al = [](const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; };
    E.remap = g.remap;
    int vw = 1;   // samples per thread in the epilogue: the widest vector V, T and every pointer allow
    for (int cand : {4, 2})
        if (vw == 1 && g.V % cand == 0 && T % cand == 0 && al(out, 4 * cand) && al(E.xmask, 4 * cand) && al(E.argmax, cand) &&
            al(E.prev_argmax, cand))
            vw = cand;
    // V % 4 == 2 on a one-segment row: 16-byte chunks along the row instead of 8-byte chunks along the windows
    const bool row = !g.remap && vw == 2 && g.NS == 1 && T % 4 == 0 && al(out, 16) && al(E.xmask, 16) && al(E.argmax, 4) &&
                     al(E.prev_argmax, 4);
    E.mV = div_magic(row ? T / 4 : g.V / vw);
    E.mNB = div_magic(g.NB);
    E.mHq = div_magic(H / pool);
    const size_t lds = (size_t)WINB * (vw > 1 ? g.V : (g.V | 1)) * sizeof(float);
    const dim3 grid((unsigned)((size_t)nb * chunks));
    // everything else on a one-segment row (odd T, e.g. the reference's 313 frames): 16-byte chunks of the block's flat output
    const bool flat = !g.remap && !row && vw < 4 && g.NS == 1 && al(out, 16) && al(E.argmax, 4) && al(E.xmask, 16);
    E.mT = div_magic(T);
    E.mPP = div_magic(MODE == 2 ? E.prev_pool : 1);
    if (flat) {
        const size_t lds_f = (size_t)WINB * g.V * sizeof(float);      // LDS row stride V (as VW > 1)
        hipLaunchKernelGGL((spec_inv_kernel<MODE, 4, false, true, FN>), grid, dim3(NTHR), lds_f, st, S, out, C, H, T, g.V, g.NB,
                           chunks, WPB, g.NS, E);
    } else if (row)
        hipLaunchKernelGGL((spec_inv_kernel<MODE, 4, true, false, FN>), grid, dim3(NTHR), lds, st, S, out, C, H, T, g.V, g.NB, chunks, WPB, g.NS, E);
    else if (vw == 4)
        hipLaunchKernelGGL((spec_inv_kernel<MODE, 4, false, false, FN>), grid, dim3(NTHR), lds, st, S, out, C, H, T, g.V, g.NB, chunks, WPB, g.NS, E);
    else if (vw == 2)
        hipLaunchKernelGGL((spec_inv_kernel<MODE, 2, false, false, FN>), grid, dim3(NTHR), lds, st, S, out, C, H, T, g.V, g.NB, chunks, WPB, g.NS, E);
    else
        hipLaunchKernelGGL((spec_inv_kernel<MODE, 1, false, false, FN>), grid, dim3(NTHR), lds, st, S, out, C, H, T, g.V, g.NB, chunks, WPB, g.NS, E);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}
template <int MODE>
static int run_spec_inv(const float* S, float* out, int nb, int C, int H, int T, const FftGeom& g, InvEpi E,
                        hipStream_t st) {
    return g.FN == 128 ? run_spec_inv_n<MODE, 128>(S, out, nb, C, H, T, g, E, st)
                       : run_spec_inv_n<MODE, 64>(S, out, nb, C, H, T, g, E, st);
}

template <class C>
static int set_lds_f(void (*kern)(CoreArgs)) {
    static bool done[ZNS_MAX_DEVICES] = {};
    const int d = current_device_slot();
    if (!done[d]) {
        ZNS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        done[d] = true;
    }
    return ZNS_OK;
}
template <class C>
static int set_lds_fw(void (*kern)(WgArgs)) {
    static bool done[ZNS_MAX_DEVICES] = {};
    const int d = current_device_slot();
    if (!done[d]) {
        ZNS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        done[d] = true;
    }
    return ZNS_OK;
}

// frequency weights of `w` in conv_core<C>'s slab order (see pack_freq_weights_kernel)
template <class C>
static int pack_fconv(const ConvArgs& a, Ptr2 w, int transposed, float* wf, hipStream_t st) {
    const long long per_wf = (long long)C::COUT * C::CIN * C::KW;
    const long long total = per_wf * a.nbr;
    const int blocks = (int)((total + 255) / 256);
    if (C::DILMAX == FDIL128)
        hipLaunchKernelGGL(pack_freq_weights_kernel<128>, dim3(blocks), dim3(256), 0, st, w, wf, C::CIN, C::COUT, C::KW, a.kw,
                           C::COB, C::CK, C::MT, C::WM, transposed, per_wf, a.nbr);
    else
        hipLaunchKernelGGL(pack_freq_weights_kernel<64>, dim3(blocks), dim3(256), 0, st, w, wf, C::CIN, C::COUT, C::KW, a.kw,
                           C::COB, C::CK, C::MT, C::WM, transposed, per_wf, a.nbr);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

// S_in [nb*FF][CIN'][P] -> S_out [nb*FF][COUT'][P] with the frequency weights of `w` (wpk: already packed, or NULL)
template <class C>
static int run_fconv(const ConvArgs& a, const FftGeom& g, Ptr2 w, int transposed, const float* s_in, float* s_out,
                     float* wf, const float* wpk, int probe_kind, hipStream_t st) {
    int rc = ZNS_OK;
    if (wpk == nullptr) {
        rc = pack_fconv<C>(a, w, transposed, wf, st);
        if (rc) return rc;
        wpk = wf;
    }
    rc = set_lds_f<C>(conv_core<C>);
    if (rc) return rc;
    CoreArgs A{};
    A.x = s_in;
    A.wpk = wpk;
    A.dx = s_out;
    A.xmask = nullptr;
    A.inv_keep = 1.f;
    A.prev_pool = 1;
    A.B = g.Bs;
    A.T = g.P;
    A.N = g.nb * g.FS;
    A.F = g.FS;
    A.dil = g.NB;
    A.ncb = ceil_div(g.P, C::WB);
    if ((C::DILMAX == FDIL128) != (g.FN == 128) || (C::KW > 1 && g.NB > C::DILMAX)) {
        set_error("conv (fft): tile configuration and window length disagree (FN %d, NB %d)", g.FN, g.NB);
        return ZNS_ERR_ARG;
    }
    // two-kind layout: consecutive (sample, slot) items per workgroup (one block per item, 16-byte aligned plain stores): the
    // largest of 4 / 3 / 2 that divides the items and still leaves a workgroup for every CU (ZNS_TUNE_IPW overrides: tuning)
    A.ipw = 1;
    if constexpr (C::HET > 0 && C::EPI_SEP) {
        const int ipw_env = tuning(ZNS_TUNE_IPW);
        if (A.ncb == 1 && C::COUT == C::COB && (g.P & 3) == 0 && (reinterpret_cast<uintptr_t>(s_out) & 15) == 0) {
            if (ipw_env > 0) {
                A.ipw = ipw_env;
            } else {
                for (int cand : {4, 3, 2})
                    if (A.ipw == 1 && A.N % cand == 0 && A.N / cand >= device_cus()) A.ipw = cand;
            }
        }
    }
    A.slot_major = tuning(ZNS_TUNE_IPW) == -2 ? 0 : 1;       // (-2: item order, for A/B timing)
    const long long blocks2 = A.ipw > 1 ? (long long)ceil_div(A.N, A.ipw)
                                        : (long long)A.N * A.ncb * (C::COUT / C::COB);
    const bool probe = probe_match(probe_kind, a.cin, a.cout, a.kh);
    const int slot = probe ? g_probe.count % g_probe.n : 0;
    if (probe) ZNS_HIP(hipEventRecord(g_probe.start[slot], st));
    hipLaunchKernelGGL(conv_core<C>, dim3((unsigned)blocks2), dim3(C::NT), C::LDS_BYTES, st, A);
    ZNS_LAUNCH_CHECK();
    if (probe) { ZNS_HIP(hipEventRecord(g_probe.stop[slot], st)); ++g_probe.count; }
    return ZNS_OK;
}

template <class C>
static int fft_fwd_t(const ConvArgs& a, const float* x, Ptr2 w, Ptr2 bias, float* y, uint8_t* argmax, float drop_p,
                     uint32_t seed, int layer, float* xspec, const float* wpk, void* ws, size_t ws_bytes, hipStream_t st) {
    const FftGeom g = geom(a);
    const FftWs L = fft_ws(a);
    if (ws_bytes < L.total) { set_error("conv fwd (fft): workspace %zu < %zu", ws_bytes, L.total); return ZNS_ERR_WORKSPACE; }
    char* base = (char*)ws;
    float* s_in = xspec ? xspec : (float*)(base + L.s_in);   // kept by the caller for backward-weights when given
    float* s_out = (float*)(base + L.s_out);
    float* wf = (float*)(base + L.wf);
    int rc = run_spec_fwd(x, s_in, g.nb, a.cin, g.H, a.T, g, a.kw / 2, g.FN, 1.f, st);
    if (rc) return rc;
    rc = run_fconv<C>(a, g, w, 0, s_in, s_out, wf, wpk, 0, st);
    if (rc) return rc;
    InvEpi E{};
    E.bias = bias;
    E.argmax = argmax;
    E.pool = a.pool;
    E.B = g.remap ? 1 : a.B;          // samples of a branch per spectral-sample index
    E.thresh = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    E.inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
    E.seed = seed;
    E.layer = layer;
    return run_spec_inv<1>(s_out, y, g.nb, a.cout, g.H, a.T, g, E, st);
}

template <class C>
static int fft_dgrad_t(const ConvArgs& a, const float* dz, Ptr2 w, float* dx, const float* x, float inv_keep,
                       const uint8_t* prev_argmax, int prev_pool, const float* dzspec, const float* wpk, void* ws,
                       size_t ws_bytes, hipStream_t st) {
    const FftGeom g = geom(a);
    const FftWs L = fft_ws(a);
    if (ws_bytes < L.total) { set_error("conv dgrad (fft): workspace %zu < %zu", ws_bytes, L.total); return ZNS_ERR_WORKSPACE; }
    char* base = (char*)ws;
    float* s_in = (float*)(base + L.s_in);
    float* s_out = (float*)(base + L.s_out);
    float* wf = (float*)(base + L.wf);
    int rc = ZNS_OK;
    if (dzspec) s_in = const_cast<float*>(dzspec);           // written by the preceding backward-weights call
    else rc = run_spec_fwd(dz, s_in, g.nb, a.cout, g.H, a.T, g, a.kw / 2, g.FN, 1.f, st);
    if (rc) return rc;
    rc = run_fconv<C>(a, g, w, 1, s_in, s_out, wf, wpk, 1, st);
    if (rc) return rc;
    InvEpi E{};
    E.xmask = x;
    E.inv_keep = inv_keep;
    E.prev_argmax = prev_argmax;
    E.prev_pool = prev_pool;
    return run_spec_inv<2>(s_out, dx, g.nb, a.cin, g.H, a.T, g, E, st);
}

template <class W>
static int fft_wgrad_t(const ConvArgs& a, const float* dz, const float* x, MPtr2 dw, MPtr2 db, int accumulate,
                       const float* xspec, float* dzspec_out, void* ws, size_t ws_bytes, hipStream_t st) {
    const FftGeom g = geom(a);
    const FftWs L = fft_ws(a);
    if (ws_bytes < L.total) { set_error("conv wgrad (fft): workspace %zu < %zu", ws_bytes, L.total); return ZNS_ERR_WORKSPACE; }
    char* base = (char*)ws;
    float* sx = (float*)(base + L.s_in);
    float* sz = (float*)(base + L.s_out);
    float* slab = (float*)(base + L.slab);
    int rc = ZNS_OK;
    if (xspec) sx = const_cast<float*>(xspec);                                         // saved by the forward call
    else rc = run_spec_fwd(x, sx, g.nb, a.cin, g.H, a.T, g, a.kw / 2, g.FN, 1.f, st);   // windows of the input: (xr, xi, xr+xi)
    if (rc) return rc;
    if (dzspec_out)   // one pass over dZ: overlapped windows for the backward-data call + the zero-padded V-blocks
        rc = run_spec_fwd(dz, dzspec_out, g.nb, a.cout, g.H, a.T, g, a.kw / 2, g.FN, 1.f, st, sz);
    else
        rc = run_spec_fwd(dz, sz, g.nb, a.cout, g.H, a.T, g, 0, g.V, -1.f, st);  // zero-padded V-blocks of dZ: (r, i, r-i)
    if (rc) return rc;
    rc = set_lds_fw<W>(wgrad_core<W>);
    if (rc) return rc;
    WgArgs A{};
    A.dz = sz;
    A.x = sx;
    A.slab = slab;
    A.B = g.Bs;
    A.T = g.P;
    A.ncb = ceil_div(g.P, W::CW);
    A.F = g.FS;
    A.dil = g.NB;
    A.KS = fwg_pick_ks<W>(a, g, &A.tiles_total, &A.tiles_per_split);
    const long long blocks = (long long)W::NQG * (W::CIN / W::CIB) * (W::COUT / W::COB) * A.KS * a.nbr * g.FS;
    const bool probe = probe_match(2, a.cin, a.cout, a.kh);
    const int slot = probe ? g_probe.count % g_probe.n : 0;
    if (probe) ZNS_HIP(hipEventRecord(g_probe.start[slot], st));
    hipLaunchKernelGGL(wgrad_core<W>, dim3((unsigned)blocks), dim3(W::NT), W::LDS_BYTES, st, A);
    ZNS_LAUNCH_CHECK();
    if (probe) { ZNS_HIP(hipEventRecord(g_probe.stop[slot], st)); ++g_probe.count; }
    const int rblocks = a.nbr * a.cout * (a.cin / 32);
    if (g.FN == 128)
        hipLaunchKernelGGL(wgrad_spec_reduce_kernel<128>, dim3(rblocks + a.nbr * a.cout), dim3(128),
                           (size_t)32 * a.kh * a.kw * sizeof(float), st, (const float*)slab, dw, A.KS, a.cout, a.cin, a.kh, a.kw,
                           accumulate, a.nbr, (const float*)sz, db, g.Bs, g.P, rblocks);
    else
        hipLaunchKernelGGL(wgrad_spec_reduce_kernel<64>, dim3(rblocks + a.nbr * a.cout), dim3(128),
                           (size_t)32 * a.kh * a.kw * sizeof(float), st, (const float*)slab, dw, A.KS, a.cout, a.cin, a.kh, a.kw,
                           accumulate, a.nbr, (const float*)sz, db, g.Bs, g.P, rblocks);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

// true when the narrow workgroup (WBa columns) covers the flattened axis with fewer padded columns than the wide one
template <class Wide, class Narrow>
static bool use_narrow(const ConvArgs& a) {
    const int P = geom(a).P;
    return (long long)ceil_div(P, Narrow::WB) * Narrow::WB < (long long)ceil_div(P, Wide::WB) * Wide::WB;
}

// One dispatcher for everything that depends on the (layer, tile variant) pair: F(tag<Cfg>) is called with the forward
// (transposed == 0) or backward-data configuration the geometry runs on.
template <class T> struct cfg_tag { typedef T type; };
template <class Fn>
static int with_cfg(const ConvArgs& a, int transposed, Fn&& f) {
    switch (fft_layer_of(a)) {
        case 2:
            if (use_win128(a)) {
                // (T = 313 per step: 15.43-15.49 ms with the 320-column blocks against 15.81-15.87 with 192-column ones)
                if (use_narrow<FF2wb, FF2w>(a)) return transposed ? f(cfg_tag<FD2w>{}) : f(cfg_tag<FF2w>{});
                return transposed ? f(cfg_tag<FD2wb>{}) : f(cfg_tag<FF2wb>{});
            }
            // (a tie -- T = 313: P = 672, 768 padded columns either way -- stays with the 256-column blocks: 993 us against 1006)
            if (use_narrow<FF2, FF2a>(a)) return transposed ? f(cfg_tag<FD2a>{}) : f(cfg_tag<FF2a>{});
            return transposed ? f(cfg_tag<FD2>{}) : f(cfg_tag<FF2>{});
        case 3:
            if (use_win128(a)) {
                if (transposed) return use_narrow<FD3w, FD3wa>(a) ? f(cfg_tag<FD3wa>{}) : f(cfg_tag<FD3w>{});
                return use_narrow<FF3w, FF3wa>(a) ? f(cfg_tag<FF3wa>{}) : f(cfg_tag<FF3w>{});
            }
            if (transposed) return use_narrow<FD3, FD3a>(a) ? f(cfg_tag<FD3a>{}) : f(cfg_tag<FD3>{});
            return use_narrow<FF3, FF3a>(a) ? f(cfg_tag<FF3a>{}) : f(cfg_tag<FF3>{});
        case 4:
            if (use_win128(a)) {
                if (transposed) return use_narrow<FD4w, FD4wa>(a) ? f(cfg_tag<FD4wa>{}) : f(cfg_tag<FD4w>{});
                return use_narrow<FF4w, FF4wa>(a) ? f(cfg_tag<FF4wa>{}) : f(cfg_tag<FF4w>{});
            }
            if (transposed) return use_narrow<FD4, FD4a>(a) ? f(cfg_tag<FD4a>{}) : f(cfg_tag<FD4>{});
            return use_narrow<FF4, FF4a>(a) ? f(cfg_tag<FF4a>{}) : f(cfg_tag<FF4>{});
        case 5:
            if (transposed) return use_narrow<FD5, FD5a>(a) ? f(cfg_tag<FD5a>{}) : f(cfg_tag<FD5>{});
            return use_narrow<FF5, FF5a>(a) ? f(cfg_tag<FF5a>{}) : f(cfg_tag<FF5>{});
        case 6:
            if (transposed) return use_narrow<FD6, FD6a>(a) ? f(cfg_tag<FD6a>{}) : f(cfg_tag<FD6>{});
            return use_narrow<FF6, FF6a>(a) ? f(cfg_tag<FF6a>{}) : f(cfg_tag<FF6>{});
        case 7: return transposed ? f(cfg_tag<FD7>{}) : f(cfg_tag<FF7>{});
        case 8: return transposed ? f(cfg_tag<FD8>{}) : f(cfg_tag<FF8>{});
    }
    return ZNS_ERR_ARG;
}

int fft_fwd(const ConvArgs& a, const float* x, Ptr2 w, Ptr2 bias, float* y, uint8_t* argmax, float drop_p,
            uint32_t seed, int layer, float* xspec, const float* wpk, void* ws, size_t ws_bytes, hipStream_t st) {
    return with_cfg(a, 0, [&](auto t) {
        return fft_fwd_t<typename decltype(t)::type>(a, x, w, bias, y, argmax, drop_p, seed, layer, xspec, wpk, ws, ws_bytes, st);
    });
}

int fft_dgrad(const ConvArgs& a, const float* dz, Ptr2 w, float* dx, const float* x, float inv_keep,
              const uint8_t* prev_argmax, int prev_pool, const float* dzspec, const float* wpk, void* ws, size_t ws_bytes,
              hipStream_t st) {
    return with_cfg(a, 1, [&](auto t) {
        return fft_dgrad_t<typename decltype(t)::type>(a, dz, w, dx, x, inv_keep, prev_argmax, prev_pool, dzspec, wpk, ws,
                                                        ws_bytes, st);
    });
}

// The frequency-weight layout depends on the tile variant (COB, CK, MT, WM), which the dispatcher picks from T: a packed
// operand may be reused by any call whose layout id is the same.
int fft_pack_layout(const ConvArgs& a, int transposed) {
    if (fft_layer_of(a) < 0) return 0;
    int id = 0;
    with_cfg(a, transposed, [&](auto t) {
        typedef typename decltype(t)::type Cfg;
        id = pack_layout_id<Cfg>(Cfg::DILMAX == FDIL128 ? 3 : 1, transposed);   // family 3: 128-point windows (192 slots)
        return ZNS_OK;
    });
    return id;
}
size_t fft_pack_bytes(const ConvArgs& a, int transposed) {
    (void)transposed;
    return fft_layer_of(a) < 0 ? 0 : (size_t)a.nbr * geom(a).FS * a.cout * a.cin * a.kh * sizeof(float);
}
int fft_pack(const ConvArgs& a, Ptr2 w, int transposed, float* dst, hipStream_t st) {
    return with_cfg(a, transposed, [&](auto t) { return pack_fconv<typename decltype(t)::type>(a, w, transposed, dst, st); });
}

int fft_wgrad(const ConvArgs& a, const float* dz, const float* x, MPtr2 dw, MPtr2 db, int accumulate,
              const float* xspec, float* dzspec_out, void* ws, size_t ws_bytes, hipStream_t st) {
    switch (fft_layer_of(a)) {
        case 2: return fft_wgrad_t<FW2>(a, dz, x, dw, db, accumulate, xspec, dzspec_out, ws, ws_bytes, st);
        case 3: return fft_wgrad_t<FW3>(a, dz, x, dw, db, accumulate, xspec, dzspec_out, ws, ws_bytes, st);
        case 4: return fft_wgrad_t<FW4>(a, dz, x, dw, db, accumulate, xspec, dzspec_out, ws, ws_bytes, st);
        case 5: return fft_wgrad_t<FW5>(a, dz, x, dw, db, accumulate, xspec, dzspec_out, ws, ws_bytes, st);
        case 6: return fft_wgrad_t<FW6>(a, dz, x, dw, db, accumulate, xspec, dzspec_out, ws, ws_bytes, st);
        case 7: return fft_wgrad_t<FW7>(a, dz, x, dw, db, accumulate, xspec, dzspec_out, ws, ws_bytes, st);
        case 8: return fft_wgrad_t<FW8>(a, dz, x, dw, db, accumulate, xspec, dzspec_out, ws, ws_bytes, st);
    }
    return ZNS_ERR_ARG;
}

}  // namespace zns
