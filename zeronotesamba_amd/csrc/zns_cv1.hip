// cv1: Conv2d(1 -> 64, (3,11), padding (1,5)) + ReLU + Dropout  (models.py:16,37-39) and its weight gradient.
// K = 33 only, so this layer is HBM-bound by SURVEY 8(d) (201 MB of output per branch at B=16, T=512) -- but the vector
// pipe binds it first: 33 FMAs + a 3-integer-multiply dropout hash (quarter rate: worth ~24 more) per output.  The
// forward kernel therefore takes the FMAs to the matrix pipe (a 64 x 34 x pixels GEMM: half their vector-pipe time, and on
// gfx950 the two pipes of a SIMD do not overlap, so it is time saved, not hidden); bias / ReLU / hash stay vector work.
#include "zns_common.h"
#include "zns_kernels.h"

namespace zns {

constexpr int CV1_KH = 3, CV1_KW = 11, CV1_CO = 64;

typedef float c1_f32x16 __attribute__((ext_vector_type(16)));
constexpr int C1_TB = 256;                 // pixels of one image row per block
constexpr int C1_RS = C1_TB + 16;          // LDS row stride (>= C1_TB + 10)
constexpr int C1_STEPS = (CV1_KH * CV1_KW + 1) / 2;     // 17 k-steps of 2 taps (the 34th tap is a zero weight)

// Block = (sample n, row h, 256 pixels): the three input rows are staged once (zero padded); a wave takes 32-pixel
// groups: M = 64 output channels (two 32 x 32 accumulators), K = 34 taps, N = 32 pixels.  A (the weights) lives in
// registers for the whole kernel; B is one LDS read per k-step at a per-lane tap offset computed once.  Epilogue on the
// accumulators: bias, ReLU, dropout hash of the output's linear index, 4-byte buffer stores (32 consecutive pixels of one
// channel per half wave = 128-byte runs; the channel goes into the scalar offset, out-of-row lanes out of range).
__global__ void __launch_bounds__(256) cv1_fwd_kernel(int B, int T, int H, Ptr2 xb, Ptr2 w,
                                                      Ptr2 bias, float* __restrict__ y, uint32_t thresh,
                                                      float inv_keep, uint32_t seed, int layer, int tblocks) {
    __shared__ float sx[CV1_KH * C1_RS];
    __shared__ float sb[CV1_CO];                   // bias: read per output in the epilogue (LDS, not 32 registers)
    int bid = blockIdx.x;
    const int tb = bid % tblocks; bid /= tblocks;
    const int h = bid % H;
    const int n = bid / H;
    const int br = n / B, b = n % B;
    const int t_blk = tb * C1_TB;
    const float* __restrict__ x = xb.p[br] + (size_t)b * H * T;      // this sample's rows (the branches' inputs may be separate tensors)
    for (int i = threadIdx.x; i < CV1_KH * (C1_TB + CV1_KW - 1); i += 256) {
        const int r = i / (C1_TB + CV1_KW - 1), c = i - r * (C1_TB + CV1_KW - 1);
        const int hh = h + r - 1, tt = t_blk + c - CV1_KW / 2;
        sx[r * C1_RS + c] = (hh >= 0 && hh < H && tt >= 0 && tt < T) ? x[(size_t)hh * T + tt] : 0.f;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int half = lane >> 5, px = lane & 31;
    const float* wb = w.p[br];
    float a0[C1_STEPS], a1[C1_STEPS];
    int off[C1_STEPS];
    if (threadIdx.x < CV1_CO) sb[threadIdx.x] = bias.p[br][threadIdx.x];
#pragma unroll
    for (int st = 0; st < C1_STEPS; ++st) {
        const int kk = 2 * st + half;
        const bool real = kk < CV1_KH * CV1_KW;
        a0[st] = real ? wb[px * (CV1_KH * CV1_KW) + kk] : 0.f;
        a1[st] = real ? wb[(px + 32) * (CV1_KH * CV1_KW) + kk] : 0.f;
        const int kq = real ? kk : 0;
        off[st] = (kq / CV1_KW) * C1_RS + kq % CV1_KW + px;
    }
    const float* sbh = sb + 4 * half;
    const uint32_t lseed = layer_seed(seed, layer, br);
    const size_t HT = (size_t)H * T;
    float* yrow = y + ((size_t)n * CV1_CO * H + h) * T;                    // channel 0 of this (sample, row)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        yrow, 0, (int)(((size_t)(CV1_CO - 1) * HT + T) * sizeof(float)), 0x00020000);
    __syncthreads();
    for (int g = wid; g < C1_TB / 32; g += 4) {
        const int t = t_blk + g * 32 + px;
        if (t_blk + g * 32 >= T) break;                                    // uniform: the row ends before this group
        c1_f32x16 acc0, acc1;
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc0[k] = 0.f; acc1[k] = 0.f; }
#pragma unroll
        for (int st = 0; st < C1_STEPS; ++st) {
            const float bval = sx[off[st] + g * 32];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[st], bval, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[st], bval, acc1, 0, 0, 0);
        }
        const unsigned voff = t < T ? (unsigned)(((size_t)(4 * half) * HT + t) * sizeof(float)) : 0x80000000u;
        // dropout hash of the output's linear index idx0 + dco * HT (hash_u32, zns_common.h), with the two multiplies that
        // depend only on idx0 done once per group: (lo0 + c) * M = lo0 * M + c * M (mod 2^32, c * M a scalar), and the high
        // word is hi0 or hi0 + 1.  v_mul_lo_u32 runs at a quarter of the vector rate: four per output were 16 of its ~30
        // issue slots.
        const uint64_t idx0 = (((uint64_t)b * CV1_CO + 4 * half) * H + h) * T + t;
        const uint32_t lo0 = (uint32_t)idx0, hi0 = (uint32_t)(idx0 >> 32);
        const uint32_t lo0m = lo0 * 0x9E3779B1u, him0 = hi0 * 0x7FEB352Du, him1 = (hi0 + 1u) * 0x7FEB352Du;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int dco = tl * 32 + (k & 3) + 8 * (k >> 2);           // channel = dco + 4 * half
                float v = (tl ? acc1[k] : acc0[k]) + sbh[dco];
                v = relu_nan(v);
                if (thresh) {
                    const uint32_t c = (uint32_t)((size_t)dco * HT);             // < 2^32: H * T < 2^26 for any clip
                    const uint32_t lo = lo0 + c;
                    const uint32_t hsh = lowbias32((lo0m + c * 0x9E3779B1u) ^ lseed ^ (lo < lo0 ? him1 : him0));
                    v = drop_apply((hsh >> 8) >= thresh, v, inv_keep);
                }
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, (unsigned)(dco * HT * sizeof(float)), 0);
            }
    }
}

// Weight gradient: dW[co][r][s] = sum_{n,h,t} dZ[n,co,h,t] * x[n,h+r-1,t+s-5],  db[co] = sum dZ.
// block -> (n, h).  Per 128-step time chunk the 64 x 128 dZ tile (row stride 129: conflict-free per-channel reads) and
// the three zero-padded input rows are staged in LDS with coalesced loads (register-prefetched one chunk ahead: see the
// loop).  The multiply runs on the matrix pipe since round 3 (see the kernel; before: 11 tap accumulators per lane on
// three waves, 33 vector FMAs per dZ element, 152 us).  The block writes its 34 x 64 partial sums tap-major (coalesced),
// cv1_wgrad_reduce adds the B*H blocks in a fixed order.
constexpr int CW_TC = 128, CW_XS = CW_TC + 16, CW_K1 = CV1_KH * CV1_KW + 1;

__global__ void __launch_bounds__(256) cv1_wgrad_kernel(int B, int T, int H, const float* __restrict__ dz,
                                                        Ptr2 xb, float* __restrict__ part) {
    // the dZ tile of the K loop and, after it, the four waves' partial sums share one buffer
    __shared__ float sbuf[4 * CW_K1 * CV1_CO > CV1_CO * (CW_TC + 1) ? 4 * CW_K1 * CV1_CO : CV1_CO * (CW_TC + 1)];
    __shared__ __attribute__((aligned(16))) float sx[CV1_KH * CW_XS];
    float* const sdz = sbuf;
    const int h = blockIdx.x % H, n = blockIdx.x / H;
    const float* __restrict__ x = xb.p[n / B] + (size_t)(n % B) * H * T;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, tap = lane & 31;
    // Matrix pipe: D[co][tap] += dZ[co][pixel] * x[tap's row][pixel + tap's column] for taps 0..31 (M = 64 channels in two
    // tiles, N = 32 taps, K = pixels); wave w owns pixels 32 w .. 32 w + 31 of every 128-pixel chunk.  Tap 32 and the bias
    // sum (a tap whose input is 1) stay on the vector pipe, lane = channel: 2 operations per dZ element instead of 34.
    c1_f32x16 accA, accB;
#pragma unroll
    for (int k = 0; k < 16; ++k) { accA[k] = 0.f; accB[k] = 0.f; }
    float acc32 = 0.f, bsum = 0.f;
    const int boff = (tap / CV1_KW) * CW_XS + tap % CV1_KW + half + 32 * wv;      // B operand: sx index of pixel 32 w + half
    const int aoff = tap * (CW_TC + 1) + half + 32 * wv;                            // A operand: channel = lane & 31 (+ 32)
    // The dZ tile of chunk c + 1 (32 floats per thread) and the three input rows travel global -> registers while chunk c is
    // being multiplied out of LDS, and registers -> LDS between the two barriers: the block no longer alternates between
    // "all waves wait for memory" and "memory idle" (round 2: 2 TB/s of dZ reads).
    constexpr int NZ = CV1_CO * CW_TC / 256, NX = (CV1_KH * CW_XS + 255) / 256;
    float vz[NZ], vx[NX];
    const float* __restrict__ dzn = dz + ((size_t)n * CV1_CO * H + h) * T;
    auto fetch = [&](int t0) {
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            const int i = threadIdx.x + j * 256;
            const int co = i / CW_TC, tt = i % CW_TC;
            vz[j] = (t0 + tt < T) ? dzn[(size_t)co * H * T + t0 + tt] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = threadIdx.x + j * 256;
            const int r = i / CW_XS, c = i % CW_XS;
            const int hh = h + r - 1, tt = t0 + c - 5;
            vx[j] = (i < CV1_KH * CW_XS && hh >= 0 && hh < H && tt >= 0 && tt < T) ? x[(size_t)hh * T + tt] : 0.f;
        }
    };
    fetch(0);
    for (int t0 = 0; t0 < T; t0 += CW_TC) {
        __syncthreads();                      // the previous chunk's LDS reads are done
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            const int i = threadIdx.x + j * 256;
            sdz[(i / CW_TC) * (CW_TC + 1) + (i % CW_TC)] = vz[j];
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int i = threadIdx.x + j * 256;
            if (i < CV1_KH * CW_XS) sx[i] = vx[j];
        }
        __syncthreads();
        if (t0 + CW_TC < T) fetch(t0 + CW_TC);   // in flight during the multiply below
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const float bq = sx[boff + 2 * ks];
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(sdz[aoff + 2 * ks], bq, accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(sdz[aoff + 32 * (CW_TC + 1) + 2 * ks], bq, accB, 0, 0, 0);
        }
        const float* zr = sdz + lane * (CW_TC + 1) + 32 * wv;
        const float* x32 = sx + (CV1_KH - 1) * CW_XS + (CV1_KW - 1) + 32 * wv;     // tap 32 = (row 2, column 10)
#pragma unroll 8
        for (int tt = 0; tt < 32; ++tt) {
            const float z = zr[tt];
            acc32 = fmaf(z, x32[tt], acc32);
            bsum += z;
        }
    }
    // the four waves' partial sums -> LDS (tap-major, as `part`), then every thread adds four values and writes them
    __syncthreads();                              // the last chunk's operand reads are done: sbuf is free
    float* mine = sbuf + wv * (CW_K1 * CV1_CO);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int co = (k & 3) + 8 * (k >> 2) + 4 * half;
        mine[tap * CV1_CO + co] = accA[k];
        mine[tap * CV1_CO + 32 + co] = accB[k];
    }
    mine[32 * CV1_CO + lane] = acc32;
    mine[(CW_K1 - 1) * CV1_CO + lane] = bsum;
    __syncthreads();
    float* pb = part + (size_t)blockIdx.x * (CW_K1 * CV1_CO);
    for (int i = threadIdx.x; i < CW_K1 * CV1_CO; i += 256)
        pb[i] = ((sbuf[i] + sbuf[CW_K1 * CV1_CO + i]) + sbuf[2 * CW_K1 * CV1_CO + i]) + sbuf[3 * CW_K1 * CV1_CO + i];
}

// part [nbr*B*H][34][64] -> dW, db.  Block: 64 consecutive columns x 16 row phases; every column is summed in a fixed
// order (phase-strided rows, then the 16 phases), so the result does not depend on scheduling.
__global__ void __launch_bounds__(1024) cv1_wgrad_reduce(int B, int H, const float* __restrict__ part, MPtr2 dw, MPtr2 db,
                                                         int accumulate, int nbr) {
    __shared__ float red[16][64];
    constexpr int NC = CW_K1 * CV1_CO;
    const int br = blockIdx.x / (NC / 64);
    const int j = (blockIdx.x % (NC / 64)) * 64 + (threadIdx.x & 63);
    const int ph = threadIdx.x >> 6;
    const float* p = part + (size_t)br * B * H * NC + j;
    float s = 0.f;
#pragma unroll 8
    for (int i = ph; i < B * H; i += 16) s += p[(size_t)i * NC];
    red[ph][threadIdx.x & 63] = s;
    __syncthreads();
    if (ph == 0) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += red[q][threadIdx.x];
        const int k = j / CV1_CO, co = j % CV1_CO;
        float* d = (k < CW_K1 - 1) ? dw.p[br] + co * (CW_K1 - 1) + k : db.p[br] + co;
        *d = accumulate ? *d + tot : tot;
    }
}

size_t cv1_workspace_bytes(const ConvArgs& a) {
    return (size_t)a.nbr * a.B * CV1_CO * a.H * (CV1_KH * CV1_KW + 1) * sizeof(float);
}

int cv1_fwd(const ConvArgs& a, Ptr2 x, Ptr2 w, Ptr2 bias, float* y, float drop_p, uint32_t seed, int layer,
            hipStream_t st) {
    // the kernel's 32-bit channel offsets (stores and the dropout index), and its masked lanes: a lane past the row's end
    // carries voffset 0x80000000 and relies on the buffer's range check (voffset + instruction offset >= num_records; the
    // scalar offset is not part of it), so num_records -- the 63 channel planes plus one row -- must stay below 2 GiB
    if (((long long)(CV1_CO - 1) * a.H * a.T + a.T) * (long long)sizeof(float) >= (1ll << 31)) {
        set_error("cv1_fwd: H * T = %lld is beyond the kernel's 32-bit channel offsets", (long long)a.H * a.T);
        return ZNS_ERR_ARG;
    }
    const int tblocks = (a.T + C1_TB - 1) / C1_TB;
    const long long blocks = (long long)a.nbr * a.B * a.H * tblocks;
    const uint32_t thresh = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
    hipLaunchKernelGGL(cv1_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a.B, a.T, a.H, x, w, bias, y, thresh,
                       inv_keep, seed, layer, tblocks);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int cv1_wgrad(const ConvArgs& a, const float* dz, Ptr2 x, MPtr2 dw, MPtr2 db, int accumulate, void* ws,
              size_t ws_bytes, hipStream_t st) {
    if (ws_bytes < cv1_workspace_bytes(a)) {
        set_error("cv1_wgrad: workspace %zu < %zu", ws_bytes, cv1_workspace_bytes(a));
        return ZNS_ERR_WORKSPACE;
    }
    float* part = (float*)ws;
    hipLaunchKernelGGL(cv1_wgrad_kernel, dim3(a.nbr * a.B * a.H), dim3(256), 0, st, a.B, a.T, a.H, dz, x, part);
    ZNS_LAUNCH_CHECK();
    hipLaunchKernelGGL(cv1_wgrad_reduce, dim3(a.nbr * (CW_K1 * CV1_CO / 64)), dim3(1024), 0, st, a.B, a.H, part, dw, db,
                       accumulate, a.nbr);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

}  // namespace zns
