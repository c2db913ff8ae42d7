// cv1: Conv2d(1 -> 64, (3,11), padding (1,5)) + ReLU + Dropout  (models.py:16,37-39) and its weight gradient.
// K = 33 only, so this layer is HBM-bound (201 MB of output per branch at B=16, T=512): direct convolution,
// 16-byte stores along T, weights read through the scalar cache.  No MFMA on purpose.
#include "zns_common.h"
#include "zns_kernels.h"

namespace zns {

constexpr int CV1_KH = 3, CV1_KW = 11, CV1_CO = 64;

// thread -> (n, h, 4 consecutive t); loops over the 64 output channels
__global__ void __launch_bounds__(256) cv1_fwd_kernel(int B, int T, int H, const float* __restrict__ x, Ptr2 w,
                                                      Ptr2 bias, float* __restrict__ y, uint32_t thresh,
                                                      float inv_keep, uint32_t seed, int layer, int T4,
                                                      long long total) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const int t0 = (int)(o % T4) * 4;
    const int h = (int)((o / T4) % H);
    const int n = (int)(o / ((long long)T4 * H));
    const int br = n / B, b = n % B;
    float xin[CV1_KH][4 + CV1_KW - 1];
#pragma unroll
    for (int r = 0; r < CV1_KH; ++r) {
        const int hh = h + r - 1;
#pragma unroll
        for (int c = 0; c < 4 + CV1_KW - 1; ++c) {
            const int tt = t0 + c - 5;
            xin[r][c] = (hh >= 0 && hh < H && tt >= 0 && tt < T) ? x[((size_t)n * H + hh) * T + tt] : 0.f;
        }
    }
    const float* wb = w.p[br];
    const float* bb = bias.p[br];
    const uint32_t lseed = layer_seed(seed, layer, br);
    const bool vec = (T % 4 == 0);
    for (int co = 0; co < CV1_CO; ++co) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < CV1_KH; ++r)
#pragma unroll
            for (int s = 0; s < CV1_KW; ++s) {
                const float wv = wb[co * (CV1_KH * CV1_KW) + r * CV1_KW + s];
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = fmaf(xin[r][k + s], wv, acc[k]);
            }
        const float bv = bb[co];
        const size_t base = (((size_t)n * CV1_CO + co) * H + h) * T + t0;
        const uint64_t idx0 = (((uint64_t)b * CV1_CO + co) * H + h) * T + t0;
        float out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = acc[k] + bv;
            v = v > 0.f ? v : 0.f;
            if (thresh) v = drop_keep(lseed, idx0 + k, thresh) ? v * inv_keep : 0.f;
            out[k] = v;
        }
        if (vec) {
            *reinterpret_cast<float4*>(y + base) = make_float4(out[0], out[1], out[2], out[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t0 + k < T) y[base + k] = out[k];
        }
    }
}

// Weight gradient: dW[co][r][s] = sum_{n,h,t} dZ[n,co,h,t] * x[n,h+r-1,t+s-5],  db[co] = sum dZ.
// block -> (n, h); 4 waves: wave r < 3 owns kernel row r, lane = output channel, 11 tap accumulators per lane;
// wave 3 accumulates the bias gradient.  Per 128-step time chunk the 64 x 128 dZ tile (row stride 129: conflict-free
// per-channel reads) and the three zero-padded input rows are staged in LDS with coalesced loads (register-prefetched one
// chunk ahead: see the loop); the input window
// of a step is wave-uniform, i.e. LDS broadcast reads.  No cross-lane reduction at all: the block writes its
// 34 x 64 partial sums tap-major (coalesced), cv1_wgrad_reduce adds the B*H blocks in a fixed order.
constexpr int CW_TC = 128, CW_XS = CW_TC + 16, CW_K1 = CV1_KH * CV1_KW + 1;

__global__ void __launch_bounds__(256) cv1_wgrad_kernel(int T, int H, const float* __restrict__ dz,
                                                        const float* __restrict__ x, float* __restrict__ part) {
    __shared__ float sdz[CV1_CO * (CW_TC + 1)];
    __shared__ __attribute__((aligned(16))) float sx[CV1_KH * CW_XS];
    const int h = blockIdx.x % H, n = blockIdx.x / H;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float acc[CV1_KW];
#pragma unroll
    for (int s = 0; s < CV1_KW; ++s) acc[s] = 0.f;
    float bsum = 0.f;
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
            vx[j] = (i < CV1_KH * CW_XS && hh >= 0 && hh < H && tt >= 0 && tt < T) ? x[((size_t)n * H + hh) * T + tt] : 0.f;
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
        const float* zr = sdz + lane * (CW_TC + 1);
        if (wv < CV1_KH) {
            const float4* xr = reinterpret_cast<const float4*>(sx + wv * CW_XS);
#pragma unroll 2
            for (int tt = 0; tt < CW_TC; tt += 4) {
                float xw[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 f = xr[tt / 4 + q];            // x[t0 + tt + 4q - 5 ...], same address in every lane
                    xw[4 * q] = f.x; xw[4 * q + 1] = f.y; xw[4 * q + 2] = f.z; xw[4 * q + 3] = f.w;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float z = zr[tt + k];
#pragma unroll
                    for (int s = 0; s < CV1_KW; ++s) acc[s] = fmaf(z, xw[k + s], acc[s]);
                }
            }
        } else {
#pragma unroll 8
            for (int tt = 0; tt < CW_TC; ++tt) bsum += zr[tt];
        }
    }
    float* pb = part + (size_t)blockIdx.x * (CW_K1 * CV1_CO);
    if (wv < CV1_KH) {
#pragma unroll
        for (int s = 0; s < CV1_KW; ++s) pb[(wv * CV1_KW + s) * CV1_CO + lane] = acc[s];
    } else {
        pb[(CW_K1 - 1) * CV1_CO + lane] = bsum;
    }
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

int cv1_fwd(const ConvArgs& a, const float* x, Ptr2 w, Ptr2 bias, float* y, float drop_p, uint32_t seed, int layer,
            hipStream_t st) {
    const int T4 = (a.T + 3) / 4;
    const long long total = (long long)a.nbr * a.B * a.H * T4;
    const uint32_t thresh = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
    hipLaunchKernelGGL(cv1_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.B, a.T, a.H, x, w,
                       bias, y, thresh, inv_keep, seed, layer, T4, total);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int cv1_wgrad(const ConvArgs& a, const float* dz, const float* x, MPtr2 dw, MPtr2 db, int accumulate, void* ws,
              size_t ws_bytes, hipStream_t st) {
    if (ws_bytes < cv1_workspace_bytes(a)) {
        set_error("cv1_wgrad: workspace %zu < %zu", ws_bytes, cv1_workspace_bytes(a));
        return ZNS_ERR_WORKSPACE;
    }
    float* part = (float*)ws;
    hipLaunchKernelGGL(cv1_wgrad_kernel, dim3(a.nbr * a.B * a.H), dim3(256), 0, st, a.T, a.H, dz, x, part);
    ZNS_LAUNCH_CHECK();
    hipLaunchKernelGGL(cv1_wgrad_reduce, dim3(a.nbr * (CW_K1 * CV1_CO / 64)), dim3(1024), 0, st, a.B, a.H, part, dw, db,
                       accumulate, a.nbr);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

}  // namespace zns
