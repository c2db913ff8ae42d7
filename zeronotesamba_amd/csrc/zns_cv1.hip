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

// block -> (n, co, h); 128 threads, each owning 4 consecutive time steps per pass (one float4 of dZ, a 3 x 14 input
// window in registers: 10.5 input loads per dZ element instead of 33).  Partial sums [33 taps + bias] per block into ws.
__global__ void __launch_bounds__(128) cv1_wgrad_kernel(int T, int H, const float* __restrict__ dz,
                                                        const float* __restrict__ x, float* __restrict__ part) {
    __shared__ float red[2][CV1_KH * CV1_KW + 1];
    const int h = blockIdx.x % H;
    const int co = (blockIdx.x / H) % CV1_CO;
    const int n = blockIdx.x / (H * CV1_CO);
    const float* zp = dz + (((size_t)n * CV1_CO + co) * H + h) * T;
    float acc[CV1_KH * CV1_KW + 1];
#pragma unroll
    for (int k = 0; k < CV1_KH * CV1_KW + 1; ++k) acc[k] = 0.f;
    const bool vec = (T % 4 == 0);
    for (int t0 = threadIdx.x * 4; t0 < T; t0 += 128 * 4) {
        float g[4];
        if (vec) {
            const float4 v = *reinterpret_cast<const float4*>(zp + t0);
            g[0] = v.x; g[1] = v.y; g[2] = v.z; g[3] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = (t0 + k < T) ? zp[t0 + k] : 0.f;
        }
        acc[CV1_KH * CV1_KW] += (g[0] + g[1]) + (g[2] + g[3]);
#pragma unroll
        for (int r = 0; r < CV1_KH; ++r) {
            const int hh = h + r - 1;
            const bool rok = hh >= 0 && hh < H;
            const float* xp = x + ((size_t)n * H + (rok ? hh : 0)) * T;
            float xw[4 + CV1_KW - 1];
#pragma unroll
            for (int c = 0; c < 4 + CV1_KW - 1; ++c) {
                const int tt = t0 + c - 5;
                xw[c] = (rok && tt >= 0 && tt < T) ? xp[tt] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < CV1_KW; ++s)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[r * CV1_KW + s] = fmaf(g[k], xw[k + s], acc[r * CV1_KW + s]);
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < CV1_KH * CV1_KW + 1; ++k) {
        float v = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wid][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < CV1_KH * CV1_KW + 1) {
        const int k = threadIdx.x;
        part[(size_t)blockIdx.x * (CV1_KH * CV1_KW + 1) + k] = red[0][k] + red[1][k];
    }
}

// wave -> (br, co, k): sums the B*H partials of that tap in a fixed lane/tree order (deterministic)
__global__ void __launch_bounds__(64) cv1_wgrad_reduce(int B, int H, const float* __restrict__ part, MPtr2 dw, MPtr2 db,
                                                       int accumulate, int nbr) {
    constexpr int K1 = CV1_KH * CV1_KW + 1;
    const int o = blockIdx.x;
    const int k = o % K1, co = (o / K1) % CV1_CO, br = o / (K1 * CV1_CO);
    float s = 0.f;
    for (int i = threadIdx.x; i < B * H; i += 64) {
        const int b = i / H, h = i - b * H;
        s += part[((size_t)((br * B + b) * CV1_CO + co) * H + h) * K1 + k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) {
        float* d = (k < K1 - 1) ? dw.p[br] + co * (K1 - 1) + k : db.p[br] + co;
        *d = accumulate ? *d + s : s;
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
    hipLaunchKernelGGL(cv1_wgrad_kernel, dim3(a.nbr * a.B * CV1_CO * a.H), dim3(128), 0, st, a.T, a.H, dz, x, part);
    ZNS_LAUNCH_CHECK();
    const int n = a.nbr * CV1_CO * (CV1_KH * CV1_KW + 1);
    hipLaunchKernelGGL(cv1_wgrad_reduce, dim3(n), dim3(64), 0, st, a.B, a.H, part, dw, db, accumulate,
                       a.nbr);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

}  // namespace zns
