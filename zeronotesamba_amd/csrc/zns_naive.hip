// Naive (one thread / one wave per output) conv kernels.  NOT the product path:
// they exist so the MFMA kernels can be validated ON DEVICE at full BASELINE sizes,
// where the CPU oracle would take minutes.  Selected with zns_set_impl(ZNS_IMPL_NAIVE).
#include "zns_common.h"
#include "zns_kernels.h"

namespace zns {

__global__ void naive_conv_fwd(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H, int pool,
                               Ptr2 x, Ptr2 w, Ptr2 bias, float* __restrict__ y,
                               uint8_t* __restrict__ argmax, uint32_t thresh, float inv_keep, uint32_t seed,
                               int layer) {
    const int Hp = H / pool;
    const long long per_branch = (long long)B * cout * Hp * T;
    const long long total = per_branch * nbr;
    const int ph = kh / 2, pw = kw / 2;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total;
         o += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(o % T);
        long long q = o / T;
        const int hp = (int)(q % Hp);
        q /= Hp;
        const int co = (int)(q % cout);
        const int n = (int)(q / cout);
        const int br = n / B;
        const float* wb = w.p[br] + (size_t)co * cin * kh * kw;
        float best = -INFINITY;
        int besta = 0;
        for (int a = 0; a < pool; ++a) {
            const int h = hp * pool + a;
            float acc = 0.f;
            for (int ci = 0; ci < cin; ++ci) {
                const float* xp = x.p[br] + ((size_t)(n - br * B) * cin + ci) * H * T;
                const float* wp = wb + (size_t)ci * kh * kw;
                for (int r = 0; r < kh; ++r) {
                    const int hh = h + r - ph;
                    if (hh < 0 || hh >= H) continue;
                    for (int s = 0; s < kw; ++s) {
                        const int tt = t + s - pw;
                        if (tt < 0 || tt >= T) continue;
                        acc = fmaf(xp[(size_t)hh * T + tt], wp[r * kw + s], acc);
                    }
                }
            }
            if (pool_takes(acc, best)) {
                best = acc;
                besta = a;
            }
        }
        float v = best + bias.p[br][co];
        v = relu_nan(v);
        if (thresh) {
            const uint64_t idx = (uint64_t)(o - (long long)br * per_branch);
            v = drop_apply(drop_keep(layer_seed(seed, layer, br), idx, thresh), v, inv_keep);
        }
        y[o] = v;
        if (argmax) argmax[o] = (uint8_t)besta;
    }
}

__global__ void naive_conv_dgrad(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H,
                                 const float* __restrict__ dz, Ptr2 w, float* __restrict__ dx,
                                 const float* __restrict__ x, float inv_keep,
                                 const uint8_t* __restrict__ prev_argmax, int prev_pool) {
    const long long total = (long long)nbr * B * cin * H * T;
    const int ph = kh / 2, pw = kw / 2;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total;
         o += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(o % T);
        long long q = o / T;
        const int h = (int)(q % H);
        q /= H;
        const int ci = (int)(q % cin);
        const int n = (int)(q / cin);
        const int br = n / B;
        float acc = 0.f;
        for (int co = 0; co < cout; ++co) {
            const float* zp = dz + ((size_t)n * cout + co) * H * T;
            const float* wp = w.p[br] + ((size_t)co * cin + ci) * kh * kw;
            for (int r = 0; r < kh; ++r) {
                const int hh = h - r + ph;
                if (hh < 0 || hh >= H) continue;
                for (int s = 0; s < kw; ++s) {
                    const int tt = t - s + pw;
                    if (tt < 0 || tt >= T) continue;
                    acc = fmaf(zp[(size_t)hh * T + tt], wp[r * kw + s], acc);
                }
            }
        }
        float g = acc;
        if (x) g = (x[o] != 0.f) ? g * inv_keep : 0.f;
        if (prev_pool > 1) {
            const int am = prev_argmax[o];
            const size_t base = (((size_t)n * cin + ci) * H + h) * prev_pool * T + t;
            for (int a = 0; a < prev_pool; ++a) dx[base + (size_t)a * T] = (a == am) ? g : 0.f;
        } else {
            dx[o] = g;
        }
    }
}

// one wave per weight element
__global__ void naive_conv_wgrad(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H,
                                 const float* __restrict__ dz, Ptr2 x, MPtr2 dw,
                                 int accumulate) {
    const long long per_branch = (long long)cout * cin * kh * kw;
    const long long o = blockIdx.x;
    const int br = (int)(o / per_branch);
    long long q = o % per_branch;
    const int s = (int)(q % kw);
    q /= kw;
    const int r = (int)(q % kh);
    q /= kh;
    const int ci = (int)(q % cin);
    const int co = (int)(q / cin);
    const int ph = kh / 2, pw = kw / 2;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        const int n = br * B + b;
        const float* zp = dz + ((size_t)n * cout + co) * H * T;
        const float* xp = x.p[br] + ((size_t)b * cin + ci) * H * T;
        for (int h = 0; h < H; ++h) {
            const int hh = h + r - ph;
            if (hh < 0 || hh >= H) continue;
            for (int t = threadIdx.x; t < T; t += 64) {
                const int tt = t + s - pw;
                if (tt < 0 || tt >= T) continue;
                acc = fmaf(zp[(size_t)h * T + t], xp[(size_t)hh * T + tt], acc);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (threadIdx.x == 0) {
        float* d = dw.p[br] + (o % per_branch);
        *d = accumulate ? *d + acc : acc;
    }
}

// one block (256 threads) per (branch, channel): db = sum over b,h,t of dz
__global__ void channel_sum_kernel(int B, int C, long long HT, const float* __restrict__ dz, MPtr2 db,
                                   int accumulate) {
    const int co = blockIdx.x % C;
    const int br = blockIdx.x / C;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* zp = dz + ((size_t)(br * B + b) * C + co) * HT;
        for (long long i = threadIdx.x; i < HT; i += blockDim.x) acc += zp[i];
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* d = db.p[br] + co;
        *d = accumulate ? *d + red[0] : red[0];
    }
}

int naive_fwd(const ConvArgs& a, Ptr2 x, Ptr2 w, Ptr2 bias, float* y, uint8_t* argmax, float drop_p,
              uint32_t seed, int layer, hipStream_t st) {
    const long long total = (long long)a.nbr * a.B * a.cout * (a.H / a.pool) * a.T;
    const int blocks = (int)((total + 255) / 256 > 65536 * 4 ? 65536 * 4 : (total + 255) / 256);
    const uint32_t thresh = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
    hipLaunchKernelGGL(naive_conv_fwd, dim3(blocks), dim3(256), 0, st, a.nbr, a.B, a.T, a.cin, a.cout, a.kh, a.kw,
                       a.H, a.pool, x, w, bias, y, argmax, thresh, inv_keep, seed, layer);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int naive_dgrad(const ConvArgs& a, const float* dz, Ptr2 w, float* dx, const float* x, float inv_keep,
                const uint8_t* prev_argmax, int prev_pool, hipStream_t st) {
    const long long total = (long long)a.nbr * a.B * a.cin * a.H * a.T;
    const int blocks = (int)((total + 255) / 256 > 65536 * 4 ? 65536 * 4 : (total + 255) / 256);
    hipLaunchKernelGGL(naive_conv_dgrad, dim3(blocks), dim3(256), 0, st, a.nbr, a.B, a.T, a.cin, a.cout, a.kh,
                       a.kw, a.H, dz, w, dx, x, inv_keep, prev_argmax, prev_pool);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int naive_wgrad(const ConvArgs& a, const float* dz, Ptr2 x, MPtr2 dw, int accumulate, hipStream_t st) {
    const long long total = (long long)a.nbr * a.cout * a.cin * a.kh * a.kw;
    hipLaunchKernelGGL(naive_conv_wgrad, dim3((unsigned)total), dim3(64), 0, st, a.nbr, a.B, a.T, a.cin, a.cout,
                       a.kh, a.kw, a.H, dz, x, dw, accumulate);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int channel_sum(int nbr, int B, int C, long long HT, const float* dz, MPtr2 db, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(channel_sum_kernel, dim3(nbr * C), dim3(256), 0, st, B, C, HT, dz, db, accumulate);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

}  // namespace zns
