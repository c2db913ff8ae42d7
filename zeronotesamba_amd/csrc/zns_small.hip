// Small fused kernels around the conv stack: DS_CNN head (fc1 + sigmoid), Down_CNN combine,
// NT-Xent forward+backward, BCE, fused Adam, on-device shift batching.
// All are HBM- or latency-bound; the design points are coalesced 16-B accesses along T and
// wavefront (64-lane) shuffle reductions.
#include "zns_common.h"
#include "zns_kernels.h"

namespace zns {

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block-wide sum for blockDim.x <= 1024; red must hold 16 floats; result broadcast to all threads
__device__ inline float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}

// ------------------------------------------------------------------ head (models.py:99-101)
// Block = 64 consecutive (n, t) positions x 4 channel quarters: a thread sums a quarter of the channels (independent
// loads, 8 in flight), the quarters are added in a fixed order.  (One thread per position walking all C channels was 34 us
// of load latency on 64 blocks.)
__global__ void __launch_bounds__(256) head_fwd_kernel(int B, int T, int C, const float* __restrict__ h, Ptr2 w, Ptr2 b,
                                                       float* __restrict__ emb, int total) {
    __shared__ float part[4][64];
    const int tl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + tl;
    float acc = 0.f;
    int br = 0;
    if (o < total) {
        const int t = o % T, n = o / T;
        br = n / B;
        const float* hp = h + (size_t)n * C * T + t;
        const float* wp = w.p[br];
        const int cpg = (C + 3) / 4, c0 = g * cpg, c1 = c0 + cpg < C ? c0 + cpg : C;
#pragma unroll 8
        for (int c = c0; c < c1; ++c) acc = fmaf(hp[(size_t)c * T], wp[c], acc);
    }
    part[g][tl] = acc;
    __syncthreads();
    if (g == 0 && o < total) {
        const float s = ((part[0][tl] + part[1][tl]) + part[2][tl]) + part[3][tl] + b.p[br][0];
        emb[o] = 1.0f / (1.0f + expf(-s));
    }
}

// one block per (n, c): writes dz row, partial dw; c==0 block also writes the partial db
__global__ void head_bwd_kernel(int B, int T, int C, const float* __restrict__ h, const float* __restrict__ emb,
                                Ptr2 dembb, Ptr2 w, float inv_keep, float* __restrict__ dz,
                                float* __restrict__ part_w, float* __restrict__ part_b) {
    __shared__ float red[16];
    const int c = blockIdx.x % C, n = blockIdx.x / C, br = n / B;
    const float wc = w.p[br][c];
    const float* __restrict__ demb = dembb.p[br] + (size_t)(n - br * B) * T;     // (the branches' gradients may be separate tensors)
    const float* hp = h + ((size_t)n * C + c) * T;
    float* zp = dz + ((size_t)n * C + c) * T;
    float sw = 0.f, sb = 0.f;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const float e = emb[(size_t)n * T + t];
        const float dpre = demb[t] * e * (1.0f - e);
        const float hv = hp[t];
        sw = fmaf(dpre, hv, sw);
        sb += dpre;
        zp[t] = (hv != 0.f) ? dpre * wc * inv_keep : 0.f;
    }
    sw = block_sum(sw, red);
    if (threadIdx.x == 0) part_w[(size_t)n * C + c] = sw;
    if (c == 0) {
        sb = block_sum(sb, red);
        if (threadIdx.x == 0) part_b[n] = sb;
    }
}

__global__ void head_bwd_reduce(int B, int C, const float* __restrict__ part_w, const float* __restrict__ part_b,
                                MPtr2 dw, MPtr2 db, int accumulate, int nbr) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nbr * (C + 1)) return;
    const int br = o / (C + 1), c = o % (C + 1);
    float s = 0.f;
    if (c < C) {
        for (int b = 0; b < B; ++b) s += part_w[(size_t)(br * B + b) * C + c];
        float* d = dw.p[br] + c;
        *d = accumulate ? *d + s : s;
    } else {
        for (int b = 0; b < B; ++b) s += part_b[br * B + b];
        float* d = db.p[br];
        *d = accumulate ? *d + s : s;
    }
}

// ------------------------------------------------------------------ Down_CNN combine (models.py:144-148)
__global__ void down_fwd_kernel(const float* __restrict__ a, const float* __restrict__ p, size_t n, int reduction,
                                float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = a[i], y = p[i];
        // torch.max(a, b) hands a NaN of either side on (fmaxf would drop it)
        out[i] = reduction == ZNS_REDUCE_MEAN ? (x + y) / 2.0f : (x != x ? x : (y != y ? y : fmaxf(x, y)));
    }
}
__global__ void down_bwd_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                const float* __restrict__ dout, size_t n, int reduction, float* __restrict__ da,
                                float* __restrict__ dp) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float g = dout[i];
        if (reduction == ZNS_REDUCE_MEAN) {
            da[i] = g * 0.5f;
            dp[i] = g * 0.5f;
        } else {  // torch.maximum backward: ties split the gradient evenly
            const float x = a[i], y = p[i];
            da[i] = x > y ? g : (x == y ? g * 0.5f : 0.f);
            dp[i] = y > x ? g : (x == y ? g * 0.5f : 0.f);
        }
    }
}

// ------------------------------------------------------------------ NT-Xent (loss_functions.py:24-55)
// Any number of rows R (the reference's batch_size comes from config.yaml), three short launches over a caller
// workspace  S[R][R] | na2[R] | np2[R] | Ga[R][R] | Gp[R][R] | s1[R] | s2[R]  (float64, see below):
//   1. ntx_dots:    one wave per dot product: S[i][j] = a_i . p_j, na2[i] = a_i . a_i, np2[j] = p_j . p_j
//   2. ntx_softmax: one thread per anchor row: C = S / (|a_i| |p_j|) (norms clamped at eps = 1e-8 as
//                   nn.CosineSimilarity), L_i = logsumexp_j(C_ij / tau) - C_ii / tau, the three statistics (summed in
//                   float64 like the reference's Python floats), dL/dC = G, and the two pre-scaled gradient operators
//                   Ga[i][j] = G_ij / |p_j|,  Gp[j][i] = G_ij / |a_i|,  s1_i = sum_j G_ij C_ij,  s2_j = sum_i G_ij C_ij
//   3. ntx_grad:    ga_i = (sum_j Ga[i][j] p_j  -  a_i s1_i / |a_i|) / |a_i|      (d cos / d a through the normalisation;
//                   gp_j = (sum_i Gp[j][i] a_i  -  p_j s2_j / |p_j|) / |p_j|       gah_i . ah_i = sum_j G_ij C_ij exactly)
// The 16 x 16 x 512 case of the benchmark took 0.14 ms as one serial 1024-thread block; it is now ~3 launch latencies.
__device__ inline double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// The R x R tables (dot products, cosines, softmax weights, the gradient operators) are FLOAT64, and so is the final
// combination of the gradient kernel; only the result is rounded to float32.  Why: with near-collinear embeddings (a freshly
// initialised model: every cosine 0.99, loss = ln(batch) to four digits) the gradient is what is left after the common
// direction cancels, and every rounded SCALAR of the chain -- a cosine, a softmax weight, s1 -- multiplies a whole embedding
// vector: its rounding error is coherent along the embedding, and fc1's weight gradient (a sum over frames against
// non-negative activations, condition number ~300 here) amplifies exactly that component.  Measured at B = 16, D = 512:
// one float32 ulp (6e-8) in s1 moves fc1.weight's gradient by 5e-5 of its norm; the float32 tables put the full-size step's
// worst gradient at 6e-5 .. 3.7e-4 of its norm from the CPU oracle depending on how the roundings fell (torch's autograd
// rounds per ELEMENT, incoherently: 7e-6).  The tables are 3 R^2 + 4 R numbers; float64 costs nothing measurable.
__global__ void __launch_bounds__(256) ntx_dots_kernel(const float* __restrict__ a, const float* __restrict__ p, int R,
                                                       int D, double* __restrict__ S, double* __restrict__ na2,
                                                       double* __restrict__ np2) {
    const int lane = threadIdx.x & 63;
    const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nq = (long long)R * R + 2 * R;
    if (q >= nq) return;
    const float *u, *v;
    double* dst;
    if (q < (long long)R * R) {
        u = a + (size_t)(q / R) * D;
        v = p + (size_t)(q % R) * D;
        dst = S + q;
    } else if (q < (long long)R * R + R) {
        u = v = a + (size_t)(q - (long long)R * R) * D;
        dst = na2 + (q - (long long)R * R);
    } else {
        u = v = p + (size_t)(q - (long long)R * R - R) * D;
        dst = np2 + (q - (long long)R * R - R);
    }
    double s = 0.0;
    for (int d = lane; d < D; d += 64) s = fma((double)u[d], (double)v[d], s);
    s = wave_sum_f64(s);
    if (lane == 0) *dst = s;
}

__global__ void __launch_bounds__(1024) ntx_softmax_kernel(int R, int batch_len, float tau_f, float grad_scale,
                                                           const double* __restrict__ S, const double* __restrict__ na2,
                                                           const double* __restrict__ np2, double* __restrict__ Ga,
                                                           double* __restrict__ Gp, double* __restrict__ s1,
                                                           double* __restrict__ s2, float* __restrict__ out3,
                                                           float* __restrict__ loss_out, int want_grad) {
    __shared__ double red[3][16];
    const double eps = 1e-8, tau = (double)tau_f;
    double l = 0, ps = 0, ns = 0;
    for (int i = threadIdx.x; i < R; i += 1024) {
        const double nai = fmax(sqrt(na2[i]), eps);
        const double* Si = S + (size_t)i * R;
        double m = -INFINITY;
        for (int j = 0; j < R; ++j) m = fmax(m, Si[j] / (nai * fmax(sqrt(np2[j]), eps)) / tau);
        double z = 0.0, rs = 0.0;
        for (int j = 0; j < R; ++j) {
            const double c = Si[j] / (nai * fmax(sqrt(np2[j]), eps));
            z += exp(c / tau - m);
            rs += c;
        }
        const double cii = Si[i] / (nai * fmax(sqrt(np2[i]), eps));
        l += (m + log(z)) - cii / tau;
        ps += cii;
        ns += (rs - cii) / (double)(batch_len - 1);
        if (want_grad) {
            double acc = 0.0;
            for (int j = 0; j < R; ++j) {
                const double npj = fmax(sqrt(np2[j]), eps);
                const double c = Si[j] / (nai * npj);
                const double g = (exp(c / tau - m) / z - (i == j ? 1.0 : 0.0)) * (double)grad_scale / (tau * (double)batch_len);
                Ga[(size_t)i * R + j] = g / npj;
                Gp[(size_t)j * R + i] = g / nai;
                acc = fma(g, c, acc);
            }
            s1[i] = acc;
        }
    }
    // statistics: block sum in float64 (the reference accumulates Python floats)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double v[3] = {l, ps, ns};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
        if (lane == 0) red[k][wid] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; ++k) {
            double t = 0;
            for (int w = 0; w < 16; ++w) t += red[k][w];
            out3[k] = (float)(t / batch_len);
            if (k == 0 && loss_out) *loss_out = (float)(t / batch_len);
        }
    }
    if (!want_grad) return;
    // s2_j = sum_i G_ij C_ij = sum_i Gp[j][i] |a_i| * C_ij : one thread per column, after every row has been written
    __syncthreads();
    for (int j = threadIdx.x; j < R; j += 1024) {
        const double npj = fmax(sqrt(np2[j]), eps);
        double acc = 0.0;
        for (int i = 0; i < R; ++i) {
            const double nai = fmax(sqrt(na2[i]), eps);
            acc = fma(Gp[(size_t)j * R + i] * nai, S[(size_t)i * R + j] / (nai * npj), acc);
        }
        s2[j] = acc;
    }
}

// grid (ceil(D / 256), R, 2): side 0 -> ga row i, side 1 -> gp row i
__global__ void __launch_bounds__(256) ntx_grad_kernel(const float* __restrict__ a, const float* __restrict__ p, int R,
                                                       int D, const double* __restrict__ na2, const double* __restrict__ np2,
                                                       const double* __restrict__ Ga, const double* __restrict__ Gp,
                                                       const double* __restrict__ s1, const double* __restrict__ s2,
                                                       float* __restrict__ ga, float* __restrict__ gp) {
    const int d = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y, side = blockIdx.z;
    float* out = side ? gp : ga;
    if (d >= D || out == nullptr) return;
    const double* G = (side ? Gp : Ga) + (size_t)i * R;      // wave-uniform row of the operator
    const float* other = side ? a : p;
    const float* self = side ? p : a;
    const double nrm = fmax(sqrt((side ? np2 : na2)[i]), 1e-8);
    double acc = 0.0;
    for (int j = 0; j < R; ++j) acc = fma(G[j], (double)other[(size_t)j * D + d], acc);
    const double sdot = (side ? s2 : s1)[i];
    out[(size_t)i * D + d] = (float)((acc - ((double)self[(size_t)i * D + d] / nrm) * sdot) / nrm);
}

// y = x * scale[0], scale a device scalar: the chain-rule factor autograd hands a loss node's backward (dloss), applied
// without reading it back to the host and without an ATen launch; up to two arrays per launch (NT-Xent: ga and gp)
__global__ void scale_kernel(const float* __restrict__ scale, const float* __restrict__ x0, float* __restrict__ y0,
                             size_t n0, const float* __restrict__ x1, float* __restrict__ y1, size_t n1) {
    const float s = scale[0];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1; i += stride) {
        if (i < n0) y0[i] = x0[i] * s;
        else y1[i - n0] = x1[i - n0] * s;
    }
}

// ------------------------------------------------------------------ BCELoss (loader.py:16)
__global__ void __launch_bounds__(1024) bce_kernel(const float* __restrict__ o, const float* __restrict__ t,
                                                   size_t n, float* __restrict__ loss, float* __restrict__ dout,
                                                   float grad_scale) {
    __shared__ float red[16];
    float s = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 1024) {
        const float x = o[i], y = t[i];
        // ATen clamps the logs at -100; a clamp keeps a NaN (fmaxf would turn a NaN output into a finite loss)
        float l1 = logf(x), l0 = logf(1.f - x);
        l1 = l1 < -100.f ? -100.f : l1;
        l0 = l0 < -100.f ? -100.f : l0;
        s -= y * l1 + (1.f - y) * l0;
        if (dout) dout[i] = grad_scale * (x - y) / fmaxf((1.f - x) * x, 1e-12f) / (float)n;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) *loss = s / (float)n;
}

// ------------------------------------------------------------------ Adam (torch.optim.Adam defaults)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float b1, float b2, float eps, float step_size,
                            float sqrt_bc2, float grad_scale, int aligned) {
    const size_t n4 = aligned ? n / 4 : 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pf = &pp.x; const float* gf = &gg.x; float* mf = &mm.x; float* vf = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gf[k] * grad_scale;
            mf[k] = mf[k] + (gk - mf[k]) * (1.f - b1);          // exp_avg.lerp_(grad, 1-beta1)
            vf[k] = vf[k] * b2 + (1.f - b2) * gk * gk;          // mul_(beta2).addcmul_(g, g, 1-beta2)
            const float denom = sqrtf(vf[k]) / sqrt_bc2 + eps;
            pf[k] = pf[k] - step_size * (mf[k] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gk = g[i] * grad_scale;
        const float mk = m[i] + (gk - m[i]) * (1.f - b1);
        const float vk = v[i] * b2 + (1.f - b2) * gk * gk;
        m[i] = mk;
        v[i] = vk;
        p[i] = p[i] - step_size * (mk / (sqrtf(vk) / sqrt_bc2 + eps));
    }
}

// Multi-tensor form: one launch for a list of parameter tensors (the module path hands torch-style per-parameter
// state to the optimizer: 36 tensors of 1 .. 6.9 M elements).  blockIdx.y = tensor, blockIdx.x strides over it.
struct AdamSlot {
    float* p;
    const float* g;
    float* m;
    float* v;
    unsigned long long n;
};
__global__ void adam_multi_kernel(const AdamSlot* __restrict__ tab, float b1, float b2, float eps, float step_size,
                                  float sqrt_bc2, float grad_scale) {
    const AdamSlot s = tab[blockIdx.y];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const bool aligned = (((uintptr_t)s.p | (uintptr_t)s.g | (uintptr_t)s.m | (uintptr_t)s.v) & 15) == 0;
    const size_t n4 = aligned ? s.n / 4 : 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(s.p)[i];
        const float4 gg = reinterpret_cast<const float4*>(s.g)[i];
        float4 mm = reinterpret_cast<float4*>(s.m)[i];
        float4 vv = reinterpret_cast<float4*>(s.v)[i];
        float* pf = &pp.x; const float* gf = &gg.x; float* mf = &mm.x; float* vf = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gf[k] * grad_scale;
            mf[k] = mf[k] + (gk - mf[k]) * (1.f - b1);
            vf[k] = vf[k] * b2 + (1.f - b2) * gk * gk;
            const float denom = sqrtf(vf[k]) / sqrt_bc2 + eps;
            pf[k] = pf[k] - step_size * (mf[k] / denom);
        }
        reinterpret_cast<float4*>(s.p)[i] = pp;
        reinterpret_cast<float4*>(s.m)[i] = mm;
        reinterpret_cast<float4*>(s.v)[i] = vv;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < s.n; i += stride) {
        const float gk = s.g[i] * grad_scale;
        const float mk = s.m[i] + (gk - s.m[i]) * (1.f - b1);
        const float vk = s.v[i] * b2 + (1.f - b2) * gk * gk;
        s.m[i] = mk;
        s.v[i] = vk;
        s.p[i] = s.p[i] - step_size * (mk / (sqrtf(vk) / sqrt_bc2 + eps));
    }
}

// ------------------------------------------------------------------ shift batching (pretext.py:308-318)
__global__ void shift_gather_kernel(const float* __restrict__ bank, int bank_T, const int32_t* __restrict__ clip,
                                    const int32_t* __restrict__ start, int n_items, int T,
                                    float* __restrict__ out) {
    // grid: (rows = 2*n_items*96), threads over T
    const int row = blockIdx.x;
    const int f = row % 96;
    const int i = (row / 96) % n_items;
    const int br = row / (96 * n_items);
    const float* src = bank + (((size_t)clip[i] * 2 + br) * 96 + f) * bank_T + start[i];
    float* dst = out + (size_t)row * T;
    for (int t = threadIdx.x; t < T; t += blockDim.x) dst[t] = src[t];
}


// ------------------------------------------------------------------ RMS stem gate (processing/stem_check.py:8-51)
constexpr int RMS_FRAME = 2048, RMS_HOP = 512;

struct StemPtrs {
    const float* p[8];
};

// anchor[:] = first non-drum stem; anchor[:] += every further one, in dict order (pretext.py:50-61): float32, that order
__global__ void stem_mix_kernel(StemPtrs st, int n_stems, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = st.p[0][i];
    for (int k = 1; k < n_stems; ++k) acc += st.p[k][i];
    out[i] = acc;
}

// one wave per (frame, candidate): librosa.feature.rms(frame_length=2048, hop_length=512, center=True, reflect pad)
// of both signals, then the two ratio tests of check_CL_clips.  Squares are float32 like np.abs(x)**2; the 2048-term
// sum is carried in float64 and rounded once (numpy's float32 running sum differs in the last bits only).
// Many files per launch (bank fill, pretext.py:143-170): row f of the device table is
//   [n_other, start, drums pointer, other-stem pointers ...] (int64);
// out[f][0][i] = stems_other[0][start + i] + stems_other[1][start + i] + ... (float32 adds in that order, the arithmetic
// of stem_mix_kernel), out[f][1][i] = drums[start + i], i < n: the two crops the RMS gate and the XQT read.
__global__ void __launch_bounds__(256) stage_crops_kernel(const long long* __restrict__ table, int row_len, int n,
                                                          float* __restrict__ out) {
    const long long* __restrict__ row = table + (size_t)blockIdx.y * row_len;
    const int n_other = (int)row[0];
    const long long start = row[1];
    const float* __restrict__ drums = reinterpret_cast<const float*>(row[2]) + start;
    float* __restrict__ oa = out + (size_t)blockIdx.y * 2 * n;
    float* __restrict__ op = oa + n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float acc = reinterpret_cast<const float*>(row[3])[start + i];
        for (int k = 1; k < n_other; ++k) acc += reinterpret_cast<const float*>(row[3 + k])[start + i];
        oa[i] = acc;
        op[i] = drums[i];
    }
}

__global__ void __launch_bounds__(64) rms_gate_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                                      const long long* __restrict__ start, int n_samples,
                                                      int n_frames, float* __restrict__ rms_a,
                                                      float* __restrict__ rms_p, int* __restrict__ pass_count) {
    const int f = blockIdx.x, c = blockIdx.y;
    const long long s0 = start ? start[c] : (long long)c * n_samples;
    const float* xa = a + s0;
    const float* xp = p + s0;
    double sa = 0.0, sp = 0.0;
    for (int j = threadIdx.x; j < RMS_FRAME; j += 64) {
        int idx = f * RMS_HOP + j - RMS_FRAME / 2;
        if (idx < 0) idx = -idx;
        if (idx >= n_samples) idx = 2 * (n_samples - 1) - idx;
        const float va = xa[idx], vp = xp[idx];
        sa += (double)(va * va);
        sp += (double)(vp * vp);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sa += __shfl_xor(sa, off, 64);
        sp += __shfl_xor(sp, off, 64);
    }
    if (threadIdx.x == 0) {
        const float ra = sqrtf((float)(sa / RMS_FRAME)), rp = sqrtf((float)(sp / RMS_FRAME));
        if (rms_a) rms_a[(size_t)c * n_frames + f] = ra;
        if (rms_p) rms_p[(size_t)c * n_frames + f] = rp;
        if (ra > rp / 2.0f && ra < rp * 4.0f) atomicAdd(pass_count + c, 1);   // integer: order-independent
    }
}

}  // namespace zns

using namespace zns;

extern "C" {

int zns_head_fwd_f32(int nbr, int B, int T, int C, const float* h, const float* const* w, const float* const* b,
                     float* emb, void* stream) {
    ZNS_CHECK_ARG(nbr >= 1 && nbr <= 2 && B > 0 && T > 0 && C > 0, "head_fwd: bad shape");
    Ptr2 pw{}, pb{};
    for (int i = 0; i < nbr; ++i) { pw.p[i] = w[i]; pb.p[i] = b[i]; }
    const int total = nbr * B * T;
    hipLaunchKernelGGL(head_fwd_kernel, dim3((total + 63) / 64), dim3(256), 0, (hipStream_t)stream, B, T, C, h, pw,
                       pb, emb, total);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int zns_head_bwd_f32(int nbr, int B, int T, int C, const float* h, const float* emb, const float* demb,
                     const float* demb1, const float* const* w, float inv_keep, float* dz, float* const* dw, float* const* db,
                     int accumulate, void* ws, size_t ws_bytes, void* stream) {
    ZNS_CHECK_ARG(nbr >= 1 && nbr <= 2 && B > 0 && T > 0 && C > 0, "head_bwd: bad shape");
    ZNS_CHECK_ARG(demb1 == nullptr || nbr == 2, "head_bwd: demb1 (branch 1's own gradient tensor) needs nbr = 2");
    Ptr2 pd{};
    pd.p[0] = demb;
    pd.p[1] = demb1 ? demb1 : demb + (size_t)B * T;
    const size_t need = ((size_t)nbr * B * C + (size_t)nbr * B) * sizeof(float);
    if (ws_bytes < need) { set_error("head_bwd: workspace %zu < %zu", ws_bytes, need); return ZNS_ERR_WORKSPACE; }
    Ptr2 pw{}; MPtr2 pdw{}, pdb{};
    for (int i = 0; i < nbr; ++i) { pw.p[i] = w[i]; pdw.p[i] = dw[i]; pdb.p[i] = db[i]; }
    float* part_w = (float*)ws;
    float* part_b = part_w + (size_t)nbr * B * C;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(head_bwd_kernel, dim3(nbr * B * C), dim3(128), 0, st, B, T, C, h, emb, pd, pw, inv_keep, dz,
                       part_w, part_b);
    ZNS_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_bwd_reduce, dim3((nbr * (C + 1) + 127) / 128), dim3(128), 0, st, B, C, part_w, part_b,
                       pdw, pdb, accumulate, nbr);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int zns_down_fwd_f32(const float* a, const float* p, size_t n, int reduction, float* out, void* stream) {
    ZNS_CHECK_ARG(reduction == ZNS_REDUCE_MAX || reduction == ZNS_REDUCE_MEAN, "down_fwd: bad reduction");
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(down_fwd_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, (hipStream_t)stream, a, p, n,
                       reduction, out);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int zns_down_bwd_f32(const float* a, const float* p, const float* dout, size_t n, int reduction, float* da, float* dp,
                     void* stream) {
    ZNS_CHECK_ARG(reduction == ZNS_REDUCE_MAX || reduction == ZNS_REDUCE_MEAN, "down_bwd: bad reduction");
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(down_bwd_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, (hipStream_t)stream, a, p, dout, n,
                       reduction, da, dp);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

size_t zns_ntxent_workspace_bytes(int rows) {
    if (rows < 1) return 256;
    return (((size_t)3 * rows * rows + 4 * (size_t)rows) * sizeof(double) + 255) & ~(size_t)255;
}

int zns_ntxent_f32(const float* a, const float* p, int rows, int D, int batch_len, float tau, float* out3, float* loss_out,
                   float* ga, float* gp, float grad_scale, void* ws, size_t ws_bytes, void* stream) {
    // the log-softmax pass is ONE workgroup (a thread per row, O(rows^2)): sized for batch sizes the reference can hold
    // (config.yaml batch_size 16; a 2048-row batch is already 3 x 32 MB of scratch), refused beyond
    ZNS_CHECK_ARG(rows >= 1 && rows <= 2048, "ntxent: rows=%d (supported: 1..2048)", rows);
    ZNS_CHECK_ARG(D >= 1 && batch_len >= 2 && rows <= batch_len, "ntxent: bad D/batch_len");
    if (ws == nullptr || ws_bytes < zns_ntxent_workspace_bytes(rows)) {
        set_error("ntxent: workspace %zu < %zu", ws_bytes, zns_ntxent_workspace_bytes(rows));
        return ZNS_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t R = (size_t)rows;
    double* S = (double*)ws;        // float64 tables: see ntx_dots_kernel
    double* na2 = S + R * R;
    double* np2 = na2 + R;
    double* Ga = np2 + R;
    double* Gp = Ga + R * R;
    double* s1 = Gp + R * R;
    double* s2 = s1 + R;
    const long long nq = (long long)R * R + 2 * R;
    hipLaunchKernelGGL(ntx_dots_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, a, p, rows, D, S, na2, np2);
    ZNS_LAUNCH_CHECK();
    const int want_grad = (ga != nullptr || gp != nullptr) ? 1 : 0;
    hipLaunchKernelGGL(ntx_softmax_kernel, dim3(1), dim3(1024), 0, st, rows, batch_len, tau, grad_scale, (const double*)S,
                       (const double*)na2, (const double*)np2, Ga, Gp, s1, s2, out3, loss_out, want_grad);
    ZNS_LAUNCH_CHECK();
    if (want_grad) {
        hipLaunchKernelGGL(ntx_grad_kernel, dim3((unsigned)((D + 255) / 256), (unsigned)rows, 2), dim3(256), 0, st, a, p,
                           rows, D, (const double*)na2, (const double*)np2, (const double*)Ga, (const double*)Gp,
                           (const double*)s1, (const double*)s2, ga, gp);
        ZNS_LAUNCH_CHECK();
    }
    return ZNS_OK;
}

int zns_scale_f32(const float* scale, const float* x0, float* y0, size_t n0, const float* x1, float* y1, size_t n1,
                  void* stream) {
    ZNS_CHECK_ARG(scale != nullptr && (n0 == 0 || (x0 && y0)) && (n1 == 0 || (x1 && y1)), "scale: null pointer");
    if (n0 + n1 == 0) return ZNS_OK;
    const size_t want = (n0 + n1 + 255) / 256;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)(want > 2048 ? 2048 : want)), dim3(256), 0, (hipStream_t)stream, scale,
                       x0, y0, n0, x1, y1, n1);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int zns_bce_f32(const float* out, const float* target, size_t n, float* loss, float* dout, float grad_scale,
                void* stream) {
    ZNS_CHECK_ARG(n > 0, "bce: empty input");
    hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, out, target, n, loss, dout,
                       grad_scale);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int zns_adam_step_f32(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                      float eps, int step, float grad_scale, void* stream) {
    ZNS_CHECK_ARG(step >= 1, "adam: step must be >= 1");
    const int aligned = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    if (n == 0) return ZNS_OK;
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1);
    const float sqrt_bc2 = (float)sqrt(bc2);
    const size_t want = ((aligned ? n / 4 : n) + 255) / 256;
    const int blocks = (int)(want > 4096 ? 4096 : (want ? want : 1));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1, beta2, eps,
                       step_size, sqrt_bc2, grad_scale, aligned);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int zns_adam_multi_f32(const void* table, int n_tensors, size_t max_n, float lr, float beta1, float beta2, float eps,
                       int step, float grad_scale, void* stream) {
    ZNS_CHECK_ARG(step >= 1, "adam: step must be >= 1");
    ZNS_CHECK_ARG(table != nullptr && n_tensors >= 1 && n_tensors <= 65535, "adam_multi: bad table (n_tensors=%d)", n_tensors);
    if (max_n == 0) return ZNS_OK;
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    const size_t want = (max_n / 4 + 255) / 256;
    const int bx = (int)(want > 2048 ? 2048 : (want ? want : 1));
    hipLaunchKernelGGL(adam_multi_kernel, dim3(bx, n_tensors), dim3(256), 0, (hipStream_t)stream, (const AdamSlot*)table,
                       beta1, beta2, eps, (float)((double)lr / bc1), (float)sqrt(bc2), grad_scale);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int zns_shift_gather_f32(const float* bank, int bank_T, const int32_t* clip, const int32_t* start, int n_items,
                         int T, float* out, void* stream) {
    ZNS_CHECK_ARG(n_items > 0 && T > 0 && bank_T >= T, "shift_gather: bad shape");
    hipLaunchKernelGGL(shift_gather_kernel, dim3(2 * n_items * 96), dim3(128), 0, (hipStream_t)stream, bank, bank_T,
                       clip, start, n_items, T, out);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int zns_stem_mix_f32(const float* const* stems, int n_stems, long long n, float* out, void* stream) {
    ZNS_CHECK_ARG(n_stems >= 1 && n_stems <= 8 && n > 0, "stem_mix: need 1..8 stems and n > 0 (got %d, %lld)", n_stems, n);
    StemPtrs st{};
    for (int i = 0; i < n_stems; ++i) st.p[i] = stems[i];
    hipLaunchKernelGGL(stem_mix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, st,
                       n_stems, n, out);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int zns_stage_crops_f32(const long long* table, int n_files, int max_other, int n_samples, float* out, void* stream) {
    ZNS_CHECK_ARG(table != nullptr && out != nullptr && n_files >= 1 && max_other >= 1 && max_other <= 8 && n_samples > 0,
                  "stage_crops: need a table, 1..8 other stems per file and n_samples > 0 (got %d files, %d, %d)", n_files,
                  max_other, n_samples);
    const int bx = (n_samples + 2047) / 2048;      // 8 samples per thread
    hipLaunchKernelGGL(stage_crops_kernel, dim3((unsigned)bx, (unsigned)n_files), dim3(256), 0, (hipStream_t)stream, table,
                       3 + max_other, n_samples, out);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int zns_rms_frames(int n_samples) { return n_samples < 0 ? 0 : 1 + n_samples / RMS_HOP; }

int zns_rms_gate_f32(const float* anchor, const float* positive, const long long* start, int n_cand, int n_samples,
                     float* rms_a, float* rms_p, int* pass_count, void* stream) {
    ZNS_CHECK_ARG(n_cand >= 1 && n_samples > RMS_FRAME / 2,
                  "rms_gate: need n_cand >= 1 and more than %d samples for the reflect padding (got %d, %d)",
                  RMS_FRAME / 2, n_cand, n_samples);
    ZNS_CHECK_ARG(pass_count != nullptr, "rms_gate: pass_count is required");
    hipStream_t st = (hipStream_t)stream;
    ZNS_HIP(hipMemsetAsync(pass_count, 0, sizeof(int) * n_cand, st));
    const int n_frames = zns_rms_frames(n_samples);
    hipLaunchKernelGGL(rms_gate_kernel, dim3(n_frames, n_cand), dim3(64), 0, st, anchor, positive, start, n_samples,
                       n_frames, rms_a, rms_p, pass_count);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

}  // extern "C"
