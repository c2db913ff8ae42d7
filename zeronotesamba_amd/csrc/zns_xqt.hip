// generate_XQT on gfx950: log(|VQT or CQT| + 1e-9), 96 bins, hop 256, 16 kHz  (input_rep.py:11-57).
//
// The reference delegates to librosa 0.8.1 (vqt/cqt -> per-octave FFT filter bank + resampy kaiser_fast
// decimation).  Here the per-octave "rfft -> sparse basis product" is folded on the host (float64) into an
// equivalent strided complex FIR bank g[k][n] (<= 12 x 256 taps per octave), applied by one kernel that reads
// the decimation pyramid once and writes the log-magnitudes coalesced along time.  HBM-bound by design:
// no MFMA (12 filters x <=256 taps is not a dense contraction worth reshaping).
//
// Numerics that are restated exactly (see oracle/vqt_ref.py, which documents the recalled third-party
// algorithm; PARITY UNPINNED there): complex64 filter storage, complex64 basis after sparsify_rows(1%),
// sqrt(2^i) rescale, resampy's tap-by-tap float32 accumulation with float64 products, float32 /sqrt(0.5).
#include <algorithm>
#include <cmath>
#include <complex>
#include <mutex>
#include <vector>

#include "zns_common.h"
#include "zns_kernels.h"

namespace zns {

namespace {

constexpr int XQ_SR = 16000, XQ_HOP = 256, XQ_BINS = 96, XQ_BPO = 12, XQ_OCT = 8;
constexpr int KF_ZEROS = 16, KF_PREC = 9;
constexpr double KF_ROLLOFF = 0.85, KF_BETA = 8.555504641634386;
constexpr double HANN_BW = 1.50018310546875;
constexpr int NL = 32, NR = 31;   // decimation taps touched for the exact 2:1 ratio

struct DecTaps {
    double l[NL];   // taps applied to x[2t - i]
    double r[NR];   // taps applied to x[2t + 1 + k]
};

struct Bank {
    int n_fft[XQ_OCT];
    int hop[XQ_OCT];
    int off[XQ_OCT];        // float2 offset of octave i inside `fir`
    float2* fir = nullptr;  // device: [octave][n_fft][12] complex (tap-major: one frame step reads 12 neighbours)
    DecTaps* taps = nullptr;   // device: float64 decimation taps
    bool ready = false;
};
Bank g_bank[2];
std::mutex g_bank_mu;

double sinc_pi(double x) { return x == 0.0 ? 1.0 : std::sin(M_PI * x) / (M_PI * x); }

void build_taps(double* left, double* right) {
    // resampy/filters.py::sinc_window(16, 9, kaiser(beta), 0.85), then interp_win *= 0.5, step = 256
    const int num_bits = 1 << KF_PREC, n = num_bits * KF_ZEROS, step = num_bits / 2, nwin = n + 1;
    const double i0b = std::cyl_bessel_i(0.0, KF_BETA);
    auto win = [&](int idx) {
        const double pos = (double)KF_ZEROS * idx / n;                      // linspace(0, zeros, n+1)
        const double sw = KF_ROLLOFF * sinc_pi(KF_ROLLOFF * pos);
        const double r = (double)idx / n;                                   // kaiser(2n+1)[n + idx]
        const double taper = std::cyl_bessel_i(0.0, KF_BETA * std::sqrt(std::max(0.0, 1.0 - r * r))) / i0b;
        return 0.5 * taper * sw;
    };
    for (int i = 0; i < nwin / step && i < NL; ++i) left[i] = win(i * step);
    for (int k = 0; k < (nwin - step) / step && k < NR; ++k) right[k] = win(step + k * step);
}

double gamma_of(int mode) {
    const double alpha = std::pow(2.0, 1.0 / XQ_BPO) - 1.0;
    return mode == ZNS_MODE_VQT ? 24.7 * alpha / 0.108 : 0.0;
}

// host: per-octave folded FIR bank, float64 -> float2
void build_fir(int mode, std::vector<std::vector<std::complex<double>>>& fir, int* n_fft_out) {
    const double alpha = std::pow(2.0, 1.0 / XQ_BPO) - 1.0, Q = 1.0 / alpha, gamma = gamma_of(mode);
    const double fmin = 440.0 * std::pow(2.0, (12 - 69) / 12.0);   // note_to_hz("C0")
    std::vector<double> full_len(XQ_BINS);
    for (int k = 0; k < XQ_BINS; ++k) full_len[k] = Q * XQ_SR / (fmin * std::pow(2.0, (double)k / XQ_BPO) + gamma / alpha);
    const double fmin_t = fmin * std::pow(2.0, (double)(XQ_BINS - XQ_BPO) / XQ_BPO);
    fir.assign(XQ_OCT, {});
    for (int i = 0; i < XQ_OCT; ++i) {
        const double sr = XQ_SR / std::pow(2.0, i), f0 = fmin_t * std::pow(2.0, -i);
        double lengths[XQ_BPO], freqs[XQ_BPO], maxlen = 0;
        for (int k = 0; k < XQ_BPO; ++k) {
            freqs[k] = f0 * std::pow(2.0, (double)k / XQ_BPO);
            lengths[k] = Q * sr / (freqs[k] + gamma / alpha);
            maxlen = std::max(maxlen, lengths[k]);
        }
        const int n_fft = (int)std::pow(2.0, std::ceil(std::log2(maxlen)));
        n_fft_out[i] = n_fft;
        const int nf = n_fft / 2 + 1;
        std::vector<std::complex<double>> out((size_t)XQ_BPO * n_fft);
        for (int k = 0; k < XQ_BPO; ++k) {
            // filters.constant_q: arange(-ilen//2, ilen//2) * 2 pi j f / sr, periodic hann, L1 norm, pad_center
            const double lo = std::floor(-lengths[k] / 2.0), hi = std::floor(lengths[k] / 2.0);
            const int n = (int)(hi - lo);
            std::vector<std::complex<double>> sig(n);
            double l1 = 0;
            for (int j = 0; j < n; ++j) {
                const double t = lo + j;
                const double w = 0.5 - 0.5 * std::cos(2.0 * M_PI * j / n);
                sig[j] = std::polar(1.0, t * 2.0 * M_PI * freqs[k] / sr) * w;
                l1 += std::abs(sig[j]);
            }
            std::vector<std::complex<float>> basis(n_fft, std::complex<float>(0.f, 0.f));
            const int lpad = (n_fft - n) / 2;
            for (int j = 0; j < n; ++j) basis[lpad + j] = std::complex<float>(sig[j] / l1);          // complex64 storage
            const double scale = lengths[k] / (double)n_fft;                                          // basis *= lengths/n_fft,
            for (int j = 0; j < n_fft; ++j) basis[j] = std::complex<float>(std::complex<double>(basis[j]) * scale);  // rounded to complex64
            // fft (float64) of the complex64 basis, keep f = 0 .. n_fft/2
            std::vector<std::complex<double>> F(nf);
            std::vector<double> mags(nf);
            double norm = 0;
            for (int f = 0; f < nf; ++f) {
                std::complex<double> s(0, 0);
                for (int j = 0; j < n_fft; ++j)
                    s += std::complex<double>(basis[j]) * std::polar(1.0, -2.0 * M_PI * (double)((long long)f * j % n_fft) / n_fft);
                F[f] = s;
                mags[f] = std::abs(s);
                norm += mags[f];
            }
            // util.sparsify_rows(quantile = 0.01)
            std::vector<double> srt(mags);
            std::sort(srt.begin(), srt.end());
            double cum = 0;
            int tidx = 0;
            for (int f = 0; f < nf; ++f) {
                cum += srt[f] / norm;
                if (!(cum < 0.01)) { tidx = f; break; }
            }
            const double thr = srt[tidx];
            const double oct_scale = std::sqrt(std::pow(2.0, i));
            std::vector<std::complex<float>> Bs(nf);
            for (int f = 0; f < nf; ++f) {
                std::complex<float> v = mags[f] >= thr ? std::complex<float>(F[f]) : std::complex<float>(0.f, 0.f);
                Bs[f] = std::complex<float>(std::complex<double>(v) * oct_scale);                        // fft_basis[:] *= sqrt(2**i)
            }
            // fold rfft + sparse product into a time-domain FIR, apply the final 1/sqrt(length)
            const int kbin = XQ_BINS - XQ_BPO * (i + 1) + k;
            const double inv = 1.0 / std::sqrt(full_len[kbin]);
            for (int j = 0; j < n_fft; ++j) {
                std::complex<double> s(0, 0);
                for (int f = 0; f < nf; ++f)
                    s += std::complex<double>(Bs[f]) * std::polar(1.0, -2.0 * M_PI * (double)((long long)f * j % n_fft) / n_fft);
                out[(size_t)k * n_fft + j] = s * inv;
            }
        }
        fir[i] = std::move(out);
    }
}

int ensure_bank(int mode, Bank** out) {
    std::lock_guard<std::mutex> lk(g_bank_mu);
    Bank& b = g_bank[mode];
    if (!b.ready) {
        std::vector<std::vector<std::complex<double>>> fir;
        build_fir(mode, fir, b.n_fft);
        size_t total = 0;
        for (int i = 0; i < XQ_OCT; ++i) {
            b.hop[i] = XQ_HOP >> i;
            b.off[i] = (int)total;
            total += fir[i].size();
        }
        std::vector<float2> h(total);
        for (int i = 0; i < XQ_OCT; ++i)
            for (int k = 0; k < XQ_BPO; ++k)
                for (int j = 0; j < b.n_fft[i]; ++j) {
                    const std::complex<double> c = fir[i][(size_t)k * b.n_fft[i] + j];
                    h[b.off[i] + (size_t)j * XQ_BPO + k] = make_float2((float)c.real(), (float)c.imag());
                }
        double taps[NL + NR];
        build_taps(taps, taps + NL);
        ZNS_HIP(hipMalloc(&b.fir, total * sizeof(float2)));
        ZNS_HIP(hipMemcpy(b.fir, h.data(), total * sizeof(float2), hipMemcpyHostToDevice));
        ZNS_HIP(hipMalloc(&b.taps, sizeof(DecTaps)));
        ZNS_HIP(hipMemcpy(b.taps, taps, sizeof(taps), hipMemcpyHostToDevice));
        b.ready = true;
    }
    *out = &b;
    return ZNS_OK;
}

// lengths of the decimation pyramid: len[0] = n, len[i] = ceil(len[i-1] / 2)
void pyramid(int n, int* len, size_t* off) {
    len[0] = n;
    off[0] = 0;
    size_t o = 0;
    for (int i = 1; i < XQ_OCT; ++i) {
        len[i] = (len[i - 1] + 1) / 2;
        off[i] = o;
        o += ((size_t)len[i] + 3) / 4 * 4;
    }
    off[XQ_OCT] = o;
}

}  // namespace


// thread -> 4 consecutive outputs of one clip.  The 72-sample input window is fetched once (18 aligned 16-byte
// loads when the rows allow it) and kept in registers as float64; the 63 float64 taps are wave-uniform (scalar
// cache).  Exact restatement of resampy's accumulation: y (float32) += weight (float64) * x, i.e. the float64
// product-sum is rounded to float32 after EVERY tap, left wing i = 0..31 first, then the right wing.
__global__ void __launch_bounds__(256) decimate2_kernel(const float* __restrict__ in, long long in_stride, int n_in,
                                                        float* __restrict__ out, long long out_stride, int n_quads,
                                                        const DecTaps* __restrict__ tp, int n_clips, int vec_ok) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (long long)n_clips * n_quads) return;
    const int q = (int)(o % n_quads), c = (int)(o / n_quads);
    const float* x = in + (long long)c * in_stride;
    const int t0 = q * 4;
    const int n_out = n_in / 2;   // resampy: int(n * 0.5); librosa pads to ceil with zeros
    const int base = 2 * t0 - 32; // xw[j] = x[base + j]
    double xw[72];
    if (vec_ok) {
#pragma unroll
        for (int v = 0; v < 18; ++v) {
            const int i0 = base + 4 * v;
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i0 >= 0 && i0 < n_in) f = *reinterpret_cast<const float4*>(x + i0);
            xw[4 * v + 0] = (double)f.x;
            xw[4 * v + 1] = (i0 + 1 < n_in) ? (double)f.y : 0.0;
            xw[4 * v + 2] = (i0 + 2 < n_in) ? (double)f.z : 0.0;
            xw[4 * v + 3] = (i0 + 3 < n_in) ? (double)f.w : 0.0;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 72; ++j) {
            const int i0 = base + j;
            xw[j] = (i0 >= 0 && i0 < n_in) ? (double)x[i0] : 0.0;
        }
    }
    // a tap outside the signal contributes w * 0 = +0: identical to resampy skipping it
    double acc[4] = {0.0, 0.0, 0.0, 0.0};   // always holds a float32-representable value
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const double w = tp->l[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = (double)(float)(acc[k] + w * xw[32 + 2 * k - i]);
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const double w = tp->r[j];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = (double)(float)(acc[k] + w * xw[32 + 2 * k + 1 + j]);
    }
    float y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = (t0 + k < n_out) ? (float)acc[k] / 0.70710677f : 0.f;   // y_hat /= sqrt(0.5) in float32
    *reinterpret_cast<float4*>(out + (long long)c * out_stride + t0) = make_float4(y[0], y[1], y[2], y[3]);
}

struct XqtLevels {
    const float* y[XQ_OCT];
    long long stride[XQ_OCT];
    int len[XQ_OCT];
    int n_fft[XQ_OCT];
    int hop[XQ_OCT];
    int off[XQ_OCT];
};

constexpr int XQ_FPB = 64;   // frames per block
constexpr int XQ_GRP = 4;    // filter groups per block (3 filters each): 256 threads

struct TapChunk {
    float2 c[4][3];   // 4 consecutive taps x 3 filters, wave-uniform
};

__device__ __forceinline__ void load_taps(TapChunk& d, const float2* __restrict__ g) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) d.c[j][k] = g[j * XQ_BPO + k];
}

__device__ __forceinline__ void apply_taps(const TapChunk& tc, const float* x, float* re, float* im) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            re[k] = fmaf(tc.c[j][k].x, x[j], re[k]);
            im[k] = fmaf(tc.c[j][k].y, x[j], im[k]);
        }
}

// block: 64 consecutive frames of one (octave, clip); 256 threads = 64 frames x 4 groups of 3 filters.
// The frames' windows are staged into LDS with coalesced loads, 8 in flight per thread (row stride n_fft + 1:
// conflict-free per-frame reads).  The filter taps are wave-uniform: they stream through the scalar cache from the
// tap-major [n][12] device layout, one 4-tap chunk ahead of the FMAs (the x values are fetched from LDS one chunk
// ahead too, because LDS and scalar loads share a wait counter).
__global__ void __launch_bounds__(256) xqt_filter_kernel(XqtLevels L, const float2* __restrict__ fir, int n_frames,
                                                         float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float rows[];
    const int oct = blockIdx.y, clip = blockIdx.z;
    const int n_fft = L.n_fft[oct], hop = L.hop[oct], len = L.len[oct];
    const int lg = 31 - __builtin_clz(n_fft);
    const int f0 = blockIdx.x * XQ_FPB;
    const float* y = L.y[oct] + (long long)clip * L.stride[oct];
    const int rs = n_fft + 1;
    // XQ_FPB * n_fft is a multiple of 256 * 4 (n_fft >= 16)
    for (int i0 = threadIdx.x; i0 < XQ_FPB * n_fft; i0 += 256 * 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 256;
            const int f = i >> lg, n = i & (n_fft - 1);
            int idx = (f0 + f) * hop - n_fft / 2 + n;
            if (idx < 0) idx = -idx;                       // np.pad(mode="reflect")
            if (idx >= len) idx = 2 * (len - 1) - idx;
            idx = idx < 0 ? 0 : idx;                       // frames past the end are never written out
            v[u] = (f0 + f < n_frames) ? y[idx] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 256;
            rows[(i >> lg) * rs + (i & (n_fft - 1))] = v[u];
        }
    }
    __syncthreads();
    const int f = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float2* gf = fir + L.off[oct] + g * 3;
    const float* xr = rows + f * rs;
    float re[3] = {0.f, 0.f, 0.f}, im[3] = {0.f, 0.f, 0.f};
    TapChunk ta, tb;
    float xa[4], xb[4];
    load_taps(ta, gf);
#pragma unroll
    for (int j = 0; j < 4; ++j) xa[j] = xr[j];
    for (int n = 0; n < n_fft; n += 8) {   // n_fft is a power of two >= 16
        load_taps(tb, gf + (n + 4) * XQ_BPO);
#pragma unroll
        for (int j = 0; j < 4; ++j) xb[j] = xr[n + 4 + j];
        apply_taps(ta, xa, re, im);
        const int nn = (n + 8 < n_fft) ? n + 8 : 0;   // last prefetch wraps to a valid address, result unused
        load_taps(ta, gf + nn * XQ_BPO);
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[j] = xr[nn + j];
        apply_taps(tb, xb, re, im);
    }
    const int t = f0 + f;
    if (t >= n_frames) return;
    const int k0 = XQ_BINS - XQ_BPO * (oct + 1) + g * 3;
    float* o = out + ((size_t)clip * XQ_BINS + k0) * n_frames + t;
#pragma unroll
    for (int k = 0; k < 3; ++k) o[(size_t)k * n_frames] = logf(sqrtf(re[k] * re[k] + im[k] * im[k]) + 1e-9f);
}

}  // namespace zns

using namespace zns;

extern "C" {

int zns_xqt_host_bank(int mode, int* n_fft8, float* fir, size_t fir_floats, double* taps63) {
    if (mode != ZNS_MODE_VQT && mode != ZNS_MODE_CQT) {
        set_error("Mode can only be vqt or cqt!");
        return ZNS_ERR_MODE;
    }
    std::vector<std::vector<std::complex<double>>> bank;
    int nf[XQ_OCT];
    build_fir(mode, bank, nf);
    size_t total = 0;
    for (int i = 0; i < XQ_OCT; ++i) total += bank[i].size() * 2;
    if (n_fft8) for (int i = 0; i < XQ_OCT; ++i) n_fft8[i] = nf[i];
    if (fir) {
        ZNS_CHECK_ARG(fir_floats >= total, "xqt_host_bank: need %zu floats", total);
        size_t o = 0;
        for (int i = 0; i < XQ_OCT; ++i)
            for (auto& c : bank[i]) { fir[o++] = (float)c.real(); fir[o++] = (float)c.imag(); }
    }
    if (taps63) build_taps(taps63, taps63 + NL);
    return ZNS_OK;
}

int zns_xqt_frames(int n_samples) { return n_samples < 0 ? 0 : 1 + n_samples / XQ_HOP; }

size_t zns_xqt_workspace_bytes(int n_clips, int n_samples, int mode) {
    (void)mode;
    if (n_clips <= 0 || n_samples <= 0) return 256;
    int len[XQ_OCT];
    size_t off[XQ_OCT + 1];
    pyramid(n_samples, len, off);
    return ((size_t)n_clips * off[XQ_OCT] * sizeof(float) + 255) & ~(size_t)255;
}

int zns_xqt_f32(const float* wav, int n_clips, int n_samples, int sample_rate, int mode, float* out, void* ws,
                size_t ws_bytes, void* stream) {
    if (mode != ZNS_MODE_VQT && mode != ZNS_MODE_CQT) {
        set_error("Mode can only be vqt or cqt!");
        return ZNS_ERR_MODE;
    }
    ZNS_CHECK_ARG(sample_rate == XQ_SR, "xqt: only sample_rate=16000 is built (the reference's working rate), got %d",
                  sample_rate);
    ZNS_CHECK_ARG(n_clips >= 1 && n_samples >= 1, "xqt: empty input (n_clips=%d n_samples=%d)", n_clips, n_samples);
    Bank* b = nullptr;
    int rc = ensure_bank(mode, &b);
    if (rc) return rc;
    int len[XQ_OCT];
    size_t off[XQ_OCT + 1];
    pyramid(n_samples, len, off);
    for (int i = 0; i < XQ_OCT; ++i)
        ZNS_CHECK_ARG(len[i] > b->n_fft[i] / 2, "xqt: signal of %d samples is too short for octave %d (reflect padding)",
                      n_samples, i);
    const size_t need = zns_xqt_workspace_bytes(n_clips, n_samples, mode);
    if (ws_bytes < need) {
        set_error("xqt: workspace %zu < %zu", ws_bytes, need);
        return ZNS_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* pyr = (float*)ws;
    XqtLevels L{};
    L.y[0] = wav;
    L.stride[0] = n_samples;
    for (int i = 0; i < XQ_OCT; ++i) {
        L.len[i] = len[i];
        L.n_fft[i] = b->n_fft[i];
        L.hop[i] = b->hop[i];
        L.off[i] = b->off[i];
        if (i > 0) {
            L.y[i] = pyr + (size_t)n_clips * off[i];
            L.stride[i] = (long long)(((size_t)len[i] + 3) / 4 * 4);
        }
    }
    for (int i = 1; i < XQ_OCT; ++i) {
        const int n_quads = (len[i] + 3) / 4;
        const long long total = (long long)n_clips * n_quads;
        const int vec_ok = (L.stride[i - 1] % 4 == 0) && ((uintptr_t)L.y[i - 1] % 16 == 0);
        hipLaunchKernelGGL(decimate2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, L.y[i - 1],
                           L.stride[i - 1], len[i - 1], (float*)L.y[i], L.stride[i], n_quads, b->taps, n_clips, vec_ok);
        ZNS_LAUNCH_CHECK();
    }
    const int n_frames = zns_xqt_frames(n_samples);
    int max_fft = 0;
    for (int i = 0; i < XQ_OCT; ++i) max_fft = std::max(max_fft, b->n_fft[i]);
    const size_t lds = (size_t)XQ_FPB * (max_fft + 1) * sizeof(float);
    hipLaunchKernelGGL(xqt_filter_kernel, dim3((n_frames + XQ_FPB - 1) / XQ_FPB, XQ_OCT, n_clips), dim3(64 * XQ_GRP), lds, st, L, b->fir,
                       n_frames, out);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

}  // extern "C"
