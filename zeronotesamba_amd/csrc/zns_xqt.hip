// generate_XQT on gfx950: log(|VQT or CQT| + 1e-9), 96 bins, hop 256, 16 kHz  (input_rep.py:11-57).
//
// The reference delegates to librosa 0.8.1 (vqt/cqt -> per-octave FFT filter bank + resampy kaiser_fast
// decimation).  Here the per-octave "rfft -> sparse basis product" is folded on the host (float64) into an
// equivalent strided complex FIR bank g[k][n] (<= 12 x 256 taps per octave).  Two stages:
//   * the octave pyramid (seven 2:1 decimations).  Default arithmetic: resampy's taps (rounded once to float32) in
//     resampy's order, one float32 fused multiply-add per tap (issued in pairs as v_pk_fma_f32, zns_xqt.h) -- HBM-bound as
//     SURVEY 8(d) prices the stage.  resampy itself
//     rounds a float64 product-sum to float32 after every tap; the two differ in a fraction of those roundings by one unit in
//     the last place, which is the size of the restatement's own rounding noise (tests/test_vqt.py states the distance).
//     ZNS_XQT_EXACT selects the validation implementation that emulates resampy's rounding bit for bit: three dependent
//     half-rate instructions per tap (v_fma_f64, v_cvt_f32_f64, v_cvt_f64_f32), bound by float64 issue (12.7e12 taps/s);
//   * the filter bank as per-octave GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, A operand in registers).
// This file holds the per-stage launches (decimate_chain_f32_kernel / decimate_chain_kernel per level or fused tail, then
// xqt_filter_kernel): the plan for stages = 1 / 2, for the cqt bank and for the exact pyramid.  The default for the whole
// vqt transform is the fused kernel of zns_xqt_stream.hip (same arithmetic, one launch, a quarter of the HBM traffic).
//
// Numerics that are restated exactly (see oracle/vqt_ref.py, which documents the recalled third-party
// algorithm; PARITY UNPINNED there): complex64 filter storage, complex64 basis after sparsify_rows(1%),
// sqrt(2^i) rescale, float32 /sqrt(0.5); with ZNS_XQT_EXACT also resampy's tap-by-tap float32 accumulation of float64 products.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <complex>
#include <mutex>
#include <vector>

#include "zns_common.h"
#include "zns_kernels.h"
#include "zns_xqt.h"

namespace zns {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int KF_ZEROS = 16, KF_PREC = 9;
constexpr double KF_ROLLOFF = 0.85, KF_BETA = 8.555504641634386;
constexpr double HANN_BW = 1.50018310546875;
Bank g_bank[2][ZNS_MAX_DEVICES];   // per (mode, device): the FIR bank lives in that device's memory
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

}  // namespace

int ensure_bank(int mode, Bank** out) {
    std::lock_guard<std::mutex> lk(g_bank_mu);
    Bank& b = g_bank[mode][current_device_slot()];
    if (!b.ready) {
        std::vector<std::vector<std::complex<double>>> fir;
        build_fir(mode, fir, b.n_fft);
        size_t total = 0;
        for (int i = 0; i < XQ_OCT; ++i) {
            b.hop[i] = XQ_HOP >> i;
            b.off[i] = (int)total;
            total += fir[i].size();
        }
        // A operand of v_mfma_f32_32x32x2_f32 for the GEMM  D[row][frame] = sum_n A[row][n] * B[n][frame]:
        // row 2k = Re g_k, row 2k+1 = Im g_k (k < 12; rows 24..31 are zero), k-step s holds taps n = 2s, 2s+1;
        // lane l supplies A[l & 31][2s + (l >> 5)]
        size_t tota = 0;
        for (int i = 0; i < XQ_OCT; ++i) { b.offa[i] = (int)tota; tota += (size_t)(b.n_fft[i] / 2) * 64; }
        std::vector<float> ha(tota, 0.f);
        for (int i = 0; i < XQ_OCT; ++i)
            for (int sidx = 0; sidx < b.n_fft[i] / 2; ++sidx)
                for (int l = 0; l < 64; ++l) {
                    const int row = l & 31, n = 2 * sidx + (l >> 5);
                    if (row >= 2 * XQ_BPO) continue;
                    const std::complex<double> c = fir[i][(size_t)(row >> 1) * b.n_fft[i] + n];
                    ha[b.offa[i] + (size_t)sidx * 64 + l] = (float)((row & 1) ? c.imag() : c.real());
                }
        double taps[NL + NR];
        build_taps(taps, taps + NL);
        ZNS_HIP(hipMalloc(&b.fira, tota * sizeof(float)));
        ZNS_HIP(hipMemcpy(b.fira, ha.data(), tota * sizeof(float), hipMemcpyHostToDevice));
        // A operand of v_mfma_f32_16x16x4_f32 (the fused kernel): row tile rt covers rows 16 rt .. 16 rt + 15 of the same 24 real
        // rows, k-step s holds taps 4 s .. 4 s + 3; lane l supplies A[16 rt + (l & 15)][4 s + (l >> 4)]
        {
            size_t tot = 0;
            for (int i = 0; i < XQ_OCT; ++i) { b.offa16[i] = (int)tot; tot += (size_t)2 * (b.n_fft[i] / 4) * 64; }
            std::vector<float> h16(tot, 0.f);
            for (int i = 0; i < XQ_OCT; ++i)
                for (int rt = 0; rt < 2; ++rt)
                    for (int sidx = 0; sidx < b.n_fft[i] / 4; ++sidx)
                        for (int l = 0; l < 64; ++l) {
                            const int row = 16 * rt + (l & 15), n = 4 * sidx + (l >> 4);
                            if (row >= 2 * XQ_BPO) continue;
                            const std::complex<double> c = fir[i][(size_t)(row >> 1) * b.n_fft[i] + n];
                            h16[b.offa16[i] + ((size_t)rt * (b.n_fft[i] / 4) + sidx) * 64 + l] = (float)((row & 1) ? c.imag() : c.real());
                        }
            ZNS_HIP(hipMalloc(&b.fira16, tot * sizeof(float)));
            ZNS_HIP(hipMemcpy(b.fira16, h16.data(), tot * sizeof(float), hipMemcpyHostToDevice));
        }
        ZNS_HIP(hipMalloc(&b.taps, sizeof(DecTaps)));
        ZNS_HIP(hipMemcpy(b.taps, taps, sizeof(taps), hipMemcpyHostToDevice));
        for (int i = 0; i < NL; ++i) b.tapsf[i] = (float)taps[i];
        b.ready = true;
    }
    *out = &b;
    return ZNS_OK;
}

namespace {

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


// ---------------------------------------------------------------------------------------------------------
// Decimation chain: `nlev` consecutive 2:1 decimations of the pyramid in ONE launch.
//
// Arithmetic (exact restatement of resampy's accumulation): y (float32) += weight (float64) * x, i.e. the float64
// product-sum is rounded to float32 after EVERY tap, left wing i = 0..31 first, then the right wing k = 0..30.  One
// tap = v_fma_f64 + v_cvt_f32_f64 + v_cvt_f64_f32, three dependent half-rate instructions: 12.35 cycles per
// wave-instruction-triple per SIMD measured (tools/ubench_f64.hip), i.e. 12.7e12 taps/s per chip -- that issue rate,
// not HBM, bounds this kernel (1.0e7 taps per 10 s clip, 3.2 ms per 4096 clips at the ceiling).
//
// Structure.  A block owns 4*GL outputs of the chain's LAST level of one clip and computes, level by level, exactly
// the (overlapping) ranges the next level needs; every level's outputs go to global memory (the filter kernel reads the
// pyramid) and, as float64, into an LDS window that is the next level's input -- a sample is converted to float64 once,
// not once per thread that touches it.  Thread = 4 consecutive outputs (group g), 4 independent accumulation chains.
// Window layout: local sample m (origin = 2*T0 - 32, a multiple of 8) lives at [(m & 7) * S + (m >> 3)]: the 14 samples
// a chunk of 8 taps needs are then 14 conflict-free ds_read_b64 (consecutive lanes -> consecutive columns) at
// compile-time offsets from one per-chunk pointer.  Taps arrive 8 at a time through the scalar cache (all 63 at once
// need 126 SGPRs: the previous kernel spilled 52 of them to VGPR lanes).
//
// Ranges.  Level j (1..nlev) of block b computes groups g < G[j] starting at output T0[j]:
//   T0[nlev] = b * 4 * GL, G[nlev] = GL;   T0[j] = 2 T0[j+1] - 32, G[j] = 2 G[j+1] + 18   (what level j+1 reads)
// Outputs outside [0, n_out) are zeros (outside the signal; [n_out, ceil(n/2)) is librosa's fix_length padding).
// A level's outputs are written to global memory by the block that owns them (its share of the last level, scaled).
constexpr int DC_S = 265;                 // columns per window row: windows of up to 8 * DC_S = 2120 samples (G <= 256)
constexpr int DC_WIN = 8 * DC_S;
constexpr int DC_MAXLEV = 4;

struct DecChain {
    const float* src;                     // level feeding the chain: [n_clips][src_stride]
    long long src_stride;
    int src_len;
    int src_vec;                          // rows are 16-byte aligned: float4 staging loads
    int nlev, GL, blocks_per_clip;
    float* dst[DC_MAXLEV];                // outputs of level j = 1..nlev
    long long dst_stride[DC_MAXLEV];
    int n_out[DC_MAXLEV];                 // resampy's output count of that level: int(n_in * 0.5)
    int G[DC_MAXLEV];
};

__device__ __forceinline__ int dc_addr(int m) { return (m & 7) * DC_S + (m >> 3); }

__global__ void __launch_bounds__(256) selftest_div_kernel(unsigned long long* __restrict__ bad, unsigned* __restrict__ first) {
    // every float32 bit pattern: 2^24 per block-row, grid-stride
    unsigned long long n_bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * 256) {
        const unsigned bits = (unsigned)i;
        const float x = __uint_as_float(bits);
        const float want = x / DC_SQRT_HALF;
        const float got = dc_div_sqrt_half((double)x);
        const bool nan_w = want != want, nan_g = got != got;
        if (nan_w ? !nan_g : (__float_as_uint(want) != __float_as_uint(got))) {
            if (n_bad == 0) atomicMin(first, bits);
            ++n_bad;
        }
    }
    if (n_bad) atomicAdd(bad, n_bad);
}

// NT taps of one wing: xv[n] = window sample (8 * col0 + 1 + n) of this thread's group; left wing tap u of output k reads
// xv[7 + 2k - u], right wing xv[2k + u]
template <int NT, bool LEFT, int RS = DC_S>
__device__ __forceinline__ void dc_chunk(const double* __restrict__ p, const double* __restrict__ w, double (&acc)[4]) {
    double xv[14];
#pragma unroll
    for (int n = 0; n < 6 + NT; ++n) xv[n] = p[((1 + n) & 7) * RS + ((1 + n) >> 3)];
    double wv[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) wv[u] = w[u];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = (double)(float)(acc[k] + wv[u] * xv[LEFT ? 7 + 2 * k - u : 2 * k + u]);
}

__global__ void __launch_bounds__(256) decimate_chain_kernel(DecChain P, const DecTaps* __restrict__ tp) {
    extern __shared__ __attribute__((aligned(16))) double dwin[];     // [2][DC_WIN] (one buffer when nlev == 1)
    const int clip = blockIdx.x / P.blocks_per_clip, b = blockIdx.x - clip * P.blocks_per_clip;
    // T0[j] - 32 = 2 (T0[j+1] - 32)  ->  closed form, no runtime-indexed array
    auto T0 = [&](int j) { return 32 + (b * 4 * P.GL - 32) * (1 << (P.nlev - 1 - j)); };
    // ---- stage the source window [o, o + 8 G[0] + 72) as float64, zero outside the signal
    {
        const int o = 2 * T0(0) - 32, nwin = 8 * P.G[0] + 72;
        const float* x = P.src + (long long)clip * P.src_stride;
        if (P.src_vec) {
            for (int q = threadIdx.x; q < nwin / 4; q += 256) {
                const int i0 = o + 4 * q;
                float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i0 >= 0 && i0 < P.src_len) f = *reinterpret_cast<const float4*>(x + i0);   // rows are padded to 4
                double* d = dwin + ((4 * q) & 7) * DC_S + (q >> 1);
                d[0] = (double)f.x;
                d[DC_S] = (i0 + 1 < P.src_len) ? (double)f.y : 0.0;
                d[2 * DC_S] = (i0 + 2 < P.src_len) ? (double)f.z : 0.0;
                d[3 * DC_S] = (i0 + 3 < P.src_len) ? (double)f.w : 0.0;
            }
        } else {
            for (int m = threadIdx.x; m < nwin; m += 256) {
                const int i0 = o + m;
                dwin[dc_addr(m)] = (i0 >= 0 && i0 < P.src_len) ? (double)x[i0] : 0.0;
            }
        }
    }
    __syncthreads();
    for (int j = 0; j < P.nlev; ++j) {
        const double* __restrict__ src = dwin + (j & 1) * DC_WIN;
        double* __restrict__ nxt = dwin + ((j + 1) & 1) * DC_WIN;
        const bool last = j + 1 == P.nlev;
        const int own_lo = (b * 4 * P.GL) << (P.nlev - 1 - j), own_hi = ((b + 1) * 4 * P.GL) << (P.nlev - 1 - j);
        float* __restrict__ y = P.dst[j] + (long long)clip * P.dst_stride[j];
        const int T0j = T0(j);
        for (int g = threadIdx.x; g < P.G[j]; g += 256) {
            double acc[4] = {0.0, 0.0, 0.0, 0.0};   // always holds a float32-representable value
            // serial chunks of 8 taps (taps through the scalar cache): 48 registers, 8 waves per SIMD hide every wait (a
            // software-pipelined form with all taps in scalar registers, 83 registers / 5 waves, measured 9 % slower here)
            const double* p = src + g + 3;
#pragma unroll 1
            for (int c = 0; c < 4; ++c, --p) dc_chunk<8, true>(p, tp->l + 8 * c, acc);
            p = src + g + 4;
#pragma unroll 1
            for (int c = 0; c < 3; ++c, ++p) dc_chunk<8, false>(p, tp->r + 8 * c, acc);
            dc_chunk<7, false>(p, tp->r + 24, acc);
            const int t0 = T0j + 4 * g;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = (t0 + k >= 0 && t0 + k < P.n_out[j]) ? dc_div_sqrt_half(acc[k]) : 0.f;   // y_hat /= sqrt(0.5), float32
            if (t0 >= own_lo && t0 < own_hi && (long long)t0 < P.dst_stride[j])
                *reinterpret_cast<float4*>(y + t0) = make_float4(v[0], v[1], v[2], v[3]);
            if (!last) {
                double* d = nxt + (4 * (g & 1)) * DC_S + (g >> 1);
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k * DC_S] = (double)v[k];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// The DEFAULT decimation chain: float32 arithmetic.  Same taps (rounded once to float32), same ORDER of accumulation as
// resampy (left wing i = 0..31, then right wing k = 0..30), one v_fma_f32 per tap: acc = fmaf(w, x, acc).  resampy's
// `y += w * x` with float64 w is RN32(acc + w64 x); the fused multiply-add is RN32(acc + w32 x): the two differ by the
// tap's rounding (2^-25 relative) ahead of the SAME rounding step, i.e. in a fraction of the rounding decisions by one
// unit in the last place -- the distance to the restatement is measured by tests/test_vqt.py (1e-5 of the loudest bin,
// the contract's 1e-4 in the log domain) and is the size of the restatement's own rounding noise (an exactly rounded
// float64 accumulation of the same taps lies 5x FURTHER from it).  A tap costs one full-rate instruction instead of the
// exact kernel's three half-rate ones (12.35 -> 2 cycles per wave and SIMD): the stage is HBM-bound, as SURVEY 8(d) prices it.
// Block / range structure, staging and the chained levels are those of decimate_chain_kernel; the LDS window holds
// float32: 4-sample chunk c lives in plane (c & 1) at column (c >> 1), so the 18 chunks a thread needs (samples 8g .. 8g + 71
// of the window for its outputs 4g .. 4g + 3) are 9 + 9 ds_read_b128 with consecutive lanes on consecutive 16-byte columns.
constexpr int DCF_PLANE = 4 * DC_S + 44;      // floats per plane: 265 columns of one chunk, padded to 16 (mod 64) banks
constexpr int DCF_WIN = 2 * DCF_PLANE;

__device__ __forceinline__ void dcf_taps63(const float* __restrict__ win, int g, const TapPairs& w, float (&acc)[4]) {
    const float4* __restrict__ pa = reinterpret_cast<const float4*>(win) + g;
    const float4* __restrict__ pb = reinterpret_cast<const float4*>(win + DCF_PLANE) + g;
    f32x2 xp[36];                // xp[q] = samples (2 q, 2 q + 1) of the window span 8 g .. 8 g + 71
#pragma unroll
    for (int h = 0; h < 9; ++h) {
        float4 a = pa[h], b = pb[h];
        if (h == 0) asm volatile("" : "+v"(a.x));     // sample 0 is never used: keeps the compiler from splitting the 16-byte reads
        xp[4 * h] = f32x2{a.x, a.y}; xp[4 * h + 1] = f32x2{a.z, a.w};
        xp[4 * h + 2] = f32x2{b.x, b.y}; xp[4 * h + 3] = f32x2{b.z, b.w};
    }
    taps63_pairs<2>(xp, w, acc);                      // the packed chain (zns_xqt.h): same taps, same order, same bits
}

__global__ void __launch_bounds__(256) decimate_chain_f32_kernel(DecChain P, TapPairs W) {
    extern __shared__ __attribute__((aligned(16))) float fwin[];      // [2][DCF_WIN] (one buffer when nlev == 1)
    const int clip = blockIdx.x / P.blocks_per_clip, b = blockIdx.x - clip * P.blocks_per_clip;
    auto T0 = [&](int j) { return 32 + (b * 4 * P.GL - 32) * (1 << (P.nlev - 1 - j)); };
    {
        const int o = 2 * T0(0) - 32, nch = 2 * P.G[0] + 18;          // chunks of the source window
        const float* x = P.src + (long long)clip * P.src_stride;
        for (int q = threadIdx.x; q < nch; q += 256) {
            const int i0 = o + 4 * q;
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (P.src_vec && i0 >= 0 && i0 + 3 < P.src_len) {
                f = *reinterpret_cast<const float4*>(x + i0);
            } else {
                if (i0 >= 0 && i0 < P.src_len) f.x = x[i0];
                if (i0 + 1 >= 0 && i0 + 1 < P.src_len) f.y = x[i0 + 1];
                if (i0 + 2 >= 0 && i0 + 2 < P.src_len) f.z = x[i0 + 2];
                if (i0 + 3 >= 0 && i0 + 3 < P.src_len) f.w = x[i0 + 3];
            }
            *reinterpret_cast<float4*>(fwin + (q & 1) * DCF_PLANE + (q >> 1) * 4) = f;
        }
    }
    __syncthreads();
    for (int j = 0; j < P.nlev; ++j) {
        const float* __restrict__ src = fwin + (j & 1) * DCF_WIN;
        float* __restrict__ nxt = fwin + ((j + 1) & 1) * DCF_WIN;
        const bool last = j + 1 == P.nlev;
        const int own_lo = (b * 4 * P.GL) << (P.nlev - 1 - j), own_hi = ((b + 1) * 4 * P.GL) << (P.nlev - 1 - j);
        float* __restrict__ y = P.dst[j] + (long long)clip * P.dst_stride[j];
        const int T0j = T0(j);
        for (int g = threadIdx.x; g < P.G[j]; g += 256) {
            float acc[4];
            dcf_taps63(src, g, W, acc);
            const int t0 = T0j + 4 * g;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = (t0 + k >= 0 && t0 + k < P.n_out[j]) ? dc_div_sqrt_half((double)acc[k]) : 0.f;   // y_hat /= sqrt(0.5), float32
            if (t0 >= own_lo && t0 < own_hi && (long long)t0 < P.dst_stride[j])
                *reinterpret_cast<float4*>(y + t0) = make_float4(v[0], v[1], v[2], v[3]);
            if (!last) *reinterpret_cast<float4*>(nxt + (g & 1) * DCF_PLANE + (g >> 1) * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
    }
}

struct XqtLevels {
    const float* y[XQ_OCT];
    long long stride[XQ_OCT];
    int vec[XQ_OCT];          // rows of this level are 16-byte aligned (base and stride)
    int len[XQ_OCT];
    int n_fft[XQ_OCT];
    int hop[XQ_OCT];
    int off[XQ_OCT];
};

// ---------------------------------------------------------------------------------------------------------
// Filter bank: V[k][t] = sum_n g_k[n] * ypad_i[t * hop_i + n] for the 12 filters of every octave, then log(|V| + 1e-9).
//
// Per octave this is a small dense GEMM  D[24 real rows][frames] = A[24][n_fft] * B[n_fft][frames]  (B is the strided
// view B[n][t] = ypad[t hop + n] of the octave's signal): it runs on the fp32 matrix cores as tiles of 32 frames, one
// v_mfma_f32_32x32x2_f32 per two taps.  fp32 MFMA has the fp32 FMA peak, so nothing is gained in rate (24 of the 32 rows
// are used) -- what it buys is operand delivery: the whole A operand (the octave's filter bank) sits in n_fft / 2
// registers per lane for as many tiles as the wave processes, and B costs ONE conflict-free ds_read_b32 per MFMA, where
// the previous per-thread FMA kernel spent its time streaming taps through the scalar cache and frames through LDS
// (16 % of the FMA peak).  The accumulation order per output is n = 0, 1, 2, ... exactly as a scalar fmaf chain (an fp32
// MFMA is bit for bit an fmaf chain), so results are identical to the kernel this replaces.
//
// Work split: a block owns (octave, clip, 4 * tpw consecutive 32-frame tiles) and each of its four waves a run of tpw
// tiles: a wave loads its octave's A operand once and walks its tiles; all waves of a block do the same amount of work
// and blocks are numbered octave-major by falling n_fft, so the long items (64 MFMAs per tile) are dispatched first and
// the 8-MFMA ones fill the tail.  (Round 2 first gave a block ALL octaves of a tile group, its waves sharing the
// (octave, tile) items along a host-built longest-processing-time schedule: 93 % balanced on paper, but a wave walked
// two or three octaves one exposed latency after the other -- 1.78 ms per 4096 clips against 1.69 ms, 39 us per 32 clips
// against 23 us.)  Every wave stages its tile's signal segment into its OWN LDS region (no workgroup barrier anywhere):
// element u of the segment lives at u + u / hopL (hopL = min(hop, n_fft): one pad word per frame step), so lane j's
// operand (frame j, tap c) sits at j (hopL + 1) + c + c / hopL -- an odd lane stride for every octave: conflict free.
constexpr int FB_TPW = 8;          // most tiles of 32 frames per wave (one for small batches: see zns_xqt_stages_f32)

struct FbRun {
    int oct, tile0, ntile;
};
struct FbArgs {
    XqtLevels L;
    const float* fira;
    int offa[XQ_OCT];
    int n_frames, tiles, tpw, blocks_per_clip, n_clips, wave_floats;
    unsigned char order[XQ_OCT];       // octaves by falling n_fft
    float* out;
};

// One staged tile (32 frames from t0, segment image in `lw`) times the octave's A operand, then the epilogue.  Shared by
// the instantiations of the per-octave launch (fb_run).
// LG >= 0: log2(min(hop, n_fft)) known at compile time -- every B operand read is then `base + immediate`; LG < 0: taken
// from `hop` at run time (the per-octave launch: one address per k-step, computed once per run of tiles).
template <int NFFT, int LG = -1, int BB = 4>
__device__ __forceinline__ void fb_multiply(float* __restrict__ out, int n_frames, int hop, int oct, int clip, int t0,
                                            const float* __restrict__ lw, int lane, const float (&a)[NFFT / 2]) {
    const int hopL = LG >= 0 ? (1 << (LG >= 0 ? LG : 0)) : (hop < NFFT ? hop : NFFT);
    const int lg = LG >= 0 ? LG : 31 - __builtin_clz(hopL);
    const int j = lane & 31, kk = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    const float* __restrict__ bp = lw + j * (hopL + 1) + kk;
    // B operands are read BB k-steps at a time and ONE BATCH AHEAD: the reads of batch i + 1 are issued before the MFMAs of
    // batch i, so the dependent MFMA chain (one every 64 cycles) does not wait for LDS (read-then-multiply per batch, or
    // the compiler's own pairing through two registers, left ~40 idle cycles per two MFMAs).  The scheduling fences keep
    // that order and keep the compiler from hoisting all n_fft / 2 reads at once (with the A operand and the prefetched
    // tile resident that would spill).  Two batches of 4 = the 8 registers the un-pipelined form used.
    float bv[2][BB];
#pragma unroll
    for (int u = 0; u < BB; ++u) bv[0][u] = bp[2 * u + ((2 * u) >> lg)];
#pragma unroll
    for (int s0 = 0; s0 < NFFT / 2; s0 += BB) {
        const int cur = (s0 / BB) & 1;
        if (s0 + BB < NFFT / 2) {
#pragma unroll
            for (int u = 0; u < BB; ++u) bv[cur ^ 1][u] = bp[2 * (s0 + BB + u) + ((2 * (s0 + BB + u)) >> lg)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < BB; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s0 + u], bv[cur][u], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- rows 2k / 2k+1 of one filter are neighbouring registers of one lane: |V|, log, coalesced along time
    const int t = t0 + j;
    if (t < n_frames) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int k = (q & 1) + 4 * (q >> 1) + 2 * kk;                       // row = (reg & 3) + 8 (reg >> 2) + 4 kk, reg = 2q
            const float re = acc[2 * q], im = acc[2 * q + 1];
            const int kbin = XQ_BINS - XQ_BPO * (oct + 1) + k;
            out[((size_t)clip * XQ_BINS + kbin) * n_frames + t] = xqt_log_magnitude(re, im);
        }
    }
}

// LG >= 0: log2(min(hop, n_fft)) as a compile-time constant (the octaves of the VQT bank): the tile image's size and every
// per-piece guard are then constants, the prefetch is straight-line code and the B operand reads are `base + immediate`.
// LG < 0: taken from `hop` at run time (the CQT bank: 256 taps in every octave).
template <int NFFT, int LG = -1>
__device__ __forceinline__ void fb_run(const FbArgs& P, const FbRun r, int clip, float* __restrict__ lw, int lane) {
    const int oct = r.oct;
    const int hop_rt = P.L.hop[oct], len = P.L.len[oct];
    constexpr int LGC = LG >= 0 ? LG : 0;
    // hop itself is only a constant when it is below n_fft (segment layout); in the windows layout it stays run time
    const int hop = (LG >= 0 && (1 << LGC) < NFFT) ? (1 << LGC) : hop_rt;
    const int hopL = LG >= 0 ? (1 << LGC) : (hop < NFFT ? hop : NFFT);
    const int lg = LG >= 0 ? LG : 31 - __builtin_clz(hopL);
    const float* __restrict__ y = P.L.y[oct] + (long long)clip * P.L.stride[oct];
    float a[NFFT / 2];
    const float* __restrict__ ga = P.fira + P.offa[oct] + lane;
#pragma unroll
    for (int s = 0; s < NFFT / 2; ++s) a[s] = ga[s * 64];
    const int j = lane & 31, kk = lane >> 5;
    // Tile = frames t0 .. t0+31 with librosa's center=True reflect padding of n_fft / 2 at both ends of the signal.
    // Element e of a tile: windows layout (hop >= n_fft) e = f * n_fft + c, else the contiguous segment e = u; at most
    // 32 * n_fft elements = NV float4 per lane.  An interior tile travels global -> registers -> LDS, 16 bytes per lane,
    // and its registers are loaded while the PREVIOUS tile is being multiplied (one memory latency per run of tiles is
    // exposed, not one per tile); the two tiles at the ends of the signal are staged element by element.
    constexpr int NV = NFFT / 8;
    const bool win = LG >= 0 ? (1 << LGC) == NFFT : hop >= NFFT;
    const int nel = win ? 32 * NFFT : 31 * hop + NFFT;
    const bool vec_ok = (hop & 3) == 0 && P.L.vec[oct];
    auto dst = [&](int e) { return win ? (e / NFFT) * (NFFT + 1) + (e & (NFFT - 1)) : e + (e >> lg); };
    float4 v[NV];
    auto interior = [&](int t0) {      // the tile touches neither end of the signal and can move 16 bytes per lane
        return vec_ok && t0 * hop - NFFT / 2 >= 0 && (t0 + 31) * hop + NFFT / 2 <= len;
    };
    // lane's float4 number u is element e = 256 u + 4 lane: both its source and its LDS position are (lane part) + u * (a
    // wave-uniform stride), because n_fft and hopL divide 256 -- one address per lane, immediates / scalars for the rest
    const int e0 = 4 * lane;
    const int sstride = win ? (256 / NFFT) * hop : 256;                        // source floats between consecutive u
    const int dstride = win ? (256 / NFFT) * (NFFT + 1) : 256 + (256 >> lg);   // LDS floats between consecutive u
    const int soff0 = win ? (e0 / NFFT) * hop + (e0 & (NFFT - 1)) : e0;
    float* __restrict__ lp = lw + dst(e0);
    auto load_tile = [&](int t0) {     // interior tiles only: global -> registers
        const float* __restrict__ sp = y + (t0 * hop - NFFT / 2 + soff0);
#pragma unroll
        for (int u = 0; u < NV; ++u)
            if (256 * u + e0 < nel) v[u] = *reinterpret_cast<const float4*>(sp + u * sstride);
    };
    auto store_tile = [&]() {          // registers -> LDS (hopL is a multiple of 4 here: the 4 floats stay adjacent)
#pragma unroll
        for (int u = 0; u < NV; ++u)
            if (256 * u + e0 < nel) {
                float* d = lp + u * dstride;
                d[0] = v[u].x;
                d[1] = v[u].y;
                d[2] = v[u].z;
                d[3] = v[u].w;
            }
    };
    auto stage_edge = [&](int t0) {    // tiles at the ends of the signal (and unaligned rows): element-wise, reflected
        const int m_lo = t0 * hop - NFFT / 2;
        for (int base = 0; base < nel; base += 64 * 8) {
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * 64 + lane;
                int m = win ? (t0 + e / NFFT) * hop + (e & (NFFT - 1)) - NFFT / 2 : m_lo + e;
                m = m < 0 ? -m : m;
                m = m >= len ? 2 * (len - 1) - m : m;
                m = m < 0 ? 0 : (m >= len ? len - 1 : m);                               // frames past the last one are never written out
                w[u] = y[m];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * 64 + lane;
                if (e < nel) lw[dst(e)] = w[u];
            }
        }
    };
    const int first = r.tile0;
    if (first >= P.tiles) return;
    // first tile of the run: staged synchronously
    if (interior(first * 32)) { load_tile(first * 32); store_tile(); } else stage_edge(first * 32);
    // ... and the A operand is KNOWN to have landed before the loop (consuming its registers puts the wait here, once).
    // Its loads are older than every tile prefetch; left pending into the loop, the first MFMA of EVERY tile carried a
    // vmcnt wait that also covered (part of) the prefetch just issued: the memory latency exposed per tile, the matrix
    // pipe 43 % busy (round 2).
#pragma unroll
    for (int s0 = 0; s0 < NFFT / 2; s0 += 8)
        asm volatile("" ::"v"(a[s0]), "v"(a[s0 + 1]), "v"(a[s0 + 2]), "v"(a[s0 + 3]), "v"(a[s0 + 4]), "v"(a[s0 + 5]),
                     "v"(a[s0 + 6]), "v"(a[s0 + 7]));
    for (int ti = 0; ti < r.ntile; ++ti) {
        const int tile = first + ti;
        if (tile >= P.tiles) break;
        const int t0 = tile * 32;
        // the NEXT tile: requested now, in flight while this tile is multiplied, written to LDS after the multiply -- the
        // loaded values live inside one iteration (carried across the loop edge, two of the 64 registers were copied right
        // behind the loads: another wait for the whole prefetch in front of the first MFMA)
        const bool more = ti + 1 < r.ntile && tile + 1 < P.tiles;
        const bool pre = more && interior(t0 + 32);
        if (pre) load_tile(t0 + 32);
        // same-wave LDS traffic is ordered: no barrier between staging writes and operand reads
        fb_multiply<NFFT, LG>(P.out, P.n_frames, hop, oct, clip, t0, lw, lane, a);
        if (pre) store_tile(); else if (more) stage_edge(t0 + 32);
    }
}

// MAXFFT bounds the instantiated octave sizes (VQT: 128, CQT: 256): the A operand's registers set the kernel's VGPR
// budget, and the VQT kernel should keep two workgroups per CU so that one wave stages while another multiplies.
template <int MAXFFT>
__global__ void __launch_bounds__(256, MAXFFT <= 128 ? 2 : 1) xqt_filter_kernel(FbArgs P) {
    extern __shared__ __attribute__((aligned(16))) float fbl[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* lw = fbl + wid * P.wave_floats;
    int bid = blockIdx.x;
    const int group = bid % P.blocks_per_clip; bid /= P.blocks_per_clip;
    const int clip = bid % P.n_clips;
    const FbRun r{P.order[bid / P.n_clips], (4 * group + wid) * P.tpw, P.tpw};
    const int nf = P.L.n_fft[r.oct], hp = P.L.hop[r.oct];
    const int key = nf * 16 + (31 - __builtin_clz(hp < nf ? hp : nf));       // (n_fft, log2 of the staging stride)
    switch (key) {
        // the VQT bank (octaves 0..7: n_fft 128,128,128,128,64,32,32,16; hop 256 >> octave)
        case 128 * 16 + 7: fb_run<128, 7>(P, r, clip, lw, lane); break;
        case 128 * 16 + 6: fb_run<128, 6>(P, r, clip, lw, lane); break;
        case 128 * 16 + 5: fb_run<128, 5>(P, r, clip, lw, lane); break;
        case 64 * 16 + 4: fb_run<64, 4>(P, r, clip, lw, lane); break;
        case 32 * 16 + 3: fb_run<32, 3>(P, r, clip, lw, lane); break;
        case 32 * 16 + 2: fb_run<32, 2>(P, r, clip, lw, lane); break;
        case 16 * 16 + 1: fb_run<16, 1>(P, r, clip, lw, lane); break;
        default:
            // anything else (the CQT bank) through the run-time form
            switch (nf) {
                case 16: fb_run<16>(P, r, clip, lw, lane); break;
                case 32: fb_run<32>(P, r, clip, lw, lane); break;
                case 64: fb_run<64>(P, r, clip, lw, lane); break;
                case 128: fb_run<128>(P, r, clip, lw, lane); break;
                default:
                    if constexpr (MAXFFT > 128) fb_run<256>(P, r, clip, lw, lane);
                    break;
            }
            break;
    }
}


}  // namespace zns

using namespace zns;

extern "C" {

int zns_selftest_div_sqrt_half(unsigned long long* mismatches, unsigned* first_bad_bits, void* stream) {
    ZNS_CHECK_ARG(mismatches != nullptr && first_bad_bits != nullptr, "selftest: null device pointers");
    hipStream_t st = (hipStream_t)stream;
    ZNS_HIP(hipMemsetAsync(mismatches, 0, sizeof(unsigned long long), st));
    ZNS_HIP(hipMemsetAsync(first_bad_bits, 0xFF, sizeof(unsigned), st));
    hipLaunchKernelGGL(selftest_div_kernel, dim3(4096), dim3(256), 0, st, mismatches, first_bad_bits);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

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

int zns_xqt_debug_stamps(unsigned long long* host_out, size_t n) {
    unsigned long long* d = xqt_stream_stamps();
    ZNS_CHECK_ARG(d != nullptr && host_out != nullptr && n <= 32 * 12 * 8, "xqt_debug_stamps: no stamped launch (zns_test_set_tuning: XQT_STAMPS)");
    ZNS_HIP(hipDeviceSynchronize());
    ZNS_HIP(hipMemcpy(host_out, d, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
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
    return zns_xqt_stages_f32(wav, n_clips, n_samples, sample_rate, mode, out, ws, ws_bytes, 3, stream);
}

int zns_xqt_stages_f32(const float* wav, int n_clips, int n_samples, int sample_rate, int mode, float* out, void* ws,
                       size_t ws_bytes, int stages, void* stream) {
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
        L.vec[i] = 0;
        if (i > 0) {
            L.y[i] = pyr + (size_t)n_clips * off[i];
            L.stride[i] = (long long)(((size_t)len[i] + 3) / 4 * 4);
        }
    }
    const bool exact = (stages & ZNS_XQT_EXACT) != 0;     // the float64-triple pyramid (validation implementation)
    const int plan_flags = stages & ~(3 | ZNS_XQT_EXACT);
    stages &= 3;
    ZNS_CHECK_ARG(stages >= 1 && (plan_flags & ~(ZNS_XQT_SEPARATE | ZNS_XQT_FUSED)) == 0 &&
                  plan_flags != (ZNS_XQT_SEPARATE | ZNS_XQT_FUSED),
                  "xqt: stages=%d (1 pyramid, 2 filter bank, 3 both; + ZNS_XQT_SEPARATE or ZNS_XQT_FUSED; + ZNS_XQT_EXACT)",
                  stages | plan_flags);
    // Both stages of the default arithmetic: the fused kernel (zns_xqt_stream.hip: one launch, the pyramid never leaves LDS,
    // HBM traffic = the clip in + the spectrogram out), unless the caller asks for the per-stage launches.  Same arithmetic
    // in the same order: the same bits (tests/test_vqt.py).  The cqt bank (256 taps in every octave) and the exact pyramid
    // always take the per-stage launches.
    if (stages == 3 && !(plan_flags & ZNS_XQT_SEPARATE)) {
        const bool can = !exact && xqt_stream_supported(b, n_clips, n_samples);
        ZNS_CHECK_ARG(can || !(plan_flags & ZNS_XQT_FUSED), "xqt: the fused kernel is built for the vqt bank and the default pyramid arithmetic");
        // small batches cannot fill the fused kernel's pipeline (two warm-up steps and nine steps of systolic lag around an
        // item's five): one 10 s clip 37 us against 22 us for the three chained pyramid launches + the filter bank, 32 clips
        // 53 against 45, 128 clips 114 against 106, 256 clips 195 against 218, 4096 clips 2.73 against 3.26 ms
        // (tools/bench_xqt_sizes.py, profiles/r04_xqt_bench_sizes.json)
        if (can && ((plan_flags & ZNS_XQT_FUSED) || n_clips >= 192)) return xqt_stream_launch(b, wav, n_clips, n_samples, out, st);
    }
    // decimation pyramid: levels 1..7, grouped into chains (one launch each).  Default plan: the two big levels on
    // their own (no recomputed halo where 3/4 of the taps are), then chains of two and three (the small levels' launches
    // are latency-bound: 6.5 us each for a 32-clip batch).  ZNS_TUNE_XQT_PLAN overrides the grouping (tuning aid, include/zns_test.h).
    if (stages & 1) {
        // big batches: one launch per level (the fused tail recomputes halos and runs few blocks); small ones: chains of
        // 2 + 2 + 3 levels.  tools/exp_xqt_plan_sweep.py, 32 clips, whole transform with the float32 pyramid: 49.6 us against
        // 51.8 for {1,1,2,3} (the exact pyramid's best: its taps cost six times as much, halos weigh more), 50.8 for {3,4},
        // 59.5 for one launch per level -- at this size a launch is mostly latency and there are three instead of four
        int plan[XQ_OCT] = {2, 2, 3, 1, 1, 1, 1}, nplan = 3;
        if (exact) { plan[0] = 1; plan[1] = 1; plan[2] = 2; plan[3] = 3; nplan = 4; }
        if (n_clips >= 256) { nplan = 7; for (int i = 0; i < 7; ++i) plan[i] = 1; }
        if (const int tp = tuning(ZNS_TUNE_XQT_PLAN)) {      // decimal digits, first chain first: 1123 = {1, 1, 2, 3}
            int digits[XQ_OCT], nd = 0, tot = 0;
            for (int q = tp; q > 0 && nd < XQ_OCT; q /= 10) digits[nd++] = q % 10;
            nplan = 0;
            for (int i = nd - 1; i >= 0; --i) {
                const int v = digits[i];
                ZNS_CHECK_ARG(v >= 1 && v <= DC_MAXLEV, "ZNS_TUNE_XQT_PLAN: chain of %d levels (1..%d)", v, DC_MAXLEV);
                plan[nplan++] = v;
                tot += v;
            }
            ZNS_CHECK_ARG(tot == XQ_OCT - 1, "ZNS_TUNE_XQT_PLAN must cover the %d decimation levels", XQ_OCT - 1);
        }
        static const int gl_max[DC_MAXLEV + 1] = {0, 256, 119, 50, 16};   // G[0] = (GL + 18) 2^(nlev-1) - 18 <= DC_S - 9
        int a = 0;   // source level of the next chain
        for (int c = 0; c < nplan; ++c) {
            const int nlev = plan[c];
            DecChain P{};
            P.src = L.y[a];
            P.src_stride = L.stride[a];
            P.src_len = len[a];
            P.src_vec = (L.stride[a] % 4 == 0) && ((uintptr_t)L.y[a] % 16 == 0);
            P.nlev = nlev;
            P.GL = gl_max[nlev];
            P.blocks_per_clip = (int)((L.stride[a + nlev] + 4 * P.GL - 1) / (4 * P.GL));
            for (int j = 0; j < nlev; ++j) {
                P.dst[j] = (float*)L.y[a + 1 + j];
                P.dst_stride[j] = L.stride[a + 1 + j];
                P.n_out[j] = len[a + j] / 2;   // resampy: int(n * 0.5); librosa pads to ceil with zeros
            }
            P.G[nlev - 1] = P.GL;
            for (int j = nlev - 2; j >= 0; --j) P.G[j] = 2 * P.G[j + 1] + 18;
            const dim3 grid((unsigned)((size_t)n_clips * P.blocks_per_clip));
            if (exact) {
                size_t lds = (size_t)(nlev > 1 ? 2 : 1) * DC_WIN * sizeof(double);
                if (const int pad_kib = tuning(ZNS_TUNE_XQT_LDS_PAD)) {      // tuning aid: cap the chain kernel's occupancy through its LDS size
                    lds = std::max(lds, (size_t)pad_kib * 1024);
                    static bool set[ZNS_MAX_DEVICES] = {};
                    if (!set[current_device_slot()]) {
                        ZNS_HIP(hipFuncSetAttribute((const void*)decimate_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                        set[current_device_slot()] = true;
                    }
                }
                hipLaunchKernelGGL(decimate_chain_kernel, grid, dim3(256), lds, st, P, b->taps);
            } else {
                TapPairs W;
                for (int i = 0; i < 30; ++i) W.p[i] = f32x2{b->tapsf[i], b->tapsf[i + 2]};
                const size_t lds = (size_t)(nlev > 1 ? 2 : 1) * DCF_WIN * sizeof(float);
                hipLaunchKernelGGL(decimate_chain_f32_kernel, grid, dim3(256), lds, st, P, W);
            }
            ZNS_LAUNCH_CHECK();
            a += nlev;
        }
    }
    if (!(stages & 2)) return ZNS_OK;
    for (int i = 0; i < XQ_OCT; ++i) L.vec[i] = (L.stride[i] % 4 == 0) && ((uintptr_t)L.y[i] % 16 == 0);
    const int n_frames = zns_xqt_frames(n_samples);
    FbArgs F{};
    F.L = L;
    F.fira = b->fira;
    for (int i = 0; i < XQ_OCT; ++i) F.offa[i] = b->offa[i];
    F.n_frames = n_frames;
    F.tiles = (n_frames + 31) / 32;
    // tiles per wave: a quarter of the clip's tiles (the A operand is loaded once per wave); batches under 64 clips take
    // single tiles so that the launch still covers the chip (32 clips: 1280 blocks instead of 256).  Measured, kernel
    // time of the filter bank, tpw = 1 / 5: 32 clips 23 / 29 us, 64 clips 41 / 35 us, 4096 clips 2.22 / 1.69 ms.
    F.tpw = n_clips < 64 ? 1 : std::max(1, std::min(FB_TPW, (F.tiles + 3) / 4));
    if (const int tpw = tuning(ZNS_TUNE_XQT_TPW)) F.tpw = std::max(1, std::min(FB_TPW, tpw));   // tuning aid
    F.blocks_per_clip = (F.tiles + 4 * F.tpw - 1) / (4 * F.tpw);
    F.n_clips = n_clips;
    {
        int ord[XQ_OCT];
        for (int i = 0; i < XQ_OCT; ++i) ord[i] = i;
        std::stable_sort(ord, ord + XQ_OCT, [&](int x, int y) { return b->n_fft[x] > b->n_fft[y]; });
        for (int i = 0; i < XQ_OCT; ++i) F.order[i] = (unsigned char)ord[i];
    }
    F.out = out;
    int wf = 0;   // floats of one wave's staging region: the largest segment image over the octaves
    for (int i = 0; i < XQ_OCT; ++i) {
        const int nf = b->n_fft[i], hp = b->hop[i];
        const int need = hp >= nf ? 32 * (nf + 1) : (31 * hp + nf) + (31 * hp + nf) / hp + 1;
        wf = std::max(wf, need);
    }
    F.wave_floats = (wf + 3) / 4 * 4;
    const size_t lds = (size_t)4 * F.wave_floats * sizeof(float);
    int max_fft = 0;
    for (int i = 0; i < XQ_OCT; ++i) max_fft = std::max(max_fft, b->n_fft[i]);
    static bool lds_set[2][ZNS_MAX_DEVICES] = {};
    const int big = max_fft > 128 ? 1 : 0;
    using Kern = void (*)(FbArgs);
    static const Kern kerns[2] = {xqt_filter_kernel<128>, xqt_filter_kernel<256>};
    if (!lds_set[big][current_device_slot()]) {
        ZNS_HIP(hipFuncSetAttribute((const void*)kerns[big], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lds_set[big][current_device_slot()] = true;
    }
    const dim3 grid((unsigned)((size_t)XQ_OCT * n_clips * F.blocks_per_clip));
    hipLaunchKernelGGL(kerns[big], grid, dim3(256), lds, st, F);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

}  // extern "C"
