// generate_XQT on gfx950: log(|VQT or CQT| + 1e-9), 96 bins, hop 256, 16 kHz  (input_rep.py:11-57).
//
// The reference delegates to librosa 0.8.1 (vqt/cqt -> per-octave FFT filter bank + resampy kaiser_fast
// decimation).  Here the per-octave "rfft -> sparse basis product" is folded on the host (float64) into an
// equivalent strided complex FIR bank g[k][n] (<= 12 x 256 taps per octave).  Two stages:
//   * the octave pyramid (seven 2:1 decimations).  resampy rounds its float64 product-sum to float32 after EVERY tap and
//     parity needs exactly that, so a tap is three dependent half-rate instructions (v_fma_f64, v_cvt_f32_f64,
//     v_cvt_f64_f32): the stage is bound by float64 instruction issue (12.7e12 taps/s per chip), not by HBM;
//   * the filter bank as per-octave GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, A operand in registers).
// The default plan runs them as separate launches (decimate_chain_kernel per level or fused tail, xqt_filter_kernel).
// xqt_march_kernel is a second implementation of the whole transform -- ONE persistent kernel that marches through each
// clip in time order with pyramid waves and filter-bank waves side by side in every workgroup, no halo recomputation and
// less HBM traffic (8-11 GB against 13 GB per 4096 clips); bit-identical, selected with ZNS_XQT_MARCH.  It was built to overlap the two stages and
// measured: an fp32 MFMA and a vector instruction of one SIMD do not execute concurrently on gfx950, so it is not faster.
//
// Numerics that are restated exactly (see oracle/vqt_ref.py, which documents the recalled third-party
// algorithm; PARITY UNPINNED there): complex64 filter storage, complex64 basis after sparsify_rows(1%),
// sqrt(2^i) rescale, resampy's tap-by-tap float32 accumulation with float64 products, float32 /sqrt(0.5).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <complex>
#include <mutex>
#include <vector>

#include "zns_common.h"
#include "zns_kernels.h"

namespace zns {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
    float* fira = nullptr;  // device: MFMA A operand of the filter bank, [octave][n_fft / 2 k-steps][64 lanes]
    float* firat = nullptr; // the same operand lane-major, [octave][64 lanes][n_fft / 2 k-steps]: a lane's whole operand is
                            // one contiguous run (the marching kernel reloads it per tile with 16-byte loads off ONE pointer)
    int offa[XQ_OCT];       // float offset of octave i inside `fira` / `firat`
    DecTaps* taps = nullptr;   // device: float64 decimation taps
    bool ready = false;
};
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
        std::vector<float> hat(tota, 0.f);
        for (int i = 0; i < XQ_OCT; ++i) {
            const int ns = b.n_fft[i] / 2;
            for (int sidx = 0; sidx < ns; ++sidx)
                for (int l = 0; l < 64; ++l) hat[b.offa[i] + (size_t)l * ns + sidx] = ha[b.offa[i] + (size_t)sidx * 64 + l];
        }
        ZNS_HIP(hipMalloc(&b.firat, tota * sizeof(float)));
        ZNS_HIP(hipMemcpy(b.firat, hat.data(), tota * sizeof(float), hipMemcpyHostToDevice));
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

// resampy's `y_hat /= sqrt(0.5)`-in-float32 (librosa.resample: y_hat /= np.sqrt(ratio), float32 array / float32 scalar):
// the correctly rounded float32 quotient x / c, c = float32(sqrt(0.5)).  v_div_* is a ten-instruction sequence per
// output; instead: x * RN64(1 / c) in float64, rounded once to float32.  That IS the correctly rounded quotient: the
// float64 product is within 2^-52 (relative) of x / c, while a quotient of two 24-bit significands, c not a power of
// two, is never closer than 2^-48 to a float32 rounding boundary (|X 2^k - (2M + 1) C| >= 1 with C < 2^24).  Checked
// exhaustively over all 2^32 float32 inputs on the device: zns_selftest_div_sqrt_half (tests/test_vqt.py).
constexpr float DC_SQRT_HALF = 0.70710677f;
constexpr double DC_INV_SQRT_HALF = 1.0 / (double)DC_SQRT_HALF;
__device__ __forceinline__ float dc_div_sqrt_half(double x_float32_valued) {
    return (float)(x_float32_valued * DC_INV_SQRT_HALF);
}

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

// The whole 63-tap accumulation of one group (4 outputs), software pipelined: the 14 window samples of tap chunk c + 1 are
// in flight while chunk c is multiplied, so a wave never waits for LDS inside the chain (at two waves per SIMD, the
// marching kernel's share, the serial form ran 1.4x slower than at eight).  The taps stay in scalar registers for the whole
// kernel: resampy's right wing is its left wing shifted by one (r[k] = win(256 (k + 1)) = l[k + 1]), so 32 values cover
// all 63 taps -- and with no scalar loads in flight the LDS reads can be waited for by count.
// `pg` = window base + group index (see dc_chunk for the sample layout); order of accumulation: left 0..31, right 0..30.
template <int RS>
__device__ __forceinline__ void dc_window(const double* __restrict__ p, double (&xv)[14]) {
#pragma unroll
    for (int n = 0; n < 14; ++n) xv[n] = p[((1 + n) & 7) * RS + ((1 + n) >> 3)];
}
template <int NT, bool LEFT, int W0>
__device__ __forceinline__ void dc_mac(const double (&xv)[14], const double (&tl)[32], double (&acc)[4]) {
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = (double)(float)(acc[k] + tl[W0 + u] * xv[LEFT ? 7 + 2 * k - u : 2 * k + u]);
}
template <int RS>
__device__ __forceinline__ void dc_taps63(const double* __restrict__ pg, const double (&tl)[32], double (&acc)[4]) {
    double xa[14], xb[14];
    dc_window<RS>(pg + 3, xa);
    __builtin_amdgcn_sched_barrier(0);
    dc_window<RS>(pg + 2, xb);
    __builtin_amdgcn_sched_barrier(0);
    dc_mac<8, true, 0>(xa, tl, acc);
    __builtin_amdgcn_sched_barrier(0);
    dc_window<RS>(pg + 1, xa);
    __builtin_amdgcn_sched_barrier(0);
    dc_mac<8, true, 8>(xb, tl, acc);
    __builtin_amdgcn_sched_barrier(0);
    dc_window<RS>(pg, xb);
    __builtin_amdgcn_sched_barrier(0);
    dc_mac<8, true, 16>(xa, tl, acc);
    __builtin_amdgcn_sched_barrier(0);
    dc_window<RS>(pg + 4, xa);
    __builtin_amdgcn_sched_barrier(0);
    dc_mac<8, true, 24>(xb, tl, acc);
    __builtin_amdgcn_sched_barrier(0);
    dc_window<RS>(pg + 5, xb);
    __builtin_amdgcn_sched_barrier(0);
    dc_mac<8, false, 1>(xa, tl, acc);          // right wing: r[k] = l[k + 1]
    __builtin_amdgcn_sched_barrier(0);
    dc_window<RS>(pg + 6, xa);
    __builtin_amdgcn_sched_barrier(0);
    dc_mac<8, false, 9>(xb, tl, acc);
    __builtin_amdgcn_sched_barrier(0);
    dc_window<RS>(pg + 7, xb);
    __builtin_amdgcn_sched_barrier(0);
    dc_mac<8, false, 17>(xa, tl, acc);
    __builtin_amdgcn_sched_barrier(0);
    dc_mac<7, false, 25>(xb, tl, acc);
    // The callers keep or zero each result by its output index.  Without this the optimiser sinks every output's 63-tap
    // chain into its own conditional block: four serial chains one after the other (no instruction-level parallelism
    // between them) behind ALL 69 window loads (225 registers).  The empty asm makes the four sums unconditionally live here.
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
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
            // serial chunks of 8 taps (taps through the scalar cache): 48 registers, 8 waves per SIMD hide every wait.  The
            // pipelined form the marching kernel needs (dc_taps63: 83 registers, 5 waves) measured 9 % slower here.
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
// the per-octave launch (fb_run) and the marching kernel, so both produce the same bits.
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
    float bv[2][BB];                // BB = 4 in the per-octave launch; the marching kernel (168 registers) takes 2
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
            // hardware sqrt / log2 (1 ulp each): 2e-6 absolute in the log domain at most, a fiftieth of the parity
            // tolerance, and a fifth of the instructions of the correctly rounded library forms -- the epilogue is
            // per-tile work that the 8- and 16-MFMA octaves cannot hide
            out[((size_t)clip * XQ_BINS + kbin) * n_frames + t] =
                __logf(__builtin_amdgcn_sqrtf(re * re + im * im) + 1e-9f);
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


// ---------------------------------------------------------------------------------------------------------
// xqt_march_kernel: pyramid and filter bank of a clip in ONE persistent workgroup that marches through the clip in time
// order.  768 threads: waves 0..7 are the pyramid (float64 vector pipes), waves 8..11 the filter bank (matrix pipes),
// one filter wave and two pyramid waves per SIMD, ONE barrier per step.
//
// A step advances the level-0 signal by MQ_S0 = 2048 samples (8 frames).  Level j = 1..7 then owes S_j = 2048 >> j new
// outputs: 512 + 256 + ... + 8 = 1016 groups of four over 2 x 508 ... no: S_j / 4 = 256 + 128 + ... + 4 = 508 groups on 512
// pyramid threads -- EVERY level advances in EVERY step, each one step behind the level that feeds it (a systolic
// pipeline: level j works on stream step s - j), so nothing is recomputed (the chained launches recompute a 72-sample
// halo per block and level) and all eight pyramid waves carry the same load.  Level j's outputs go to the pyramid rows in
// global memory (float32, for the filter bank) and, as float64, into the window level j + 1 reads in the next step:
// windows are double buffered [8 rows][columns] images in the layout decimate_chain_kernel uses (same dc_chunk, same
// taps, same rounding: the same bits), `h_j` samples of history slid from the previous window (zeros at a clip start).
// Output t of level j at its local step u: t in [u S_j - D_j, (u + 1) S_j - D_j), D = {0,16,24,28,32,32,32,32}: the few
// outputs a level runs behind so that the 31 samples of look-ahead it needs already exist.
//
// Filter bank: wave 8 + w owns octaves w (128 taps) and 7 - w; a 32-frame tile of octave i is complete every fourth step.
// Its segment image is staged into the wave's LDS region a quarter per step as the samples appear (global -> registers at
// the start of a step, registers -> LDS at its end, so no load latency is exposed and a quarter costs <= 16 registers),
// and multiplied (fb_multiply, A operand reloaded from L2) in the step after the last quarter -- while quarter 0 of the
// next tile is already in flight.  Pyramid rows are read back through L2 a few steps after they were written.
// A workgroup takes clips b, b + G, b + 2G, ... back to back as ONE stream: the pipeline drains once per launch.
constexpr int MQ_S0 = 2048;
constexpr int MQ_NPYR = 512, MQ_NT = 768;
constexpr int MQ_LEV = XQ_OCT - 1;                  // decimation levels
__host__ __device__ constexpr int mq_ncol(int j) { return (256 >> j) + 9; }           // (S_j + 72) / 8 columns per window
__host__ __device__ constexpr int mq_col0(int j) { return 1024 - (1024 >> j) + 18 * j; }   // 2 * sum_{i<j} mq_ncol(i)
constexpr int MQ_COLS = mq_col0(MQ_LEV);            // levels 0..6 feed a next level: two windows each
constexpr int MQ_ROW = MQ_COLS + 1;                 // odd row stride
constexpr int MQ_F64 = 8 * MQ_ROW;
__host__ __device__ constexpr int mq_D(int j) { return j == 0 ? 0 : j == 1 ? 16 : j == 2 ? 24 : j == 3 ? 28 : 32; }
__host__ __device__ constexpr int mq_h(int j) { return 2 * mq_D(j + 1) + 32 - mq_D(j); }   // history samples of level j's window
static_assert(mq_col0(1) == 2 * mq_ncol(0) && mq_col0(3) == 2 * (mq_ncol(0) + mq_ncol(1) + mq_ncol(2)), "window columns");
static_assert(mq_h(0) <= 72 && mq_h(3) <= 72 && mq_h(6) <= 72 && mq_h(3) >= 62 && mq_h(0) >= 62, "window history");
static_assert(2 * mq_D(1) >= mq_D(0) + 30 && 2 * mq_D(2) >= mq_D(1) + 30 && 2 * mq_D(3) >= mq_D(2) + 30 &&
              2 * mq_D(4) >= mq_D(3) + 30 && 2 * mq_D(5) >= mq_D(4) + 30 && 2 * mq_D(7) >= mq_D(6) + 30, "look-ahead");

struct MqSlot {
    int oct, nfft, region;     // octave, its n_fft, float offset of the staging region inside the filter part of LDS
    int rb;                    // quarter k of tile q is loaded at clip-relative step 4 q + k + rb
    int ub[5];                 // quarter k moves float4 numbers ub[k] .. ub[k+1] - 1 of the tile image
};
struct MqArgs {
    FbArgs F;                  // levels, A operand, frames, out (tpw / blocks_per_clip / order unused)
    const DecTaps* taps;
    int n;                     // level-0 samples per clip
    int K;                     // steps per clip
    int n_clips, G;            // clips, persistent workgroups
    int tail;                  // steps after a workgroup's last clip (pipeline drain)
    int nt4;                   // 4 * tiles per clip
    int dbg;                   // tuning aid (ZNS_XQT_DBG): 1 = skip the pyramid's taps, 2 = skip the filter bank's multiplies, 4 = skip its loads
    int n_out[XQ_OCT];         // resampy's output count of level j (level j = octave j's signal)
    MqSlot slot[4][2];
};

// One (filter wave, octave) slot with everything the step loop needs in registers (wave-uniform): no indexed access to
// the kernel arguments inside the loop.
struct MqOct {
    const float* y;            // the octave's pyramid rows
    long long stride;
    const float* ga;           // its A operand
    int len, hop, vec, oct, nfft, lgn, lgh, win, rb;
    int lsrc, sstride;         // interior tiles: lane's source offset inside the tile image, floats between consecutive u
    int ldst, dstride, c1, c2, c3;   // LDS position of the lane's element 4 lane (+ u dstride), offsets of the next three floats
    int ub0, ub1, ub2, ub3, ub4;
    int rel, c;                // cursor: step (a multiple of 4) inside clip c's schedule at which the next tile starts
    int valid, clip, t0;       // the tile whose quarters are being staged
};
__device__ __forceinline__ int mq_reflect(int m, int len) {
    m = m < 0 ? -m : m;
    m = m >= len ? 2 * (len - 1) - m : m;
    return m < 0 ? 0 : (m >= len ? len - 1 : m);      // frames past the last one are never written out
}

// Quarter of a tile image: float4 numbers [u_lo, u_hi) (element e = 256 u + 4 lane) global -> registers.  Source and LDS
// position of element e are both (a per-lane constant) + u * (a wave-uniform stride) [+ c for the four floats]: n_fft and
// the staging stride divide 256.  The per-lane parts are computed once per kernel (MqOct::lsrc, ldst).
template <int NV>
__device__ __forceinline__ void mq_fb_load(const MqOct& O, int clip, int t0, int u_lo, int u_hi, int lane, float4 (&v)[NV]) {
    const int nfft = O.nfft, hop = O.hop, len = O.len;
    const float* __restrict__ y = O.y + (long long)clip * O.stride;
    const bool interior = (hop & 3) == 0 && O.vec && t0 * hop - nfft / 2 >= 0 && (t0 + 31) * hop + nfft / 2 <= len;
    if (interior) {
        int lsrc = O.lsrc;
        asm volatile("" : "+v"(lsrc));     // keeps the (loop-invariant) per-piece addresses from being hoisted out of the step
                                           // loop, where they would fill -- and spill -- the register file
        const float* __restrict__ sp = y + (t0 * hop - nfft / 2) + lsrc;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = u_lo + i;
            if (u < u_hi) v[i] = *reinterpret_cast<const float4*>(sp + u * O.sstride);
        }
    } else {
        int ln = lane;
        asm volatile("" : "+v"(ln));       // as above: the index arithmetic of the (rare) edge tiles stays inside the step
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = u_lo + i;
            if (u < u_hi) {
                const int e = 256 * u + 4 * ln;
                float w[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ee = e + c;
                    const int m = (O.win ? (t0 + (ee >> O.lgn)) * hop + (ee & (nfft - 1)) : t0 * hop + ee) - nfft / 2;
                    w[c] = y[mq_reflect(m, len)];
                }
                v[i] = make_float4(w[0], w[1], w[2], w[3]);
            }
        }
    }
}

template <int NV>
__device__ __forceinline__ void mq_fb_store(const MqOct& O, int u_lo, int u_hi, int lane, const float4 (&v)[NV],
                                            float* __restrict__ lw) {
    int ldst = O.ldst;
    asm volatile("" : "+v"(ldst));         // see mq_fb_load
    float* __restrict__ d0 = lw + ldst;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int u = u_lo + i;
        if (u < u_hi) {
            float* __restrict__ d = d0 + u * O.dstride;
            d[0] = v[i].x;
            d[O.c1] = v[i].y;
            d[O.c2] = v[i].z;
            d[O.c3] = v[i].w;
        }
    }
}

// lane-major A operand of one octave -> registers (NFFT / 8 sixteen-byte loads off one pointer)
template <int NFFT>
__device__ __forceinline__ void mq_fb_operand(const MqOct& O, int lane, float (&a)[NFFT / 2]) {
    const float4* __restrict__ ga = reinterpret_cast<const float4*>(O.ga + lane * (NFFT / 2));
#pragma unroll
    for (int s = 0; s < NFFT / 8; ++s) {
        const float4 t = ga[s];
        a[4 * s] = t.x; a[4 * s + 1] = t.y; a[4 * s + 2] = t.z; a[4 * s + 3] = t.w;
    }
}

// slot 0 of a filter wave is a 128-tap octave (0..3): its operand was loaded a step ahead; slot 1 is the partner 7 - w
// (64 / 32 / 32 / 16 taps), loaded on the spot.  The (n_fft, log2 of the staging stride) pairs are those of the VQT bank.
__device__ __forceinline__ void mq_fb_compute_heavy(float* __restrict__ out, int n_frames, const MqOct& O, int clip, int t0,
                                                    const float* __restrict__ lw, int lane, const float (&a)[64]) {
    switch (O.lgh) {
        case 7: fb_multiply<128, 7, 2>(out, n_frames, O.hop, O.oct, clip, t0, lw, lane, a); break;
        case 6: fb_multiply<128, 6, 2>(out, n_frames, O.hop, O.oct, clip, t0, lw, lane, a); break;
        default: fb_multiply<128, 5, 2>(out, n_frames, O.hop, O.oct, clip, t0, lw, lane, a); break;
    }
}
template <int NFFT, int LG>
__device__ __forceinline__ void mq_fb_compute(float* __restrict__ out, int n_frames, const MqOct& O, int clip, int t0,
                                              const float* __restrict__ lw, int lane) {
    float a[NFFT / 2];
    mq_fb_operand<NFFT>(O, lane, a);
    fb_multiply<NFFT, LG, 2>(out, n_frames, O.hop, O.oct, clip, t0, lw, lane, a);
}
__device__ __forceinline__ void mq_fb_compute_light(float* __restrict__ out, int n_frames, const MqOct& O, int clip, int t0,
                                                    const float* __restrict__ lw, int lane) {
    switch (O.lgh) {
        case 4: mq_fb_compute<64, 4>(out, n_frames, O, clip, t0, lw, lane); break;
        case 3: mq_fb_compute<32, 3>(out, n_frames, O, clip, t0, lw, lane); break;
        case 2: mq_fb_compute<32, 2>(out, n_frames, O, clip, t0, lw, lane); break;
        default: mq_fb_compute<16, 1>(out, n_frames, O, clip, t0, lw, lane); break;
    }
}

__global__ void __launch_bounds__(MQ_NT) xqt_march_kernel(MqArgs P) {
    extern __shared__ __attribute__((aligned(16))) double mq_lds[];
    double* __restrict__ dwin = mq_lds;
    float* __restrict__ fbl = reinterpret_cast<float*>(mq_lds + MQ_F64);
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int nmine = (P.n_clips - b + P.G - 1) / P.G;        // clips b, b + G, ... of this workgroup
    const int steps = (nmine * P.K + P.tail + 3) & ~3;        // whole periods of four steps
    for (int i = tid; i < MQ_F64; i += MQ_NT) dwin[i] = 0.0;
    __syncthreads();
    // wave-uniform role (a scalar branch: each wave runs exactly one of the two step loops, same number of barriers)
    if (__builtin_amdgcn_readfirstlane(tid >> 6) < MQ_NPYR / 64) {
        // ------------------------------------------------------------------ pyramid role
        // level j and group g of this thread: tid 0..255 level 1, 256..383 level 2, ... 504..507 level 7, 508..511 idle
        const int r = 511 - tid;
        const bool idle = r < 4;
        const int j = idle ? 7 : 9 - (31 - __builtin_clz(r));
        const int g = tid - 512 + (512 >> (j - 1));
        const int Sj = MQ_S0 >> j;
        const int Dj = mq_D(j);
        const int src_c = mq_col0(j - 1), src_n = mq_ncol(j - 1);          // window level j reads
        const int dst_c = j < 7 ? mq_col0(j) : 0, dst_n = j < 7 ? mq_ncol(j) : 0, hj = j < 7 ? mq_h(j) : 0;
        int n_out = 0;
        float* ylev = nullptr;
        long long ystride = 0;
#pragma unroll
        for (int l = 1; l < XQ_OCT; ++l)      // constant indices into the kernel arguments (j differs per lane in the last wave)
            if (j == l) { n_out = P.n_out[l]; ylev = const_cast<float*>(P.F.L.y[l]); ystride = P.F.L.stride[l]; }
        // slide role: window of level jj, history sample i
        const int jj = tid / 72, si = tid - jj * 72;
        const bool slider = jj < MQ_LEV && si < mq_h(jj);
        const int sl_c = mq_col0(jj < MQ_LEV ? jj : 0), sl_n = mq_ncol(jj < MQ_LEV ? jj : 0), sl_S = MQ_S0 >> (jj < MQ_LEV ? jj : 0);
        // stream positions (clip c, local step u) of the three roles: level 0 loader (lag 0), own level (lag j), slide (lag jj)
        int u0 = 0, c0 = 0, uj = -j, cj = 0, us = -jj, cs = 0;
        double tl[32];                        // the taps: scalar registers for the whole loop (r[k] = l[k + 1])
#pragma unroll
        for (int i = 0; i < 32; ++i) tl[i] = P.taps->l[i];
        for (int s = 0; s < steps; ++s) {
            const int par = s & 1;
            // ---- level 0: next chunk, global -> registers (consumed at the end of the step)
            float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 < nmine) {
                const int idx = u0 * MQ_S0 + 4 * tid;
                const float* __restrict__ x = P.F.L.y[0] + (long long)(b + c0 * P.G) * P.F.L.stride[0];
                if (P.F.L.vec[0] && idx + 3 < P.n) {
                    f0 = *reinterpret_cast<const float4*>(x + idx);
                } else {
                    if (idx < P.n) f0.x = x[idx];
                    if (idx + 1 < P.n) f0.y = x[idx + 1];
                    if (idx + 2 < P.n) f0.z = x[idx + 2];
                    if (idx + 3 < P.n) f0.w = x[idx + 3];
                }
            }
            // ---- own level: four outputs from the window the level below filled in the previous step
            if (!idle && uj >= 0 && cj < nmine) {
                const int t0 = uj * Sj - Dj + 4 * g;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (t0 + 3 >= 0 && t0 < n_out && !(P.dbg & 1)) {
                    const double* __restrict__ src = dwin + src_c + par * src_n;
                    double acc[4] = {0.0, 0.0, 0.0, 0.0};   // always holds a float32-representable value
                    dc_taps63<MQ_ROW>(src + g, tl, acc);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = (t0 + k >= 0 && t0 + k < n_out) ? dc_div_sqrt_half(acc[k]) : 0.f;
                }
                if (t0 >= 0 && (long long)t0 < ystride)
                    *reinterpret_cast<float4*>(ylev + (long long)(b + cj * P.G) * ystride + t0) = make_float4(v[0], v[1], v[2], v[3]);
                if (j < 7) {
                    const int m0 = hj + 4 * g;
                    double* __restrict__ d = dwin + dst_c + (par ^ 1) * dst_n + (m0 & 7) * MQ_ROW + (m0 >> 3);
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k * MQ_ROW] = (double)v[k];
                }
            }
            // ---- slide: the last h samples of level jj's previous window open its next one (zeros when a clip starts)
            if (slider) {
                const int ms = sl_S + si;
                const double hv = (us > 0 && cs < nmine) ? dwin[sl_c + par * sl_n + (ms & 7) * MQ_ROW + (ms >> 3)] : 0.0;
                dwin[sl_c + (par ^ 1) * sl_n + (si & 7) * MQ_ROW + (si >> 3)] = hv;
            }
            // ---- level 0: registers -> float64 window of the next step
            {
                const int m0 = mq_h(0) + 4 * tid;
                double* __restrict__ d = dwin + (par ^ 1) * mq_ncol(0) + (m0 & 7) * MQ_ROW + (m0 >> 3);
                d[0] = (double)f0.x;
                d[MQ_ROW] = (double)f0.y;
                d[2 * MQ_ROW] = (double)f0.z;
                d[3 * MQ_ROW] = (double)f0.w;
            }
            if (++u0 == P.K) { u0 = 0; ++c0; }
            if (++uj == P.K) { uj = 0; ++cj; }
            if (++us == P.K) { us = 0; ++cs; }
            __syncthreads();
        }
    } else {
        // ------------------------------------------------------------------ filter-bank role
        // One filter wave shares its SIMD with two pyramid waves that always have a float64 instruction ready; it issues
        // few instructions but every step waits for it at the barrier: raise its issue priority.
        __builtin_amdgcn_s_setprio(3);
        const int lane = tid & 63;
        const int w = __builtin_amdgcn_readfirstlane((tid - MQ_NPYR) >> 6);
        MqOct O0{}, O1{};
        int reg0 = 0, reg1 = 0;
        auto fill = [&](MqOct& O, const MqSlot& S, int i, int& region) {      // i: compile-time octave of the slot
            O.y = P.F.L.y[i]; O.stride = P.F.L.stride[i]; O.ga = P.F.fira + P.F.offa[i];       // (the launcher hands the lane-major table in F.fira)
            O.len = P.F.L.len[i]; O.hop = P.F.L.hop[i]; O.vec = P.F.L.vec[i]; O.oct = i; O.nfft = S.nfft;
            O.lgn = 31 - __builtin_clz(S.nfft);
            O.win = O.hop >= O.nfft;
            O.lgh = 31 - __builtin_clz(O.win ? O.nfft : O.hop);
            {
                const int e0 = 4 * lane;
                auto dst = [&](int e) { return O.win ? (e >> O.lgn) * (O.nfft + 1) + (e & (O.nfft - 1)) : e + (e >> O.lgh); };
                O.lsrc = O.win ? (e0 >> O.lgn) * O.hop + (e0 & (O.nfft - 1)) : e0;
                O.sstride = O.win ? (256 >> O.lgn) * O.hop : 256;
                O.ldst = dst(e0);
                O.dstride = dst(256) - dst(0);
                O.c1 = dst(1) - dst(0); O.c2 = dst(2) - dst(0); O.c3 = dst(3) - dst(0);   // the same for every lane (e0 % 4 == 0)
            }
            O.rb = S.rb; O.ub0 = S.ub[0]; O.ub1 = S.ub[1]; O.ub2 = S.ub[2]; O.ub3 = S.ub[3]; O.ub4 = S.ub[4];
            O.rel = -S.rb; O.c = 0; O.valid = 0; O.clip = 0; O.t0 = 0;
            region = S.region;
        };
#pragma unroll
        for (int ww = 0; ww < 4; ++ww)
            if (w == ww) { fill(O0, P.slot[ww][0], ww, reg0); fill(O1, P.slot[ww][1], 7 - ww, reg1); }
        float* __restrict__ lw0 = fbl + reg0;
        float* __restrict__ lw1 = fbl + reg1;
        float* __restrict__ out = P.F.out;
        const int n_frames = P.F.n_frames;
        // The schedule has period four (K and every slot's start step are multiples of 4 / 4 + 2): in step phase p the
        // 128-tap slot requests quarter p of its tile and the short slot quarter (p + 2) & 3.  A quarter requested in one step
        // goes registers -> LDS at the top of the next (its loads had a whole step to land; the barrier between is a bare
        // s_barrier, plain loads survive it).  Phase 0: the 128-tap tile completed by that store is multiplied, with the A
        // operand requested in phase 3; phase 2: the short tile.  All four filter waves do the same thing in the same step.
        float4 v0[4], v1[2];
        float a0[64];
        int st0 = 0, st1 = 0;                 // a quarter of the slot is in registers
        O1.rel += 2;                          // the short slot's tiles start in phase 2
        auto start_tile = [&](MqOct& O, int& comp, int& cclip, int& ct0) {   // at the slot's quarter-0 step
            comp = O.valid; cclip = O.clip; ct0 = O.t0;                      // the tile staged over the last four steps is complete
            O.valid = O.rel >= 0 && O.rel < P.nt4 && O.c < nmine && !(P.dbg & 4);
            O.clip = b + O.c * P.G;
            O.t0 = 8 * O.rel;                                                // 32 frames per four steps
            O.rel += 4;
            if (O.rel == P.K) { O.rel = 0; ++O.c; }
        };
        for (int s = 0; s < steps; s += 4) {
            int comp, cclip, ct0;
            // ---- phase 0
            if (st0) mq_fb_store<4>(O0, O0.ub3, O0.ub4, lane, v0, lw0);
            if (st1) mq_fb_store<2>(O1, O1.ub1, O1.ub2, lane, v1, lw1);
            start_tile(O0, comp, cclip, ct0);
            st0 = O0.valid; st1 = O1.valid;
            if (st0) mq_fb_load<4>(O0, O0.clip, O0.t0, O0.ub0, O0.ub1, lane, v0);
            if (st1) mq_fb_load<2>(O1, O1.clip, O1.t0, O1.ub2, O1.ub3, lane, v1);
            if (comp && !(P.dbg & 2)) mq_fb_compute_heavy(out, n_frames, O0, cclip, ct0, lw0, lane, a0);
            __syncthreads();
            // ---- phase 1
            if (st0) { mq_fb_store<4>(O0, O0.ub0, O0.ub1, lane, v0, lw0); mq_fb_load<4>(O0, O0.clip, O0.t0, O0.ub1, O0.ub2, lane, v0); }
            if (st1) { mq_fb_store<2>(O1, O1.ub2, O1.ub3, lane, v1, lw1); mq_fb_load<2>(O1, O1.clip, O1.t0, O1.ub3, O1.ub4, lane, v1); }
            __syncthreads();
            // ---- phase 2
            if (st0) { mq_fb_store<4>(O0, O0.ub1, O0.ub2, lane, v0, lw0); mq_fb_load<4>(O0, O0.clip, O0.t0, O0.ub2, O0.ub3, lane, v0); }
            if (st1) mq_fb_store<2>(O1, O1.ub3, O1.ub4, lane, v1, lw1);
            start_tile(O1, comp, cclip, ct0);
            st1 = O1.valid;
            if (st1) mq_fb_load<2>(O1, O1.clip, O1.t0, O1.ub0, O1.ub1, lane, v1);
            if (comp && !(P.dbg & 2)) mq_fb_compute_light(out, n_frames, O1, cclip, ct0, lw1, lane);
            __syncthreads();
            // ---- phase 3
            if (st0) { mq_fb_store<4>(O0, O0.ub2, O0.ub3, lane, v0, lw0); mq_fb_load<4>(O0, O0.clip, O0.t0, O0.ub3, O0.ub4, lane, v0); }
            if (st1) { mq_fb_store<2>(O1, O1.ub0, O1.ub1, lane, v1, lw1); mq_fb_load<2>(O1, O1.clip, O1.t0, O1.ub1, O1.ub2, lane, v1); }
            if (st0) mq_fb_operand<128>(O0, lane, a0);       // for the multiply in the next step
            __syncthreads();
        }
    }
}

}  // namespace zns

using namespace zns;

// Launch plan of the marching kernel: steps per clip, the filter-bank waves' schedules and LDS regions.
static int launch_march(const Bank* b, const XqtLevels& L, const int* len, int n_clips, int n_samples, float* out,
                        hipStream_t st) {
    MqArgs A{};
    A.F.L = L;
    A.F.fira = b->firat;              // lane-major A operand
    for (int i = 0; i < XQ_OCT; ++i) A.F.offa[i] = b->offa[i];
    A.F.n_frames = zns_xqt_frames(n_samples);
    A.F.tiles = (A.F.n_frames + 31) / 32;
    A.F.n_clips = n_clips;
    A.F.out = out;
    A.taps = b->taps;
    A.n = n_samples;
    A.n_clips = n_clips;
    int ncu = 256;
    {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            ncu = v;
    }
    A.G = std::min(n_clips, ncu);
    A.nt4 = 4 * A.F.tiles;
    // steps per clip: every level must have produced its whole (padded) row, and the filter schedule (4 steps per tile)
    // of one clip must end before the next one's starts
    int K = A.nt4;
    for (int j = 0; j < XQ_OCT; ++j) {
        const long long need = (j == 0 ? (long long)n_samples : L.stride[j]) + mq_D(j);
        K = std::max(K, (int)((need + (MQ_S0 >> j) - 1) / (MQ_S0 >> j)));
        A.n_out[j] = j == 0 ? n_samples : len[j - 1] / 2;     // resampy: int(n * 0.5); librosa pads to ceil with zeros
    }
    A.K = (K + 3) & ~3;                  // the filter waves' schedule has period four
    int region = 0, max_rb = 0;
    for (int w = 0; w < 4; ++w)
        for (int sl = 0; sl < 2; ++sl) {
            MqSlot& S = A.slot[w][sl];
            const int i = sl == 0 ? w : 7 - w;
            const int nf = b->n_fft[i], hp = b->hop[i], D = mq_D(i);
            ZNS_CHECK_ARG(nf <= 128 && nf >= 16 && hp >= 2, "xqt (marching kernel): octave %d has n_fft %d", i, nf);
            S.oct = i;
            S.nfft = nf;
            S.region = region;
            const bool win = hp >= nf;
            const int nel = win ? 32 * nf : 31 * hp + nf;
            const int nv = (nel + 255) / 256;
            const int lgh = 31 - __builtin_clz(win ? nf : hp);
            {   // the compute instantiations of the kernel (mq_fb_compute_heavy / _light)
                const bool ok = sl == 0 ? (nf == 128 && lgh >= 5 && lgh <= 7)
                                        : ((nf == 64 && lgh == 4) || (nf == 32 && (lgh == 3 || lgh == 2)) || (nf == 16 && lgh == 1));
                ZNS_CHECK_ARG(ok, "xqt (marching kernel): octave %d (n_fft %d, hop %d) has no instance", i, nf, hp);
            }
            region += win ? 32 * (nf + 1) : (256 * nv + ((256 * nv) >> lgh) + 4);
            region = (region + 3) / 4 * 4;
            // the tile's last sample (32 q + 31) hop + n_fft / 2 - 1 exists after level i's local step 4 q + e, e + 1 =
            // ceil((31 hop + n_fft / 2 + D) / (8 hop)); level i runs i steps behind level 0 and its rows are visible to the
            // filter waves one barrier later; the quarters are loaded in the four steps up to then
            S.rb = (31 * hp + nf / 2 + D + 8 * hp - 1) / (8 * hp) + i - 2;
            // Later is always allowed (the rows stay in memory).  All four filter waves multiply their 128-tap tile in the
            // SAME step of every four and their short tile two steps later: a multiply takes issue slots from the two
            // pyramid waves of its SIMD, and the step lasts as long as the slowest SIMD -- four equally loaded SIMDs in one
            // step out of four instead of one straggler in three out of four (measured: 3.49 -> see DESIGN.md section 5).
            S.rb += (((sl == 0 ? 0 : 2) - S.rb) % 4 + 4) % 4;
            max_rb = std::max(max_rb, S.rb);
            S.ub[0] = 0;
            for (int k = 0; k < 4; ++k) {
                const long long avail = (long long)(k + S.rb - i) * 8 * hp - D;     // samples (relative to the tile's frame 0) that exist
                int cnt = 0;
                for (int u = 0; u < nv; ++u) {
                    const int emax = std::min(256 * u + 255, nel - 1);
                    const long long rel = win ? (long long)(emax / nf) * hp + (nf - 1) - nf / 2 : (long long)emax - nf / 2;
                    if (rel < avail) cnt = u + 1; else break;
                }
                S.ub[k + 1] = std::max(S.ub[k], std::min(cnt, S.ub[k] + (sl == 0 ? 4 : 2)));   // registers per quarter: 4 / 2 float4
            }
            ZNS_CHECK_ARG(S.ub[4] == nv, "xqt (marching kernel): octave %d staging plan covers %d of %d pieces", i, S.ub[4], nv);
        }
    A.tail = std::max(8, max_rb + 6);     // ... + the steps in which the last quarter is stored and its tile multiplied
    if (const char* e = getenv("ZNS_XQT_DBG")) A.dbg = atoi(e);
    const size_t lds = (size_t)MQ_F64 * sizeof(double) + (size_t)region * sizeof(float);
    ZNS_CHECK_ARG(lds <= 160 * 1024, "xqt (marching kernel): %zu bytes of LDS", lds);
    static bool lds_set[ZNS_MAX_DEVICES] = {};
    if (!lds_set[current_device_slot()]) {
        ZNS_HIP(hipFuncSetAttribute((const void*)xqt_march_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lds_set[current_device_slot()] = true;
    }
    hipLaunchKernelGGL(xqt_march_kernel, dim3((unsigned)A.G), dim3(MQ_NT), lds, st, A);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

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
    const int plan_flags = stages & ~3;
    stages &= 3;
    ZNS_CHECK_ARG(stages >= 1 && (plan_flags & ~(ZNS_XQT_SEPARATE | ZNS_XQT_MARCH)) == 0 &&
                  plan_flags != (ZNS_XQT_SEPARATE | ZNS_XQT_MARCH),
                  "xqt: stages=%d (1 pyramid, 2 filter bank, 3 both; + ZNS_XQT_SEPARATE or ZNS_XQT_MARCH)", stages | plan_flags);
    // Both stages: the per-stage launches, unless the caller asks for the marching kernel (ZNS_XQT_MARCH; env
    // ZNS_XQT_MARCH_MIN = n makes it the plan from n clips on).  The marching kernel is the SECOND implementation of the
    // transform (same bits, tests/test_vqt.py) and moves fewer HBM bytes (8-11 GB against 13 GB per 4096 clips), but it is not faster: on gfx950 an fp32
    // MFMA and a vector instruction of the same SIMD do not execute together (tools/ubench_f64_mfma.hip: both = sum), so
    // there is no matrix-pipe time to hide under the float64 issue stream, and at its 2 pyramid waves per SIMD the taps run
    // slower than in the 8-waves-per-SIMD chain launches (4096 clips: 6.25 ms against 5.65-6.0 ms).
    // CQT (256 taps in every octave) does not fit its register budget.
    if (stages == 3 && !(plan_flags & ZNS_XQT_SEPARATE)) {
        int max_fft = 0;
        for (int i = 0; i < XQ_OCT; ++i) max_fft = std::max(max_fft, b->n_fft[i]);
        int min_clips = 0x7fffffff;
        if (const char* e = getenv("ZNS_XQT_MARCH_MIN")) min_clips = atoi(e);    // tuning aid
        const bool can = max_fft <= 128;
        ZNS_CHECK_ARG(can || !(plan_flags & ZNS_XQT_MARCH), "xqt: the marching kernel is built for n_fft <= 128 (vqt)");
        if (can && ((plan_flags & ZNS_XQT_MARCH) || n_clips >= min_clips)) {
            for (int i = 0; i < XQ_OCT; ++i) L.vec[i] = (L.stride[i] % 4 == 0) && ((uintptr_t)L.y[i] % 16 == 0);
            return launch_march(b, L, len, n_clips, n_samples, out, st);
        }
    }
    // decimation pyramid: levels 1..7, grouped into chains (one launch each).  Default plan: the two big levels on
    // their own (no recomputed halo where 3/4 of the taps are), then chains of two and three (the small levels' launches
    // are latency-bound: 6.5 us each for a 32-clip batch).  ZNS_XQT_PLAN="n1,n2,.." overrides the grouping (tuning aid).
    if (stages & 1) {
        // big batches: one launch per level (the fused tail recomputes halos and runs few blocks); small ones: fused tail
        int plan[XQ_OCT] = {1, 1, 2, 3, 1, 1, 1}, nplan = 4;     // tools/exp_xqt_plan_sweep.py: 87 us against 91 for {1,1,1,4} at 32 clips
        if (n_clips >= 256) { nplan = 7; for (int i = 0; i < 7; ++i) plan[i] = 1; }
        if (const char* e = getenv("ZNS_XQT_PLAN")) {
            nplan = 0;
            int tot = 0;
            for (const char* q = e; *q && nplan < XQ_OCT;) {
                const int v = atoi(q);
                ZNS_CHECK_ARG(v >= 1 && v <= DC_MAXLEV, "ZNS_XQT_PLAN: chain of %d levels (1..%d)", v, DC_MAXLEV);
                plan[nplan++] = v;
                tot += v;
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
            ZNS_CHECK_ARG(tot == XQ_OCT - 1, "ZNS_XQT_PLAN must cover the %d decimation levels", XQ_OCT - 1);
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
            size_t lds = (size_t)(nlev > 1 ? 2 : 1) * DC_WIN * sizeof(double);
            if (const char* e = getenv("ZNS_XQT_LDS_PAD")) {      // tuning aid: cap the chain kernel's occupancy through its LDS size
                lds = std::max(lds, (size_t)atoi(e) * 1024);
                static bool set = false;
                if (!set) { ZNS_HIP(hipFuncSetAttribute((const void*)decimate_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; }
            }
            hipLaunchKernelGGL(decimate_chain_kernel, dim3((unsigned)((size_t)n_clips * P.blocks_per_clip)), dim3(256), lds,
                               st, P, b->taps);
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
    if (const char* e = getenv("ZNS_XQT_TPW")) F.tpw = std::max(1, std::min(FB_TPW, atoi(e)));   // tuning aid
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
