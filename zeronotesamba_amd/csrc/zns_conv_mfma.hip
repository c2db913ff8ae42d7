// fp32-MFMA implicit-GEMM convolutions for cv2..cv8 of _CNN (models.py:17-23, 41-70) on gfx950: the DIRECT (time-domain) family.
// ZNS_IMPL_AUTO runs cv2..cv8 in the frequency domain (zns_conv_fft.hip) wherever a row has at least 32 frames; these kernels
// serve shorter rows, ZNS_IMPL_DIRECT and the parity tests that hold the two families together.
//
// Why fp32 MFMA: the parity bar is 1e-4 against the fp32 CPU reference with reductions up to
// K = 26 880 (cv6); v_mfma_f32_32x32x2_f32 is bit-for-bit an fmaf chain, so the roofline for these
// kernels is the dense fp32 matrix peak (157.3 TFLOP/s).  At that rate one CU consumes only
// 2 operand dwords per lane per 64 cycles, so HBM/L2/LDS bandwidth is never the limit here; the whole
// design goal is to keep the matrix pipe issuing back-to-back:
//   * 8 waves (2 per SIMD) per workgroup, 96-128 accumulator VGPRs each,
//   * LDS-DMA (global_load_lds) staging, double buffered, ONE barrier per K-stage,
//   * operands read from LDS with immediate offsets only (no address VALU in the inner loop),
//   * bias / max-pool(H) / ReLU / dropout / argmax (forward) and ReLU-dropout-mask / max-unpool
//     (backward-data) fused into the accumulator epilogue, in registers.
//
// Forward and backward-data share one core ("conv_core"): backward-data is the same correlation with
// the weights packed transposed+flipped.  Backward-weights ("wgrad_core") is a second GEMM whose K is
// the pixel index; it writes split-K partial slabs that a small reducer sums in a fixed order
// (bitwise reproducible, no float atomics).
#include "zns_conv_core.h"

namespace zns {

// =====================================================================================================
//                                          instances + dispatch
// =====================================================================================================
//                 CIN COUT KH KW  H  MT RT CT WM WR WC CK POOL EPI
using F2 = ConvCfg<64, 64, 7, 13, 96, 2, 3, 1, 1, 2, 4, 4, 3, 0>;
using F3 = ConvCfg<64, 128, 5, 15, 32, 4, 2, 1, 1, 2, 4, 4, 1, 0>;
using F4 = ConvCfg<128, 128, 9, 17, 32, 2, 4, 1, 2, 1, 4, 4, 4, 0>;
using F5 = ConvCfg<128, 256, 3, 19, 8, 4, 2, 1, 2, 2, 2, 2, 1, 0>;
using F6 = ConvCfg<256, 256, 5, 21, 8, 1, 8, 1, 8, 1, 1, 2, 8, 0>;
using F7 = ConvCfg<256, 128, 1, 23, 1, 1, 1, 1, 4, 1, 2, 4, 1, 0>;
using F8 = ConvCfg<128, 128, 1, 25, 1, 1, 1, 1, 4, 1, 2, 4, 1, 0>;
// backward-data: core CIN = conv cout, core COUT = conv cin
using D2 = ConvCfg<64, 64, 7, 13, 96, 2, 3, 1, 1, 2, 4, 4, 1, 1>;
using D3 = ConvCfg<128, 64, 5, 15, 32, 2, 4, 1, 1, 2, 4, 4, 1, 1>;
using D4 = ConvCfg<128, 128, 9, 17, 32, 2, 4, 1, 2, 1, 4, 4, 1, 1>;
using D5 = ConvCfg<256, 128, 3, 19, 8, 4, 2, 1, 1, 4, 2, 4, 1, 1>;
using D6 = ConvCfg<256, 256, 5, 21, 8, 4, 2, 1, 2, 4, 1, 2, 1, 1>;
using D7 = ConvCfg<128, 256, 1, 23, 1, 2, 1, 1, 4, 1, 2, 2, 1, 1>;
using D8 = ConvCfg<128, 128, 1, 25, 1, 1, 1, 1, 4, 1, 2, 4, 1, 1>;
// cv7 / cv8 with 32-column workgroups (4 waves) for small launches: the batch-of-one fine-tune step (epochs.py:48-63,
// T = 1876) gives the 64-column tiles only 60 workgroups for 256 CUs.  Same packed-weight layout (COB, CK, MT, WM).
using F7s = ConvCfg<256, 128, 1, 23, 1, 1, 1, 1, 4, 1, 1, 4, 1, 0>;
using F8s = ConvCfg<128, 128, 1, 25, 1, 1, 1, 1, 4, 1, 1, 4, 1, 0>;
using D7s = ConvCfg<128, 256, 1, 23, 1, 2, 1, 1, 4, 1, 1, 2, 1, 1>;
using D8s = ConvCfg<128, 128, 1, 25, 1, 1, 1, 1, 4, 1, 1, 4, 1, 1>;
// cv7 / cv8 with the input-channel loop SPLIT over workgroups (K-split): the configuration's CIN is an eighth of the layer's;
// a workgroup runs through 8 / ks consecutive eighths, the ks workgroups of a tile write raw partial sums and
// conv_split_finish_kernel adds them in a fixed order and applies the layer's epilogue.  Why: a 32 x 32 tile's whole K loop on
// one SIMD is the unit of these launches, and there are few of them -- T = 512: 256 workgroups of 8 waves = one per CU with its
// 94 KB of operands (matrix pipes 0.62-0.78 busy); T = 313: 160 workgroups on 256 CUs; the batch-of-one fine-tune step: 59.
// With 2 input channels per stage the operands are 48 KB: three workgroups per CU, several K loops' barriers interleaved.
// `*kP` are the same shapes with the layer's full CIN: the layout the operand is packed in (never instantiated as a kernel).
constexpr int KSPLIT_PARTS = 8;
using F7k = ConvCfg<256 / KSPLIT_PARTS, 128, 1, 23, 1, 1, 1, 1, 4, 1, 2, 2, 1, 1>;
using F8k = ConvCfg<128 / KSPLIT_PARTS, 128, 1, 25, 1, 1, 1, 1, 4, 1, 2, 2, 1, 1>;
using D7k = ConvCfg<128 / KSPLIT_PARTS, 256, 1, 23, 1, 1, 1, 1, 4, 1, 2, 2, 1, 1>;
using D8k = F8k;
using F7kP = ConvCfg<256, 128, 1, 23, 1, 1, 1, 1, 4, 1, 2, 2, 1, 1>;
using F8kP = ConvCfg<128, 128, 1, 25, 1, 1, 1, 1, 4, 1, 2, 2, 1, 1>;
using D7kP = ConvCfg<128, 256, 1, 23, 1, 1, 1, 1, 4, 1, 2, 2, 1, 1>;
using D8kP = F8kP;
static_assert(F7k::NCC % 2 == 0 && F8k::NCC % 2 == 0 && D7k::NCC % 2 == 0, "parts are chained like items: even stage counts");

// true when the wide tiles would leave most CUs without a workgroup
template <class Wide>
static bool few_workgroups(const ConvArgs& a) {
    return (long long)a.nbr * a.B * ceil_div(a.T, Wide::WB) * (Wide::COUT / Wide::COB) * (Wide::H / Wide::RB) < 160;
}
//                CIN COUT KH KW  H  MT NS WMW WQ R
using W2 = WgCfg<64, 64, 7, 13, 96, 2, 4, 1, 8, 1>;
using W3 = WgCfg<64, 128, 5, 15, 32, 2, 4, 2, 4, 1>;
using W4 = WgCfg<128, 128, 9, 17, 32, 2, 4, 2, 4, 1>;
using W5 = WgCfg<128, 256, 3, 19, 8, 2, 4, 2, 4, 1>;
using W6 = WgCfg<256, 256, 5, 21, 8, 2, 4, 2, 4, 1>;
using W7 = WgCfg<256, 128, 1, 23, 1, 2, 3, 2, 4, 1>;
using W8 = WgCfg<128, 128, 1, 25, 1, 2, 4, 2, 4, 1>;

static int layer_of(const ConvArgs& a) {
    struct L { int cin, cout, kh, kw, H; };
    static const L tab[] = {{64, 64, 7, 13, 96},  {64, 128, 5, 15, 32}, {128, 128, 9, 17, 32}, {128, 256, 3, 19, 8},
                            {256, 256, 5, 21, 8}, {256, 128, 1, 23, 1}, {128, 128, 1, 25, 1}};
    for (int i = 0; i < 7; ++i)
        if (tab[i].cin == a.cin && tab[i].cout == a.cout && tab[i].kh == a.kh && tab[i].kw == a.kw && tab[i].H == a.H)
            return i + 2;
    return -1;
}

bool mfma_supported(const ConvArgs& a) { return layer_of(a) > 0 && a.nbr >= 1 && a.nbr <= 2; }

template <class C>
static size_t wg_slab_bytes(const ConvArgs& a) {
    int tt, tps;
    const int ks = wg_pick_ks<C>(a, &tt, &tps);
    return (size_t)a.nbr * ks * C::Q * C::COUT * C::CIN * sizeof(float);
}

static size_t packed_bytes(const ConvArgs& a) {
    return (size_t)a.nbr * a.cout * a.cin * a.kh * a.kw * sizeof(float);
}

template <class C>
static int ksplit_ways(const ConvArgs& a);

size_t mfma_workspace_bytes(const ConvArgs& a) {
    size_t slab = 0;
    switch (layer_of(a)) {
        case 2: slab = wg_slab_bytes<W2>(a); break;
        case 3: slab = wg_slab_bytes<W3>(a); break;
        case 4: slab = wg_slab_bytes<W4>(a); break;
        case 5: slab = wg_slab_bytes<W5>(a); break;
        case 6: slab = wg_slab_bytes<W6>(a); break;
        case 7: slab = wg_slab_bytes<W7>(a); break;
        case 8: slab = wg_slab_bytes<W8>(a); break;
        default: return 0;
    }
    const size_t pk = (packed_bytes(a) + 255) & ~(size_t)255;
    size_t need = pk > slab ? pk : slab;
    // the K-split forward / backward-data: packed operand + partial sums (COUT = max(cin, cout)), only where the launch splits
    if (layer_of(a) >= 7 && (ksplit_ways<F7k>(a) > 1 || ksplit_ways<D7k>(a) > 1)) {
        const size_t part = (size_t)KSPLIT_PARTS * a.nbr * a.B * (a.cin > a.cout ? a.cin : a.cout) * a.T * sizeof(float);
        if (pk + part > need) need = pk + part;
    }
    return need;
}

template <class C>
static int set_lds(void (*kern)(CoreArgs)) {
    static bool done[ZNS_MAX_DEVICES] = {};
    const int d = current_device_slot();
    if (!done[d]) {
        ZNS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        done[d] = true;
    }
    return ZNS_OK;
}
template <class C>
static int set_lds_wg(void (*kern)(WgArgs)) {
    static bool done[ZNS_MAX_DEVICES] = {};
    const int d = current_device_slot();
    if (!done[d]) {
        ZNS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        done[d] = true;
    }
    return ZNS_OK;
}

template <class C>
static int pack_for(const ConvArgs& a, Ptr2 w, float* dst, int transposed, hipStream_t st) {
    const long long per_branch = (long long)C::COUT * C::CIN * C::KH * C::KW;
    const long long total = per_branch * a.nbr;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, st, w, dst, C::CIN, C::COUT, C::KH, C::KW,
                       C::COB, C::CK, C::MT, C::WM, transposed, per_branch, a.nbr);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

template <class C>
static int launch_core(const ConvArgs& a, CoreArgs A, Ptr2 w, int transposed, const float* wpk, void* ws,
                       size_t ws_bytes, hipStream_t st) {
    int rc = ZNS_OK;
    if (wpk == nullptr) {
        if (ws_bytes < packed_bytes(a)) {
            set_error("conv: workspace %zu < %zu", ws_bytes, packed_bytes(a));
            return ZNS_ERR_WORKSPACE;
        }
        rc = pack_for<C>(a, w, (float*)ws, transposed, st);
        if (rc) return rc;
        wpk = (const float*)ws;
    }
    rc = set_lds<C>(conv_core<C>);
    if (rc) return rc;
    A.wpk = wpk;
    A.B = a.B;
    A.T = a.T;
    A.N = a.nbr * a.B;
    A.ncb = ceil_div(a.T, C::WB);
    const long long blocks = (long long)A.N * (C::H / C::RB) * A.ncb * (C::COUT / C::COB);
    const bool probe = probe_match(transposed ? 1 : 0, a.cin, a.cout, a.kh);
    const int slot = probe ? g_probe.count % g_probe.n : 0;
    if (probe) ZNS_HIP(hipEventRecord(g_probe.start[slot], st));
    hipLaunchKernelGGL(conv_core<C>, dim3((unsigned)blocks), dim3(C::NT), C::LDS_BYTES, st, A);
    ZNS_LAUNCH_CHECK();
    if (probe) { ZNS_HIP(hipEventRecord(g_probe.stop[slot], st)); ++g_probe.count; }
    return ZNS_OK;
}

// The partial sums of a K-split launch, [ks][N][COUT][T] (H = 1), added pairwise in a fixed tree, then the epilogue conv_core
// would have run on its accumulators: MODE 0 forward (bias, ReLU, dropout -- cv7 / cv8 have no pooling), MODE 1 backward-data
// (ReLU / dropout mask of the previous layer, max-unpool over its `pp` rows).
struct SplitFin {
    Ptr2 bias;
    float* out;
    const float* xmask;
    const uint8_t* prev_argmax;
    uint32_t thresh, seed;
    float inv_keep;
    int layer, B, cout, T, pp, ks;
};
template <int MODE>
__global__ void __launch_bounds__(256) conv_split_finish_kernel(const float* __restrict__ slab, size_t per, SplitFin F) {
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= per) return;
    // the parts are added PAIRWISE (a fixed tree: bitwise repeatable, and the rounding error of the sum of 8 grows with
    // log2(8) = 3 additions per element instead of 7)
    float s;
    if (F.ks == 8) {
        float p[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] = slab[(size_t)k * per + o];
        s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    } else if (F.ks == 4) {
        s = (slab[o] + slab[per + o]) + (slab[2 * per + o] + slab[3 * per + o]);
    } else {
        s = slab[o];
        for (int k = 1; k < F.ks; ++k) s += slab[(size_t)k * per + o];
    }
    const int col = (int)(o % F.T);
    const size_t nc = o / F.T;                      // n * cout + co
    const int co = (int)(nc % F.cout), n = (int)(nc / F.cout);
    if (MODE == 0) {
        const int br = n / F.B, b_in_br = n - br * F.B;
        float v = relu_nan(s + F.bias.p[br][co]);
        if (F.thresh) {
            const uint64_t idx = ((uint64_t)b_in_br * F.cout + co) * F.T + col;
            v = drop_apply(drop_keep(layer_seed(F.seed, F.layer, br), idx, F.thresh), v, F.inv_keep);
        }
        F.out[o] = v;
    } else {
        float g = s;
        if (F.xmask) g = (F.xmask[o] != 0.f) ? g * F.inv_keep : 0.f;
        if (F.pp > 1) {
            const int am = F.prev_argmax[o];
            const size_t base = nc * F.pp * F.T + col;
            for (int a = 0; a < F.pp; ++a) F.out[base + (size_t)a * F.T] = (a == am) ? g : 0.f;
        } else {
            F.out[o] = g;
        }
    }
}

static int ksplit_mode() { return tuning(ZNS_TUNE_KSPLIT); }   // -1: the product rule; 0: off; 2 / 4 / 8: forced split (tests)
// How many ways the K loop of a cv7 / cv8 launch is split (1 = the unsplit kernels): all eight, at every T -- never a
// function of the batch: the summation order of a clip's activations must not depend on the batch it is in
// (tests/test_gpu_fullsize.py: a batch of 80 equals its chunks of 16 bit for bit).  Measured per step, unsplit / 2 / 4 / 8
// ways: T = 512 (8 tiles per clip, 256 per batch of 16 pairs = one per CU unsplit): 23.23 / 23.12 / 23.25 / 23.23 ms (what
// the split kernels gain there, the finish launches cost: cv7 backward-data writes 134 MB of unpooled rows in any case);
// T = 313 (5 tiles per clip, 160 per batch): 16.90 / 16.77 / 16.74 / 16.73 ms; the batch-of-one fine-tune step (T = 1876, 30
// tiles): 7.24 / 7.04 / 7.02 / 6.97 ms.  The split sums are also the more accurate ones -- eight chains of an eighth of the
// terms, added pairwise: 3.5e-7 against 9.6e-7 rms of a float64 convolution, where torch's own CPU convolution sits at
// 5e-7 .. 9e-7 (tools/diag_win_accuracy.py) -- so round 5 made them the rule at T = 512 too.  (Round 4 kept T = 512 unsplit
// because the full-size oracle test's worst gradient read 1.01e-4 with the split against 6.2e-5 without: that number was
// the NT-Xent kernels' float32 tables reacting to ANY change of the embeddings' last bits, see zns_small.hip; with
// float64 tables the same test reads 4e-5 unsplit.)
// zns_test_set_tuning(ZNS_TUNE_KSPLIT, 0 / 2 / 4 / 8) overrides (tests compare the unsplit kernels).
template <class C>
static int ksplit_ways(const ConvArgs& a) {
    if (a.H != 1) return 1;
    const int m = ksplit_mode();
    if (m == 0) return 1;
    if (m == 2 || m == 4 || m == 8) return m;
    return KSPLIT_PARTS;
}
template <class C>
static size_t ksplit_slab_bytes(const ConvArgs& a) {
    return (size_t)KSPLIT_PARTS * a.nbr * a.B * C::COUT * a.T * sizeof(float);
}

// C: the part configuration; CP: its full-CIN twin (packing layout); fin: the finish kernel's arguments (out / masks set)
template <class C, class CP, int MODE>
static int launch_split(const ConvArgs& a, const float* x, Ptr2 w, int transposed, const float* wpk, SplitFin F, int ks,
                        void* ws, size_t ws_bytes, hipStream_t st) {
    static_assert(C::EPI == 1 && C::KH == 1 && C::H == 1 && CP::CIN == C::CIN * KSPLIT_PARTS, "split cfg");
    // the part kernel streams the operand packed for its full-CIN twin: everything the packing depends on must agree
    static_assert(CP::COB == C::COB && CP::CK == C::CK && CP::MT == C::MT && CP::WM == C::WM && CP::COUT == C::COUT &&
                  CP::KW == C::KW && CP::KH == C::KH, "split cfg: part and packing twin differ in the packed layout");
    const size_t pk = (packed_bytes(a) + 255) & ~(size_t)255;
    const size_t need = pk + ksplit_slab_bytes<C>(a);
    if (ws_bytes < need) { set_error("conv (K-split): workspace %zu < %zu", ws_bytes, need); return ZNS_ERR_WORKSPACE; }
    int rc = ZNS_OK;
    if (wpk == nullptr) {
        rc = pack_for<CP>(a, w, (float*)ws, transposed, st);
        if (rc) return rc;
        wpk = (const float*)ws;
    }
    float* slab = (float*)((char*)ws + pk);
    rc = set_lds<C>(conv_core<C>);
    if (rc) return rc;
    CoreArgs A{};
    A.x = x;
    A.wpk = wpk;
    A.dx = slab;
    A.xmask = nullptr;
    A.inv_keep = 1.f;
    A.prev_pool = 1;
    A.B = a.B;
    A.T = a.T;
    A.N = a.nbr * a.B;
    A.ncb = ceil_div(a.T, C::WB);
    A.ks = ks;
    A.kparts = KSPLIT_PARTS / ks;
    A.cin_full = CP::CIN;
    const long long blocks = (long long)A.N * A.ncb * (C::COUT / C::COB) * ks;
    const bool probe = probe_match(transposed ? 1 : 0, a.cin, a.cout, a.kh);
    const int slot = probe ? g_probe.count % g_probe.n : 0;
    if (probe) ZNS_HIP(hipEventRecord(g_probe.start[slot], st));
    hipLaunchKernelGGL(conv_core<C>, dim3((unsigned)blocks), dim3(C::NT), C::LDS_BYTES, st, A);
    ZNS_LAUNCH_CHECK();
    if (probe) { ZNS_HIP(hipEventRecord(g_probe.stop[slot], st)); ++g_probe.count; }
    const size_t per = (size_t)A.N * C::COUT * a.T;
    F.B = a.B;
    F.cout = C::COUT;
    F.T = a.T;
    F.ks = ks;
    hipLaunchKernelGGL(conv_split_finish_kernel<MODE>, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st,
                       (const float*)slab, per, F);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int mfma_fwd(const ConvArgs& a, const float* x, Ptr2 w, Ptr2 bias, float* y, uint8_t* argmax, float drop_p,
             uint32_t seed, int layer, const float* wpk, void* ws, size_t ws_bytes, hipStream_t st) {
    CoreArgs A{};
    A.x = x;
    A.bias = bias;
    A.y = y;
    A.argmax = argmax;
    A.thresh = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
    A.inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
    A.seed = seed;
    A.layer = layer;
    switch (layer_of(a)) {
        case 2: return launch_core<F2>(a, A, w, 0, wpk, ws, ws_bytes, st);
        case 3: return launch_core<F3>(a, A, w, 0, wpk, ws, ws_bytes, st);
        case 4: return launch_core<F4>(a, A, w, 0, wpk, ws, ws_bytes, st);
        case 5: return launch_core<F5>(a, A, w, 0, wpk, ws, ws_bytes, st);
        case 6: return launch_core<F6>(a, A, w, 0, wpk, ws, ws_bytes, st);
        case 7:
        case 8: {
            const bool l7 = layer_of(a) == 7;
            const int ks = l7 ? ksplit_ways<F7k>(a) : ksplit_ways<F8k>(a);
            if (ks > 1) {
                SplitFin F{};
                F.bias = bias; F.out = y; F.thresh = A.thresh; F.seed = seed; F.inv_keep = A.inv_keep; F.layer = layer; F.pp = 1;
                return l7 ? launch_split<F7k, F7kP, 0>(a, x, w, 0, wpk, F, ks, ws, ws_bytes, st)
                          : launch_split<F8k, F8kP, 0>(a, x, w, 0, wpk, F, ks, ws, ws_bytes, st);
            }
            if (l7) {
                if (few_workgroups<F7>(a)) return launch_core<F7s>(a, A, w, 0, wpk, ws, ws_bytes, st);
                return launch_core<F7>(a, A, w, 0, wpk, ws, ws_bytes, st);
            }
            if (few_workgroups<F8>(a)) return launch_core<F8s>(a, A, w, 0, wpk, ws, ws_bytes, st);
            return launch_core<F8>(a, A, w, 0, wpk, ws, ws_bytes, st);
        }
    }
    return ZNS_ERR_ARG;
}

int mfma_dgrad(const ConvArgs& a, const float* dz, Ptr2 w, float* dx, const float* x, float inv_keep,
               const uint8_t* prev_argmax, int prev_pool, const float* wpk, void* ws, size_t ws_bytes, hipStream_t st) {
    CoreArgs A{};
    A.x = dz;
    A.dx = dx;
    A.xmask = x;
    A.inv_keep = inv_keep;
    A.prev_argmax = prev_argmax;
    A.prev_pool = prev_pool;
    switch (layer_of(a)) {
        case 2: return launch_core<D2>(a, A, w, 1, wpk, ws, ws_bytes, st);
        case 3: return launch_core<D3>(a, A, w, 1, wpk, ws, ws_bytes, st);
        case 4: return launch_core<D4>(a, A, w, 1, wpk, ws, ws_bytes, st);
        case 5: return launch_core<D5>(a, A, w, 1, wpk, ws, ws_bytes, st);
        case 6: return launch_core<D6>(a, A, w, 1, wpk, ws, ws_bytes, st);
        case 7:
        case 8: {
            const bool l7 = layer_of(a) == 7;
            const int ks = l7 ? ksplit_ways<D7k>(a) : ksplit_ways<D8k>(a);
            if (ks > 1) {
                SplitFin F{};
                F.out = dx; F.xmask = x; F.inv_keep = inv_keep; F.prev_argmax = prev_argmax; F.pp = prev_pool;
                return l7 ? launch_split<D7k, D7kP, 1>(a, dz, w, 1, wpk, F, ks, ws, ws_bytes, st)
                          : launch_split<D8k, D8kP, 1>(a, dz, w, 1, wpk, F, ks, ws, ws_bytes, st);
            }
            if (l7) {
                if (few_workgroups<D7>(a)) return launch_core<D7s>(a, A, w, 1, wpk, ws, ws_bytes, st);
                return launch_core<D7>(a, A, w, 1, wpk, ws, ws_bytes, st);
            }
            if (few_workgroups<D8>(a)) return launch_core<D8s>(a, A, w, 1, wpk, ws, ws_bytes, st);
            return launch_core<D8>(a, A, w, 1, wpk, ws, ws_bytes, st);
        }
    }
    return ZNS_ERR_ARG;
}

// mfma_pack packs for the wide configuration of a layer; the few-workgroup variants of cv7 / cv8 must stream that operand
static_assert(same_pack_layout<F7, F7s>() && same_pack_layout<F8, F8s>() && same_pack_layout<D7, D7s>() &&
              same_pack_layout<D8, D8s>(), "the *s variants reuse the wide variants' packed operand");
int mfma_pack_layout(const ConvArgs& a, int transposed) {
    switch (layer_of(a) * 2 + (transposed ? 1 : 0)) {
        case 4: return pack_layout_id<F2>(2, 0);
        case 5: return pack_layout_id<D2>(2, 1);
        case 6: return pack_layout_id<F3>(2, 0);
        case 7: return pack_layout_id<D3>(2, 1);
        case 8: return pack_layout_id<F4>(2, 0);
        case 9: return pack_layout_id<D4>(2, 1);
        case 10: return pack_layout_id<F5>(2, 0);
        case 11: return pack_layout_id<D5>(2, 1);
        case 12: return pack_layout_id<F6>(2, 0);
        case 13: return pack_layout_id<D6>(2, 1);
        // cv7 / cv8: the K-split kernels stream 2-channel stages, the unsplit ones 4 (2 for cv7 backward-data): the operand's
        // layout follows the choice the launch will make for this geometry
        case 14: return ksplit_ways<F7k>(a) > 1 ? pack_layout_id<F7kP>(2, 0) : pack_layout_id<F7>(2, 0);
        case 15: return ksplit_ways<D7k>(a) > 1 ? pack_layout_id<D7kP>(2, 1) : pack_layout_id<D7>(2, 1);
        case 16: return ksplit_ways<F8k>(a) > 1 ? pack_layout_id<F8kP>(2, 0) : pack_layout_id<F8>(2, 0);
        case 17: return ksplit_ways<D8k>(a) > 1 ? pack_layout_id<D8kP>(2, 1) : pack_layout_id<D8>(2, 1);
    }
    return 0;
}
size_t mfma_pack_bytes(const ConvArgs& a, int transposed) { (void)transposed; return layer_of(a) > 0 ? packed_bytes(a) : 0; }
int mfma_pack(const ConvArgs& a, Ptr2 w, int transposed, float* dst, hipStream_t st) {
    switch (layer_of(a) * 2 + (transposed ? 1 : 0)) {
        case 4: return pack_for<F2>(a, w, dst, 0, st);
        case 5: return pack_for<D2>(a, w, dst, 1, st);
        case 6: return pack_for<F3>(a, w, dst, 0, st);
        case 7: return pack_for<D3>(a, w, dst, 1, st);
        case 8: return pack_for<F4>(a, w, dst, 0, st);
        case 9: return pack_for<D4>(a, w, dst, 1, st);
        case 10: return pack_for<F5>(a, w, dst, 0, st);
        case 11: return pack_for<D5>(a, w, dst, 1, st);
        case 12: return pack_for<F6>(a, w, dst, 0, st);
        case 13: return pack_for<D6>(a, w, dst, 1, st);
        case 14: return ksplit_ways<F7k>(a) > 1 ? pack_for<F7kP>(a, w, dst, 0, st) : pack_for<F7>(a, w, dst, 0, st);
        case 15: return ksplit_ways<D7k>(a) > 1 ? pack_for<D7kP>(a, w, dst, 1, st) : pack_for<D7>(a, w, dst, 1, st);
        case 16: return ksplit_ways<F8k>(a) > 1 ? pack_for<F8kP>(a, w, dst, 0, st) : pack_for<F8>(a, w, dst, 0, st);
        case 17: return ksplit_ways<D8k>(a) > 1 ? pack_for<D8kP>(a, w, dst, 1, st) : pack_for<D8>(a, w, dst, 1, st);
    }
    return ZNS_ERR_ARG;
}

template <class C>
static int launch_wgrad(const ConvArgs& a, const float* dz, const float* x, MPtr2 dw, int accumulate, void* ws,
                        size_t ws_bytes, hipStream_t st) {
    WgArgs A{};
    A.dz = dz;
    A.x = x;
    A.slab = (float*)ws;
    A.B = a.B;
    A.T = a.T;
    A.ncb = ceil_div(a.T, C::CW);
    A.KS = wg_pick_ks<C>(a, &A.tiles_total, &A.tiles_per_split);
    const size_t need = (size_t)a.nbr * A.KS * C::Q * C::COUT * C::CIN * sizeof(float);
    if (ws_bytes < need) {
        set_error("wgrad: workspace %zu < %zu", ws_bytes, need);
        return ZNS_ERR_WORKSPACE;
    }
    int rc = set_lds_wg<C>(wgrad_core<C>);
    if (rc) return rc;
    const long long blocks = (long long)C::NQG * (C::CIN / C::CIB) * (C::COUT / C::COB) * A.KS * a.nbr;
    const bool probe = probe_match(2, a.cin, a.cout, a.kh);
    const int slot = probe ? g_probe.count % g_probe.n : 0;
    if (probe) ZNS_HIP(hipEventRecord(g_probe.start[slot], st));
    hipLaunchKernelGGL(wgrad_core<C>, dim3((unsigned)blocks), dim3(C::NT), C::LDS_BYTES, st, A);
    ZNS_LAUNCH_CHECK();
    if (probe) { ZNS_HIP(hipEventRecord(g_probe.stop[slot], st)); ++g_probe.count; }
    const long long total = (long long)a.nbr * C::Q * C::COUT * C::CIN;
    const int rblocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rblocks), dim3(256), 0, st, (const float*)ws, dw, A.KS, C::Q, C::COUT,
                       C::CIN, accumulate, a.nbr);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

int mfma_wgrad(const ConvArgs& a, const float* dz, const float* x, MPtr2 dw, int accumulate, void* ws,
               size_t ws_bytes, hipStream_t st) {
    switch (layer_of(a)) {
        case 2: return launch_wgrad<W2>(a, dz, x, dw, accumulate, ws, ws_bytes, st);
        case 3: return launch_wgrad<W3>(a, dz, x, dw, accumulate, ws, ws_bytes, st);
        case 4: return launch_wgrad<W4>(a, dz, x, dw, accumulate, ws, ws_bytes, st);
        case 5: return launch_wgrad<W5>(a, dz, x, dw, accumulate, ws, ws_bytes, st);
        case 6: return launch_wgrad<W6>(a, dz, x, dw, accumulate, ws, ws_bytes, st);
        case 7: return launch_wgrad<W7>(a, dz, x, dw, accumulate, ws, ws_bytes, st);
        case 8: return launch_wgrad<W8>(a, dz, x, dw, accumulate, ws, ws_bytes, st);
    }
    return ZNS_ERR_ARG;
}

}  // namespace zns
