// Templates of the two fp32-MFMA convolution cores (conv_core: forward / backward-data, wgrad_core:
// backward-weights) shared by the direct path (zns_conv_mfma.hip) and the frequency-domain path
// (zns_conv_fft.hip).  See zns_conv_mfma.hip for the design notes.
#pragma once
#include <type_traits>
#include "zns_common.h"
#include "zns_kernels.h"

namespace zns {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

static __device__ const float g_zero_page[16] __attribute__((aligned(16))) = {0};

__device__ __forceinline__ void glds4(const float* src, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)lds_dst, 4, 0, 0);
}
__device__ __forceinline__ void glds16(const float* src, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// LDS-DMA through a buffer descriptor: wave-uniform base in SGPRs, ONE 32-bit per-lane byte offset, and hardware range
// checking (a lane whose offset is >= num_records moves zeros): no 64-bit per-lane pointer arithmetic, no select
// against a zero page.  BUF_OOB marks a lane that must read zero.
constexpr unsigned BUF_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void blds4(__amdgpu_buffer_rsrc_t rs, unsigned voff, float* lds_dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)lds_dst, 4, (int)voff, 0, 0, 0);
}
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rs, unsigned voff, float* lds_dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)lds_dst, 16, (int)voff, 0, 0, 0);
}
// the same with the LDS destination as a byte address in a scalar register (a running value the caller steps)
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(size_t)(lds_void_t*)p; }
__device__ __forceinline__ void blds16a(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_byte) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(size_t)lds_byte, 16, (int)voff, (int)soff, 0, 0);
}
__device__ __forceinline__ float lds_ld(unsigned lds_byte) {
    return *(const __attribute__((address_space(3))) float*)(size_t)lds_byte;
}
__device__ __forceinline__ void blds4a(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned lds_byte) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(size_t)lds_byte, 4, (int)voff, 0, 0, 0);
}

// MFMA 32x32 accumulator: lane&31 = column, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int acc_row(int reg, int lane_h) { return (reg & 3) + 8 * (reg >> 2) + 4 * lane_h; }

// =====================================================================================================
//                                   forward / backward-data core
// =====================================================================================================
// DILMAX > 0 selects the frequency-domain use of the core (see zns_conv_fft.hip): a 1-D correlation with KW taps
// spaced `dil` (runtime, <= DILMAX) apart along the flattened (row, time-block) axis, weights chosen per frequency.
//
// HET selects the 12-wave layout for an 11-column-tile output (P = 352): a 4 x 11 grid of 32x32 tiles does not divide
// over 4 SIMDs with equal waves, so waves 0..7 own (4 channel tiles x column tile w) and waves 8..11 own (channel
// tile w-8 x column tiles 8, 9, 10): 2 x 4 + 3 = 11 tiles on every SIMD.
template <int CIN_, int COUT_, int KH_, int KW_, int H_, int MT_, int RT_, int CT_, int WM_, int WR_, int WC_,
          int CK_, int POOL_, int EPI_, int DILMAX_ = 0, int HET_ = 0>
struct ConvCfg {
    static constexpr int DILMAX = DILMAX_;
    static constexpr int HET = HET_;
    // HET = number of type-A waves NA (8: 11 column tiles, 12 waves; 4: 7 column tiles, 8 waves).  Waves 0..NA-1 own
    // (4 channel tiles x column tile w), waves NA..NA+3 own (channel tile w-NA x column tiles NA, NA+1, NA+2).
    static_assert(!HET || ((HET == 8 || HET == 4) && MT_ == 4 && RT_ == 1 && CT_ == 1 && WM_ == 1 && WR_ == 1 &&
                           WC_ == HET + 4 && KH_ == 1 && EPI_ == 1),
                  "HET layout: NA + 4 waves, 4 channel tiles, NA + 3 column tiles");
    static constexpr int CIN = CIN_, COUT = COUT_, KH = KH_, KW = KW_, H = H_;
    static constexpr int MT = MT_, RT = RT_, CT = CT_, WM = WM_, WR = WR_, WC = WC_, CK = CK_;
    static constexpr int POOL = POOL_, EPI = EPI_;
    static constexpr int NW = WM * WR * WC, NT = NW * 64;
    static constexpr int COB = WM * MT * 32, RB = WR * RT, WB = HET ? (HET + 3) * 32 : WC * CT * 32;
    static constexpr int PH = KH / 2, PW = KW / 2;
    // Patch rows start at a 16-byte aligned image column (w0 - pw rounded down to a multiple of 4) and are a
    // multiple of 4 floats long, so the patch can be staged with 16-byte LDS-DMA whenever T % 4 == 0.
    static constexpr int PR = RB + KH - 1, PCW = (WB + (KW - 1) * (DILMAX ? DILMAX : 1) + 3 + 3) / 4 * 4;
    static constexpr int PS = (PR * PCW + 3) / 4 * 4;         // plane stride, 16-byte aligned
    static constexpr int PATCH = ((CK * PS + 255) / 256) * 256;  // floats per patch buffer (whole 1-KiB DMA pieces)
    static constexpr int SLAB = CK * KW * COB;                // floats per weight stage
    static constexpr int NCC = CIN / CK, NST = NCC * KH;
    static constexpr int OPS_BYTES = (2 * SLAB + 2 * PATCH) * 4;     // the double-buffered operands
    // two-kind layout: the transposed-store tiles of the epilogue (32 x 36 floats per wave) get LDS of their own when it
    // fits, so that a workgroup can run several (sample, slot) items back to back with item i's stores beside item
    // i + 1's first loads and multiplies (see conv_core)
    static constexpr bool EPI_SEP = HET > 0 && OPS_BYTES + NW * 32 * 36 * 4 <= 160 * 1024;
    static constexpr int LDS_BYTES = OPS_BYTES + (EPI_SEP ? NW * 32 * 36 * 4 : 0);
    static constexpr int HP = H / POOL;
    static_assert(CIN % CK == 0 && CK % 2 == 0, "CK");
    // several items per workgroup (EPI_SEP): the next item's first slab and patch are requested into buffers 0 during the
    // current item's LAST stage, which must therefore read buffers 1: an even number of stages, one kernel row
    static_assert(!EPI_SEP || (KH == 1 && NCC % 2 == 0), "items per workgroup need an even stage count");
    static_assert(COUT % COB == 0, "COB");
    static_assert(H % RB == 0, "RB");
    static_assert(SLAB % 256 == 0, "slab must be whole 1-KiB DMA pieces");
    static_assert(RT % POOL == 0, "pool must divide the row tiles of a wave");
    static_assert(MT == 1 || MT == 2 || MT == 4, "MT");
    static_assert(MT * RT * CT * 16 <= 160, "accumulator budget");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(CK * PS * 4 < 65536, "ds_read immediate range");
};

// Identifier of the packed-weight layout a configuration streams: everything pack_weights_kernel /
// pack_freq_weights_kernel take from the configuration (COB, CK, MT, WM) plus the layer geometry, the kernel family and
// the operator orientation.  Two configurations may share a packed operand exactly when these ids are equal, so the id is
// COMPUTED from the constants, never listed by hand.
template <class C>
constexpr int pack_layout_id(int family /* 1 = frequency domain, 2 = direct */, int transposed) {
    static_assert(C::COB / 32 < 16 && C::CK < 128 && C::MT < 8 && C::WM < 16 && C::CIN / 32 < 16 && C::COUT / 32 < 16, "id fields");
    return (family << 29) | ((transposed ? 1 : 0) << 28) | ((C::KH > 1 ? 1 : 0) << 27) | ((C::COB / 32) << 23) |
           (C::CK << 16) | (C::MT << 13) | (C::WM << 9) | ((C::CIN / 32) << 5) | (C::COUT / 32);
}
template <class A, class B>
constexpr bool same_pack_layout() {
    return A::COB == B::COB && A::CK == B::CK && A::MT == B::MT && A::WM == B::WM && A::CIN == B::CIN &&
           A::COUT == B::COUT && A::KH == B::KH && A::KW == B::KW;
}

struct CoreArgs {
    const float* x;        // [N, CIN, H, T]
    const float* wpk;      // packed weights, branch stride = COUT*CIN*KH*KW
    int B, T, N;           // N = nbr*B (x F in the frequency-domain use)
    int ncb;               // column blocks
    int dil, F;            // DILMAX > 0 only: tap spacing, number of frequency slots (weights indexed by (branch, f))
    int ipw;               // two-kind layout with EPI_SEP: consecutive samples per workgroup (>= 1; the grid shrinks accordingly)
    int slot_major;        // DILMAX > 0: walk the (sample, slot) items slot-major (see conv_core)
    // split of the input-channel loop over `ks` workgroups (0 / 1: none): the configuration's CIN is then ONE part's channels,
    // `cin_full` the tensor's (x, packed weights); part k writes its raw sums to dx + k * N * COUT * H * T (EPI 1, no mask,
    // no unpool) and conv_split_finish_kernel adds the parts in order and applies the layer's epilogue
    int ks, cin_full;
    int kparts;            // consecutive CIN-channel parts one workgroup runs through (>= 1): the split is cin_full / (CIN * kparts) ways
    // EPI 0
    Ptr2 bias;
    float* y;
    uint8_t* argmax;
    uint32_t thresh;
    float inv_keep;
    uint32_t seed;
    int layer;
    // EPI 1
    float* dx;
    const float* xmask;
    const uint8_t* prev_argmax;
    int prev_pool;
};

// Input patch [CK][PR][PCW] (+pad) of the workgroup's tile.  The tile is fixed for the whole kernel, so each wave
// computes the per-lane source BYTE offsets of ITS DMA instructions once (patch_offsets: BUF_OOB = outside the image or
// the padding -> the hardware range check moves zeros); staging a ci-chunk is then a buffer descriptor rebuilt from
// scalars (base = the chunk's first channel) and one DMA instruction per piece -- nothing on the vector pipe, which in
// the short-stage layouts (cv5 / cv6: 20 MFMAs per wave and stage) paid 14 address / select instructions per stage for
// the per-lane 64-bit pointers and zero-page selects this replaced.
// NWL = number of waves that issue the DMA pieces (all of them, or only the youngest wave of every SIMD: see conv_core)
template <class C>
struct PatchPlan {
    // first loader wave (round 3's 11-wave cv3 backward-data layout measured 2 % slower with four loaders: NW % 4 != 0 -> all load)
    static constexpr int LW0 = C::HET ? C::HET : (C::DILMAX > 0 && C::NW >= 8 && C::NW % 4 == 0) ? C::NW - 4 : 0;
    static constexpr int NWL = C::NW - LW0;
    static constexpr int NPI = (C::PATCH / 64 + NWL - 1) / NWL;       // 4-byte pieces per loader wave (16-byte mode uses a quarter)
    unsigned voff[NPI];
    bool wide;                                                         // 16-byte DMA pieces (wave-uniform)
};

// pw_al = image columns between the patch origin and the tile origin (multiple of 4, >= the padding pw)
template <class C>
__device__ __forceinline__ void patch_offsets(PatchPlan<C>& pl, const float* xn, int h0, int w0, int T, int wid, int lane,
                                              int pw_al, int live_cols) {
    pl.wide = ((T & 3) == 0) && ((reinterpret_cast<uintptr_t>(xn) & 15) == 0);
#pragma unroll
    for (int i = 0; i < PatchPlan<C>::NPI; ++i) {
        const int L = ((wid + i * PatchPlan<C>::NWL) * 64 + lane) * (pl.wide ? 4 : 1);   // wid = loader wave index
        const int ci = L / C::PS;
        const int rem = L - ci * C::PS;
        const int row = rem / C::PCW;
        const int col = rem - row * C::PCW;
        const int hh = h0 - C::PH + row, ww = w0 - pw_al + col;
        // T % 4 == 0 and a 4-aligned origin: a 16-byte piece is entirely inside or entirely outside the image
        const bool ok = (L < C::CK * C::PS) && (row < C::PR) && (col < live_cols) && hh >= 0 && hh < C::H && ww >= 0 && ww < T;
        pl.voff[i] = ok ? (unsigned)(((ci * C::H + hh) * T + ww) * 4) : BUF_OOB;
    }
}

template <class C>
__device__ __forceinline__ void load_patch(const PatchPlan<C>& pl, const float* xn, int cc, int T, float* lds_patch,
                                           int wid) {
    constexpr int NWL = PatchPlan<C>::NWL;
    const float* base = xn + (size_t)cc * C::CK * C::H * T;
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(base, (unsigned)((size_t)C::CK * C::H * T * sizeof(float)));
    if (pl.wide) {
        const unsigned dst = lds_addr(lds_patch) + wid * 1024;
#pragma unroll
        for (int i = 0; i < (PatchPlan<C>::NPI + 3) / 4; ++i) {
            const int q = wid + i * NWL;
            if (q < C::PATCH / 256) blds16a(rs, pl.voff[i], 0, dst + i * NWL * 1024);
        }
    } else {
        const unsigned dst = lds_addr(lds_patch) + wid * 256;
#pragma unroll
        for (int i = 0; i < PatchPlan<C>::NPI; ++i) {
            const int q = wid + i * NWL;
            if (q < C::PATCH / 64) blds4a(rs, pl.voff[i], dst + i * NWL * 256);
        }
    }
}

// one weight slab (SLAB floats, contiguous thanks to the packing) in 1-KiB pieces: descriptor base = the slab, the piece
// number goes into the scalar offset, the lane's 16 bytes into ONE constant vector offset
template <class C>
__device__ __forceinline__ void load_slab(const float* wslab, float* lds_slab, int wid, int lane) {
    constexpr int NI = C::SLAB / 256;
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(wslab, (unsigned)(C::SLAB * sizeof(float)));
    const unsigned dst = lds_addr(lds_slab);
    for (int q = wid; q < NI; q += PatchPlan<C>::NWL) blds16a(rs, (unsigned)lane * 16u, (unsigned)q * 1024u, dst + q * 1024);
}

// Epilogue helper: one 32x32 accumulator tile -> global rows of `ld` floats, through a per-wave LDS tile so that every
// lane writes 16 bytes (four consecutive columns): 4 store instructions per tile instead of 16.  Epilogues run with the
// matrix pipes idle (one workgroup per CU), so their instruction count is paid in full.  Requirements: ld % 4 == 0,
// 16-byte aligned base, col0 % 4 == 0; columns >= ncols are skipped.  tb: this wave's 32 x ETS floats of LDS.
constexpr int ETS = 36;
__device__ __forceinline__ void store_tile_t(float* tb, const f32x16& acc, float* __restrict__ dst0, size_t ld, int col0,
                                             int ncols, int lane) {
    const int lane_j = lane & 31, lane_h = lane >> 5, rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int k = 0; k < 16; ++k) tb[acc_row(k, lane_h) * ETS + lane_j] = acc[k];
    // same-wave LDS traffic is ordered: no barrier between the writes above and the reads below
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = rr + 8 * q;
        const f32x4 v = *reinterpret_cast<const f32x4*>(tb + row * ETS + c4);
        if (col0 + c4 < ncols) *reinterpret_cast<f32x4*>(dst0 + (size_t)row * ld + col0 + c4) = v;
    }
}

// One K stage.  Operands are read through LDS BYTE ADDRESSES held in registers for the whole kernel: `a_addr` = this
// wave's A base in the stage's slab buffer, `tap[s]` = its B base for kernel column s in patch buffer 0 (the tap spacing
// is a run-time value in the frequency-domain use, so every tap needs its own base); everything else -- channel pair,
// row / column tile, and with BOFF the patch buffer's parity -- is the read instruction's immediate offset.  With the
// bases recomputed per stage from pointers (the previous form) a stage paid 6 (cv6) to 36 (cv4) vector adds on the pipe
// its MFMAs need.
__device__ __forceinline__ f32x4 lds_ld4(unsigned lds_byte) {
    return *(const __attribute__((address_space(3))) f32x4*)(size_t)lds_byte;
}
__device__ __forceinline__ f32x2 lds_ld2(unsigned lds_byte) {
    return *(const __attribute__((address_space(3))) f32x2*)(size_t)lds_byte;
}
// NCT <= CT: the wave's LIVE column tiles (the tiles beyond the row in the last column block are neither read nor multiplied)
template <class C, int BOFF, int NCT = C::CT>
__device__ __forceinline__ void compute_stage(unsigned a_addr, const unsigned (&tap)[C::KW],
                                              f32x16 (&acc)[C::MT][C::RT][C::CT]) {
    // steps k = (channel pair, kernel column); the operands of step k + 1 are requested before the MFMAs of step k
    constexpr int K2 = (C::CK / 2) * C::KW;
    float a[2][C::MT], b[2][C::RT][C::CT];
    auto fetch = [&](int k, float (&av)[C::MT], float (&bv)[C::RT][C::CT]) {
        const int cp = k / C::KW, s = k % C::KW;
        const unsigned ap = a_addr + k * (C::WM * 64 * C::MT) * 4;
        if constexpr (C::MT == 4) {
            const f32x4 v = lds_ld4(ap);
            av[0] = v[0]; av[1] = v[1]; av[2] = v[2]; av[3] = v[3];
        } else if constexpr (C::MT == 2) {
            const f32x2 v = lds_ld2(ap);
            av[0] = v[0]; av[1] = v[1];
        } else {
            av[0] = lds_ld(ap);
        }
#pragma unroll
        for (int rt = 0; rt < C::RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                bv[rt][ct] = lds_ld(tap[s] + BOFF + (2 * cp * C::PS + rt * C::PCW + ct * 32) * 4);
    };
    fetch(0, a[0], b[0]);
#pragma unroll
    for (int k = 0; k < K2; ++k) {
        if (k + 1 < K2) fetch(k + 1, a[(k + 1) & 1], b[(k + 1) & 1]);
#pragma unroll
        for (int m = 0; m < C::MT; ++m)
#pragma unroll
            for (int rt = 0; rt < C::RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    acc[m][rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k & 1][m], b[k & 1][rt][ct], acc[m][rt][ct], 0, 0, 0);
    }
}

// HET layout (see ConvCfg): one stage for a wave of type A (TA: 4 channel tiles x 1 column tile) or B (1 x 3).  The
// wave kind is a template argument: the whole K loop is instantiated once per kind (a run-time branch inside the loop
// makes the two kinds' accumulators meet in a copy -- 24 v_mov_b64 per stage).
template <class C, int BOFF, bool TA>
__device__ __forceinline__ void compute_stage_het(unsigned a_addr, const unsigned (&tap)[C::KW], f32x16 (&acc)[4]) {
    if constexpr (TA) {
#pragma unroll
        for (int cp = 0; cp < C::CK / 2; ++cp)
#pragma unroll
            for (int s = 0; s < C::KW; ++s) {
                const f32x4 a = lds_ld4(a_addr + (cp * C::KW + s) * 1024);
                const float b = lds_ld(tap[s] + BOFF + 2 * cp * C::PS * 4);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b, acc[m], 0, 0, 0);
            }
    } else {
        // Type B waves (one channel tile x three column tiles: four operand reads per three MFMAs) are the youngest
        // wave of their SIMD and run LAST in every stage (cycle stamps: the three waves of a SIMD multiply one after the
        // other in age order: done after 10.1 k / 19.1 k / 26.8 k cycles of a 27.3 k-cycle stage), alone on the pipe, so
        // their own read latency is exposed: a whole channel pair's operands (KW k-steps, 4 KW registers) are requested
        // ahead of the pair being multiplied; a type B wave has the registers (48 accumulators against type A's 64).
        float a[2][C::KW], b[2][C::KW][3];
        auto fetch = [&](int cp, float (&av)[C::KW], float (&bv)[C::KW][3]) {
#pragma unroll
            for (int s = 0; s < C::KW; ++s) {
                av[s] = lds_ld(a_addr + (cp * C::KW + s) * 1024);
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) bv[s][ct] = lds_ld(tap[s] + BOFF + (2 * cp * C::PS + ct * 32) * 4);
            }
        };
        fetch(0, a[0], b[0]);
#pragma unroll
        for (int cp = 0; cp < C::CK / 2; ++cp) {
            if (cp + 1 < C::CK / 2) fetch(cp + 1, a[(cp + 1) & 1], b[(cp + 1) & 1]);
#pragma unroll
            for (int s = 0; s < C::KW; ++s)
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cp & 1][s], b[cp & 1][s][ct], acc[ct], 0, 0, 0);
        }
    }
}

template <class C>
__global__ void __launch_bounds__(C::NT) conv_core(CoreArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto slab = [&](int i) -> float* { return smem + (i & 1) * C::SLAB; };
    auto patch = [&](int i) -> float* { return smem + 2 * C::SLAB + (i & 1) * C::PATCH; };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane_j = lane & 31, lane_h = lane >> 5;
    const int wc = wid % C::WC, wr = (wid / C::WC) % C::WR, wm = wid / (C::WC * C::WR);

    // block -> (column block, row block, co block, sample)
    int bid = blockIdx.x;
    const int cb = bid % A.ncb;
    bid /= A.ncb;
    const int rb = bid % (C::H / C::RB);
    bid /= (C::H / C::RB);
    const int cob = bid % (C::COUT / C::COB);
    bid /= (C::COUT / C::COB);
    const int ksn = (!C::HET && A.ks > 1) ? A.ks : 1;
    const int ksi = bid % ksn;
    bid /= ksn;
    const int cinf = (!C::HET && A.cin_full > 0) ? A.cin_full : C::CIN;
    // several consecutive samples per workgroup (two-kind layout, host guarantees one block per sample then)
    const int ipw = (C::HET && C::EPI_SEP && A.ipw > 1) ? A.ipw : 1;
    if constexpr (C::DILMAX > 0) {
        // Frequency-domain launches: item = (sample, slot), and every item of a slot streams the SAME weight operator (cv4:
        // 590 KB per slot, 113 MB per branch).  In item order a CU's L2 sees 24 different operators per sample and meets each
        // again a whole sample pass (14 MB per XCD) later; walked slot-major in groups of eight slots -- block q -> slot
        // 8 (q / 8 Ns) + q % 8, sample (q / 8) % Ns -- the eight XCDs (block id mod 8) each keep ONE operator in their 4 MB L2
        // while the Ns samples pass through it.
        if (ipw == 1 && A.slot_major && (A.F & 7) == 0) {
            const int Ns = A.N / A.F, q = bid;
            const int grp = q / (8 * Ns), r = q - grp * 8 * Ns;
            bid = (r >> 3) * A.F + grp * 8 + (r & 7);
        }
    }
    int n = bid * ipw;
    if constexpr (C::DILMAX > 0) {
        // slot 2 = the third Karatsuba part of frequency slot 0, which holds the two REAL bins (DC, Nyquist): there is no such
        // product, nobody reads its output plane
        if (ipw == 1 && n % A.F == 2) return;
    }
    const int nitems = ipw > 1 ? (A.N - n < ipw ? A.N - n : ipw) : 1;
    auto branch_of = [&](int nn) { return C::DILMAX ? nn / (A.B * A.F) : nn / A.B; };
    const int br = branch_of(n);
    const int dil = C::DILMAX ? A.dil : 1;
    const int pw = C::DILMAX ? (C::KW / 2) * dil : C::PW;
    const int h0 = rb * C::RB, w0 = cb * C::WB;
    const int T = A.T;

    const int kp = (ksn > 1 && A.kparts > 1) ? A.kparts : 1;       // CIN-channel parts this workgroup accumulates
    int part = ksi * kp;                                           // the current one
    auto x_of = [&](int nn) { return A.x + ((size_t)nn * cinf + (size_t)part * C::CIN) * C::H * T; };
    auto w_of = [&](int nn) {
        const int b2 = branch_of(nn), widx = C::DILMAX ? b2 * A.F + nn % A.F : b2;
        return A.wpk + (size_t)widx * C::COUT * cinf * C::KH * C::KW + (size_t)cob * (cinf / C::CK * C::KH) * C::SLAB +
               (size_t)part * C::NST * C::SLAB;
    };
    const float* xn = x_of(n);
    const float* wbase = w_of(n);
    const float* xn_next = xn;          // the item after the current one (ipw > 1)
    const float* wbase_next = wbase;
    bool has_next = false;

    f32x16 acc[C::MT][C::RT][C::CT];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int rt = 0; rt < C::RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < C::CT; ++ct)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[m][rt][ct][k] = 0.f;

    // Frequency-domain launches (8 or 12 waves, 2-3 per SIMD): the last-dispatched wave of every SIMD loses instruction
    // arbitration to its older partners in every stage (priority, then age); ONE static priority raise for that last
    // third removes its start-of-stage penalty (dominant kernel 2.385 -> 2.345 ms, 0.845 -> 0.857 of peak; raising the
    // OLDER third instead, or the same in the backward-weights kernels, measured nothing).
    if constexpr (C::DILMAX > 0 && C::NW >= 8) {
        if (wid >= C::NW - 4) __builtin_amdgcn_s_setprio(1);
    }
    const bool type_a = !C::HET || wid < C::HET;       // HET: the last four waves own one channel tile x 3 column tiles
    const int het_m0 = wid - C::HET;
    const int pw_al = (pw + 3) & ~3;   // patch origin = tile origin - pw_al (16-byte aligned); reads skip pw_al - pw
    const int abase = C::HET ? lane * 4 + (type_a ? 0 : het_m0) : wm * (64 * C::MT) + lane * C::MT;
    const int bbase = C::HET ? lane_h * C::PS + (type_a ? wid : C::HET) * 32 + lane_j + (pw_al - pw)
                             : lane_h * C::PS + (wr * C::RT) * C::PCW + wc * C::CT * 32 + lane_j + (pw_al - pw);

    // Who issues the DMA pieces: in the frequency-domain layouts of 8 or 12 waves only the last four -- the youngest wave of
    // every SIMD (in the two-kind layout: the type B waves), which multiplies LAST in a stage (the waves of a SIMD run in
    // age order) and would otherwise wait; the older waves start their MFMAs right after the barrier.  Elsewhere every
    // wave issues its share.
    const bool loader = wid >= PatchPlan<C>::LW0;        // (two-kind layout: exactly the type B waves)
    const int lw = wid - PatchPlan<C>::LW0;              // loader wave index
    PatchPlan<C> plan;
    if (loader) {
        patch_offsets<C>(plan, xn, h0, w0, T, lw, lane, pw_al,
                         (C::DILMAX ? C::WB + (C::KW - 1) * dil : C::WB + C::KW - 1) + (pw_al - pw));
        load_patch<C>(plan, xn, 0, T, patch(0), lw);
        load_slab<C>(wbase, slab(0), lw, lane);
    }

    // operand base addresses (see compute_stage): A per slab buffer, B per kernel column in patch buffer 0
    unsigned a_addr[2], tap[C::KW];
    a_addr[0] = lds_addr(slab(0)) + abase * 4;
    a_addr[1] = lds_addr(slab(1)) + abase * 4;
#pragma unroll
    for (int s = 0; s < C::KW; ++s) tap[s] = lds_addr(patch(0)) + (bbase + (C::DILMAX ? s * dil : s)) * 4;
    // opaque: the registers hold the FULL addresses (the compiler would otherwise keep the variable part and re-add the
    // buffer's constant offset, which is beyond the reads' 16-bit immediate in the large-slab layouts, per use)
    asm volatile("" : "+v"(a_addr[0]), "+v"(a_addr[1]));
#pragma unroll
    for (int s = 0; s < C::KW; ++s) asm volatile("" : "+v"(tap[s]));
    constexpr int PBYTES = C::PATCH * 4;
    // a wave whose column tiles all lie beyond the row (the 256-column cv2 layout at P = 960 = 3 x 256 + 192: the fourth
    // wave's 64 columns are padding) skips its multiplies -- it still loads and meets the barriers -- and leaves its SIMD's
    // matrix pipe to the co-resident workgroups
    const bool idle = !C::HET && w0 + wc * C::CT * 32 >= T;
    // the frequency-domain layouts with several column tiles per wave also drop the DEAD tiles of a partly live wave (cv3
    // backward-data: 11 column tiles in two blocks of 2 x 3): wave-uniform, the K loop is instantiated per live count
    const int live_ct = C::HET ? C::CT : (T - (w0 + wc * C::CT * 32) + 31) / 32;

    // one stage of the K loop; PAR >= 0: buffer parity known at compile time (an immediate), -1: run-time parity;
    // TAC: the wave kind of the two-kind layout
    auto stage = [&](auto PARC, auto TAC, int st, int cc, int r) {
        constexpr int PAR = decltype(PARC)::value;
        constexpr bool TA = decltype(TAC)::value;
        // only the waves that issued DMA pieces wait for them (vmcnt also counts a wave's global STORES: a non-loader wave
        // that has just stored an item's tiles would otherwise sit here until its stores are acknowledged)
        if (C::HET ? !TA : loader) wait_vm0();
        __syncthreads();
        if (C::HET ? !TA : loader) {
            if (st + 1 < C::NST) {
                load_slab<C>(wbase + (size_t)(st + 1) * C::SLAB, slab(st + 1), lw, lane);
                if (r == C::KH - 1) load_patch<C>(plan, xn, cc + 1, T, patch(cc + 1), lw);
            } else if (has_next) {          // the next item's first stage (NST is even: buffers 0 are free again)
                load_slab<C>(wbase_next, slab(0), lw, lane);
                load_patch<C>(plan, xn_next, 0, T, patch(0), lw);
            }
        }
        if (idle) return;
        if constexpr (PAR >= 0) {
            if constexpr (C::HET) {
                f32x16(&hacc)[4] = *reinterpret_cast<f32x16(*)[4]>(&acc[0][0][0]);
                compute_stage_het<C, PAR * PBYTES, TA>(a_addr[PAR], tap, hacc);
            } else {
                compute_stage<C, PAR * PBYTES, decltype(TAC)::value>(a_addr[PAR], tap, acc);
            }
        } else {
            unsigned tr[C::KW];
            const unsigned boff = (cc & 1) * PBYTES + r * C::PCW * 4;
#pragma unroll
            for (int s = 0; s < C::KW; ++s) tr[s] = tap[s] + boff;
            if constexpr (C::HET) {
                f32x16(&hacc)[4] = *reinterpret_cast<f32x16(*)[4]>(&acc[0][0][0]);
                compute_stage_het<C, 0, TA>(a_addr[st & 1], tr, hacc);
            } else {
                compute_stage<C, 0>(a_addr[st & 1], tr, acc);
            }
        }
    };
    auto k_stages = [&](auto TAC) {
        if constexpr (C::KH == 1 && C::NCC % 2 == 0) {
            // the frequency-domain use (and cv7 / cv8): stage = channel chunk, two per trip, so that the double
            // buffers' parity is part of the immediates
            for (int cc = 0; cc < C::NCC; cc += 2) {
                stage(std::integral_constant<int, 0>{}, TAC, cc, cc, 0);
                stage(std::integral_constant<int, 1>{}, TAC, cc + 1, cc + 1, 0);
            }
        } else {
            int st = 0;
            for (int cc = 0; cc < C::NCC; ++cc)
                for (int r = 0; r < C::KH; ++r, ++st) stage(std::integral_constant<int, -1>{}, TAC, st, cc, r);
        }
    };
    if constexpr (C::HET) {
        // tile t of this wave: type A -> (channel tile t, column tile wid); type B -> (channel tile wid-8, column tile 8+t)
        const int pp = A.prev_pool;
        const bool plain = A.xmask == nullptr && pp == 1 && (T & 3) == 0 && (reinterpret_cast<uintptr_t>(A.dx) & 15) == 0;
        // plain stores (the frequency-domain use): 16 bytes per lane through store_tile_t.  With LDS of its own (EPI_SEP) a
        // wave stores as soon as ITS multiplies are done -- the waves of a SIMD finish one after the other -- and the
        // workgroup goes on with the next sample: the stores and the next first loads ride beside the other waves' MFMAs
        // instead of costing a prologue and an epilogue per sample with the matrix pipes idle.
        auto store_plain = [&](int nn) {
            if constexpr (!C::EPI_SEP) __syncthreads();            // the tiles reuse the operand buffers
            float* tb = smem + (C::EPI_SEP ? C::OPS_BYTES / 4 : 0) + wid * (32 * ETS);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!type_a && t == 3) break;
                const int co0 = cob * C::COB + (type_a ? t : het_m0) * 32;
                store_tile_t(tb, acc[t][0][0], A.dx + ((size_t)nn * C::COUT + co0) * T, (size_t)T,
                             w0 + (type_a ? wid : C::HET + t) * 32, T, lane);
            }
        };
        auto items = [&](auto TAC) {
            for (int it = 0; it < nitems; ++it) {
                has_next = it + 1 < nitems;
                if (has_next) { xn_next = x_of(n + 1); wbase_next = w_of(n + 1); }
                k_stages(TAC);
                if (!plain) break;                                 // (nitems == 1: the host only batches plain launches)
                store_plain(n);
                if (has_next) {
                    ++n; xn = xn_next; wbase = wbase_next;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int k = 0; k < 16; ++k) acc[t][0][0][k] = 0.f;
                }
            }
        };
        if (type_a) items(std::true_type{});       // the barriers inside pair up across the two kinds: same trip counts
        else items(std::false_type{});
        if (plain) return;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (!type_a && t == 3) break;
            const int co0 = cob * C::COB + (type_a ? t : het_m0) * 32;
            const int col = w0 + (type_a ? wid : C::HET + t) * 32 + lane_j;
            if (col >= T) continue;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int co = co0 + acc_row(k, lane_h);
                const size_t o = ((size_t)n * C::COUT + co) * T + col;
                float g = acc[t][0][0][k];
                if (A.xmask) g = (A.xmask[o] != 0.f) ? g * A.inv_keep : 0.f;
                if (pp > 1) {
                    const int am = A.prev_argmax[o];
                    const size_t base = ((size_t)n * C::COUT + co) * pp * T + col;
                    for (int a = 0; a < pp; ++a) A.dx[base + (size_t)a * T] = (a == am) ? g : 0.f;
                } else {
                    A.dx[o] = g;
                }
            }
        }
        return;
    } else if constexpr (C::DILMAX > 0 && C::KH == 1 && C::NCC % 2 == 0 && C::CT > 1) {
        // (TAC carries the live column-tile count here; the barriers inside pair up: same trip counts)
        auto by_live = [&](auto self, auto NC) {
            constexpr int NL = decltype(NC)::value;
            if constexpr (NL >= C::CT) {
                k_stages(std::integral_constant<int, C::CT>{});      // (also the idle waves: they only load and meet the barriers)
            } else {
                if (live_ct == NL) k_stages(std::integral_constant<int, NL>{});
                else self(self, std::integral_constant<int, NL + 1>{});
            }
        };
        by_live(by_live, std::integral_constant<int, 1>{});
    } else if constexpr (C::KH == 1 && C::NCC % 2 == 0) {
        // a K-split workgroup runs through its kp parts like the two-kind layout through its items: the next part's first slab
        // and patch are requested during the current part's last stage (NST is even: buffers 0 are free again)
        for (int q = 0; q < kp; ++q) {
            has_next = q + 1 < kp;
            if (has_next) { ++part; xn_next = x_of(n); wbase_next = w_of(n); }
            k_stages(std::integral_constant<int, C::CT>{});
            if (has_next) { xn = xn_next; wbase = wbase_next; }
        }
        has_next = false;
    } else {
        k_stages(std::integral_constant<int, C::CT>{});
    }

    // ------------------------------------------------------------------ epilogue
    if constexpr (C::EPI == 0) {
        const uint32_t lseed = layer_seed(A.seed, A.layer, br);
        const int b_in_br = n - br * A.B;
        const float* bias = A.bias.p[br];
#pragma unroll
        for (int m = 0; m < C::MT; ++m) {
            const int co0 = cob * C::COB + (wm * C::MT + m) * 32;
#pragma unroll
            for (int ct = 0; ct < C::CT; ++ct) {
                const int col = w0 + (wc * C::CT + ct) * 32 + lane_j;
                if (col >= T) continue;
#pragma unroll
                for (int g = 0; g < C::RT / C::POOL; ++g) {
                    const int hp = (h0 + wr * C::RT) / C::POOL + g;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int co = co0 + acc_row(k, lane_h);
                        float best = acc[m][g * C::POOL][ct][k];
                        int am = 0;
#pragma unroll
                        for (int a = 1; a < C::POOL; ++a) {
                            const float v = acc[m][g * C::POOL + a][ct][k];
                            if (pool_takes(v, best)) { best = v; am = a; }
                        }
                        float v = best + bias[co];
                        v = relu_nan(v);
                        const size_t o = (((size_t)n * C::COUT + co) * C::HP + hp) * T + col;
                        if (A.thresh) {
                            const uint64_t idx = (((uint64_t)b_in_br * C::COUT + co) * C::HP + hp) * T + col;
                            v = drop_apply(drop_keep(lseed, idx, A.thresh), v, A.inv_keep);
                        }
                        A.y[o] = v;
                        if (C::POOL > 1 && A.argmax) A.argmax[o] = (uint8_t)am;
                    }
                }
            }
        }
    } else {
        const int pp = A.prev_pool;
        float* const dxp = A.dx + (size_t)ksi * A.N * C::COUT * C::H * T;      // (a split part's own slab)
        if constexpr (C::LDS_BYTES >= C::NW * 32 * ETS * 4) {
            if (A.xmask == nullptr && pp == 1 && (T & 3) == 0 && (reinterpret_cast<uintptr_t>(dxp) & 15) == 0) {
                __syncthreads();                           // every wave is done with the last stage's operands
                float* tb = smem + wid * (32 * ETS);
#pragma unroll
                for (int m = 0; m < C::MT; ++m)
#pragma unroll
                    for (int rt = 0; rt < C::RT; ++rt)
#pragma unroll
                        for (int ct = 0; ct < C::CT; ++ct) {
                            const int co0 = cob * C::COB + (wm * C::MT + m) * 32;
                            const int row = h0 + wr * C::RT + rt;
                            store_tile_t(tb, acc[m][rt][ct], dxp + (((size_t)n * C::COUT + co0) * C::H + row) * T,
                                         (size_t)C::H * T, w0 + (wc * C::CT + ct) * 32, T, lane);
                        }
                return;
            }
        }
#pragma unroll
        for (int m = 0; m < C::MT; ++m) {
            const int co0 = cob * C::COB + (wm * C::MT + m) * 32;
#pragma unroll
            for (int ct = 0; ct < C::CT; ++ct) {
                const int col = w0 + (wc * C::CT + ct) * 32 + lane_j;
                if (col >= T) continue;
#pragma unroll
                for (int rt = 0; rt < C::RT; ++rt) {
                    const int row = h0 + wr * C::RT + rt;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int co = co0 + acc_row(k, lane_h);
                        const size_t o = (((size_t)n * C::COUT + co) * C::H + row) * T + col;
                        float g = acc[m][rt][ct][k];
                        if (A.xmask) g = (A.xmask[o] != 0.f) ? g * A.inv_keep : 0.f;
                        if (pp > 1) {
                            const int am = A.prev_argmax[o];
                            const size_t base = (((size_t)n * C::COUT + co) * C::H + row) * pp * T + col;
                            for (int a = 0; a < pp; ++a) dxp[base + (size_t)a * T] = (a == am) ? g : 0.f;
                        } else {
                            dxp[o] = g;
                        }
                    }
                }
            }
        }
    }
}

// Pack OIHW weights into the stage-slab order conv_core streams:
//   [branch][cob][stage = cc*KH + r][cp][s][wm][lane_h][i][m]
// transposed != 0 packs the backward-data operator: Wc[co_c][ci_c][r][s] = W[ci_c][co_c][KH-1-r][KW-1-s].
static __global__ void pack_weights_kernel(Ptr2 w, float* __restrict__ dst, int cin_c, int cout_c, int kh, int kw, int cob_sz,
                                    int ck, int mt, int wm_n, int transposed, long long per_branch, int nbr) {
    const long long total = per_branch * nbr;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total;
         o += (long long)gridDim.x * blockDim.x) {
        const int br = (int)(o / per_branch);
        long long q = o % per_branch;
        const int m = (int)(q % mt); q /= mt;
        const int i = (int)(q % 32); q /= 32;
        const int h = (int)(q % 2); q /= 2;
        const int wm = (int)(q % wm_n); q /= wm_n;
        const int s = (int)(q % kw); q /= kw;
        const int cp = (int)(q % (ck / 2)); q /= (ck / 2);
        const int r = (int)(q % kh); q /= kh;
        const int ncc = cin_c / ck;
        const int cc = (int)(q % ncc); q /= ncc;
        const int cob = (int)q;
        const int co = cob * cob_sz + (wm * mt + m) * 32 + i;
        const int ci = cc * ck + 2 * cp + h;
        float v;
        if (!transposed) {
            v = w.p[br][(((size_t)co * cin_c + ci) * kh + r) * kw + s];
        } else {
            // original tensor is [cout_o = cin_c][cin_o = cout_c][kh][kw]
            v = w.p[br][(((size_t)ci * cout_c + co) * kh + (kh - 1 - r)) * kw + (kw - 1 - s)];
        }
        dst[o] = v;
    }
}

// =====================================================================================================
//                                        backward-weights core
// =====================================================================================================
template <int CIN_, int COUT_, int KH_, int KW_, int H_, int MT_, int NS_, int WMW_, int WQ_, int R_, int DILMAX_ = 0,
          int WCI_ = 1, int CW_ = 64>
struct WgCfg {
    static constexpr int WCI = WCI_;         // 32-channel input tiles per workgroup (one per wave group)
    static constexpr int CIN = CIN_, COUT = COUT_, KH = KH_, KW = KW_, H = H_;
    static constexpr int DILMAX = DILMAX_;   // > 0: frequency-domain use, taps `dil` apart (see ConvCfg)
    static constexpr int MT = MT_, NS = NS_, WMW = WMW_, WQ = WQ_, R = R_;
    static constexpr int NW = WMW * WQ * WCI, NT = NW * 64;
    static constexpr int COB = WMW * MT * 32;
    static constexpr int CIB = 32 * WCI;
    static constexpr int G = WQ * NS, Q = KH * KW, NQG = (Q + G - 1) / G;
    static constexpr int CW = CW_;                           // pixels per K-tile: one image row segment (32 or 64)
    static constexpr int LPR = CW / 4;                       // 16-byte chunks (= DMA lanes) per dZ row
    static constexpr int RPI = 64 / LPR;                     // dZ rows staged by one 16-byte DMA instruction
    static constexpr int ZKD = 64 / CW;                      // swizzle key = (co / ZKD) & (LPR - 1): conflict-free b128 reads
    static constexpr int PH = KH / 2, PW = KW / 2;
    static constexpr int RSPAN = KH == 1 ? 1 : (KW - 1 + G - 1) / KW + 1;  // kernel rows a workgroup's shifts touch
    static constexpr int PRW = RSPAN, PCW = CW + (KW - 1) * (DILMAX ? DILMAX : 1);
    static constexpr int PS = (PRW * PCW) | 1;               // odd plane stride: lanes (= channels) hit distinct banks
    static constexpr int ZBUF = COB * CW;                    // dZ tile [COB][CW], 16-byte chunks XOR-swizzled by the key above
    static constexpr int XBUF = ((CIB * PS + 64 * NW - 1) / (64 * NW)) * (64 * NW);   // whole pieces for every wave
    static constexpr int NXI = XBUF / 64 / NW;               // X-patch DMA instructions per wave
    static constexpr int NZI = (COB / RPI + NW - 1) / NW;    // dZ-tile DMA instructions per wave (16-byte pieces)
    static constexpr int LDS_BYTES = 2 * (ZBUF + XBUF) * 4;
    static constexpr int WPE = (NW == 4 && 2 * LDS_BYTES <= 160 * 1024) ? 2 : 1;   // 4-wave workgroups: two per CU
    static_assert(R == 1, "one image row per K-tile");
    static_assert(CW == 32 || CW == 64, "K-tile width");
    static_assert(CIN % CIB == 0 && COUT % COB == 0, "tiling");
    static_assert(COB % 4 == 0 && MT * NS * 16 <= 160, "accumulator budget");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

struct WgArgs {
    const float* dz;   // [N, COUT, H, T]
    const float* x;    // [N, CIN, H, T]
    float* slab;       // [nbr][KS][Q][COUT][CIN]
    int B, T, ncb;     // column blocks of 64
    int KS, tiles_per_split, tiles_total;   // per branch
    int dil, F;        // DILMAX > 0 only: tap spacing; frequency slots (the "branch" grid index is then branch*F + f)
};

// Stage one K-tile (image row h0, columns w0..w0+63 of sample n):
//   dZ  [COB][64]            16-byte chunk c of channel co is stored at chunk position c ^ (co & 15)
//   X   [32][PRW][PCW] (+pad) rows h0+r0-PH .., columns w0-PW .., zero outside the image
// Both go through buffer descriptors rebuilt per tile from scalars (base = tile origin): the per-lane byte offsets
// xvoff / zvoff are tile independent and computed once per kernel, so an interior tile costs one buffer_load..lds plus an
// M0 update per 256-byte / 1-KiB piece and NO vector ALU work (the previous per-lane 64-bit pointer + zero-page select
// version spent ~900 scalar/vector instructions per stage, a fifth of the stage, with the matrix pipes idle).
// xrc: (row << 16 | col) of the lane's element inside the patch, for the tiles that touch the image border.
template <class C>
__device__ __forceinline__ void wg_load_tile(const WgArgs& A, int br, int cob, int cit, int r0, int tile, float* zb,
                                             float* xb, int wid, int lane, const unsigned (&xvoff)[C::NXI],
                                             const int (&xrc)[C::NXI], const unsigned (&zvoff)[C::NZI], int pcw) {
    const int T = A.T;
    const int cbk = tile % A.ncb;
    const int h0 = (tile / A.ncb) % C::H;
    const int b = tile / (A.ncb * C::H);
    const int n = C::DILMAX ? ((br / A.F) * A.B + b) * A.F + br % A.F : br * A.B + b;
    const int w0 = cbk * C::CW;
    const float* zrow = A.dz + (((size_t)n * C::COUT + (size_t)cob * C::COB) * C::H + h0) * T + w0;
    const size_t zplane = (size_t)C::H * T;
    const bool fastz = ((T & 3) == 0) && (w0 + C::CW <= T) && ((reinterpret_cast<uintptr_t>(A.dz) & 15) == 0);
    if (fastz) {
        // LPR lanes x 16 B = one row segment; one DMA instruction stages RPI channels (1 KiB, LDS-linear)
        const __amdgpu_buffer_rsrc_t rz = buf_rsrc(zrow, (unsigned)(((C::COB - 1) * zplane + C::CW) * sizeof(float)));
#pragma unroll
        for (int i = 0; i < C::NZI; ++i) {
            const int q = wid + i * C::NW;
            if (q < C::COB / C::RPI) blds16(rz, zvoff[i], zb + q * 256);
        }
    } else {
        for (int q = wid; q < C::COB * C::CW / 64; q += C::NW) {
            const int co = q * C::ZKD + lane / C::CW, l = lane % C::CW;
            const int col = (((l >> 2) ^ ((co / C::ZKD) & (C::LPR - 1))) << 2) + (l & 3);
            const float* src = (w0 + col < T) ? zrow + (size_t)co * zplane + col : g_zero_page;
            glds4(src, zb + q * 64);
        }
    }
    const int hbase = h0 + r0 - C::PH, wbase = w0 - (C::DILMAX ? (C::KW / 2) * A.dil : C::PW);
    const bool interior = hbase >= 0 && hbase + C::PRW <= C::H && wbase >= 0 && wbase + pcw <= T;
    const float* xbase = A.x + ((size_t)n * C::CIN + (size_t)cit * C::CIB) * C::H * T + (ptrdiff_t)hbase * T + wbase;
    const __amdgpu_buffer_rsrc_t rx =
        buf_rsrc(xbase, (unsigned)((((size_t)(C::CIB - 1) * C::H + C::PRW - 1) * T + pcw) * sizeof(float)));
    if (interior) {
        // the lean path (28 of 30 tiles): per piece one M0 update (tile base + constant) and the DMA instruction --
        // nothing on the vector pipe (which the partner workgroup's wave keeps full of MFMAs: every vector instruction
        // issued here waits for a gap between them), no branches, no loop-invariant per-piece constants (27 LDS
        // addresses + 27 `piece < pieces needed at this tap spacing` masks used to sit in scalar registers, half of them
        // spilled to VGPR lanes and fetched back with v_readlane_b32, a vector-pipe instruction per piece).  Pieces past
        // the patch of this tap spacing carry out-of-range offsets and move zeros into the unused tail of the buffer.
        unsigned dst = lds_addr(xb) + wid * 256;
        asm volatile("" : "+s"(dst));                  // opaque per tile: the piece addresses are not loop invariant
#pragma unroll
        for (int i = 0; i < C::NXI; ++i) blds4a(rx, xvoff[i], dst + i * C::NW * 256);
    } else {
        // border tiles (2 of 30): the in-image test per piece.  Tile origin and LDS base opaque, so that nothing of this
        // rare path is computed ahead of the K loop and kept in (spilled) scalar registers through the common one.
        int hb = hbase, wb = wbase;
        unsigned dst = lds_addr(xb) + wid * 256;
        asm volatile("" : "+s"(hb), "+s"(wb), "+s"(dst));
#pragma unroll
        for (int i = 0; i < C::NXI; ++i) {
            const int hh = hb + (xrc[i] >> 16), ww = wb + (xrc[i] & 0xffff);
            const bool ok = (unsigned)hh < (unsigned)C::H && (unsigned)ww < (unsigned)T;
            blds4a(rx, ok ? xvoff[i] : BUF_OOB, dst + i * C::NW * 256);
        }
    }
}

// pixel group of a K-tile before which wave group `grp` issues the next tile's DMA pieces: group 0 at the start, the
// later groups three quarters into their own multiply (halfway for group 1 measured 1 % slower for cv4, a quarter /
// half 2-3 % slower: the waves of a SIMD multiply in age order, a later group's early pieces only compete with the
// older group's reads)
constexpr int wg_stagger_at(int grp, int gsteps) { return grp == 0 ? 0 : (3 * gsteps) / 4; }

template <class C>
__global__ void __launch_bounds__(C::NT, C::WPE) wgrad_core(WgArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto zbuf = [&](int i) -> float* { return smem + (i & 1) * C::ZBUF; };
    auto xbuf = [&](int i) -> float* { return smem + 2 * C::ZBUF + (i & 1) * C::XBUF; };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane_j = lane & 31, lane_h = lane >> 5;
    const int wq = wid % C::WQ, wci = (wid / C::WQ) % C::WCI, wmw = wid / (C::WQ * C::WCI);

    int bid = blockIdx.x;
    const int qg = bid % C::NQG; bid /= C::NQG;
    const int cit = bid % (C::CIN / C::CIB); bid /= (C::CIN / C::CIB);
    const int cob = bid % (C::COUT / C::COB); bid /= (C::COUT / C::COB);
    const int ks = bid % A.KS;
    const int br = bid / A.KS;
    if constexpr (C::DILMAX > 0) {
        if (br % A.F == 2) return;       // (frequency slot 0 has no third Karatsuba part: see conv_core)
    }

    const int q0 = qg * C::G;
    const int r0 = q0 / C::KW;
    const int t_begin = ks * A.tiles_per_split;
    int t_end = t_begin + A.tiles_per_split;
    if (t_end > A.tiles_total) t_end = A.tiles_total;

    f32x16 acc[C::MT][C::NS];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int s = 0; s < C::NS; ++s)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[m][s][k] = 0.f;

    // X-patch geometry: in the frequency-domain use the row width follows the runtime tap spacing, so only the
    // columns that are read get staged (the LDS buffer is sized for DILMAX)
    const int pcw = C::DILMAX ? C::CW + (C::KW - 1) * A.dil : C::PCW;
    const int ps = C::DILMAX ? ((C::PRW * pcw) | 1) : C::PS;
    // per-lane source byte offsets of this wave's DMA instructions, relative to the tile origin (tile independent)
    unsigned xvoff[C::NXI], zvoff[C::NZI];
    int xrc[C::NXI];
#pragma unroll
    for (int i = 0; i < C::NXI; ++i) {
        const int L = (wid + i * C::NW) * 64 + lane;
        const int ci = L / ps;
        const int rem = L - ci * ps;
        const int row = rem / pcw;
        const int col = rem - row * pcw;
        const bool ok = ci < C::CIB && row < C::PRW;
        xvoff[i] = ok ? (unsigned)(((ci * C::H + row) * A.T + col) * 4) : BUF_OOB;
        xrc[i] = (row << 16) | col;
    }
#pragma unroll
    for (int i = 0; i < C::NZI; ++i) {
        const int q = wid + i * C::NW;
        const int co = C::RPI * q + lane / C::LPR;
        const int c = (lane % C::LPR) ^ ((co / C::ZKD) & (C::LPR - 1));
        zvoff[i] = (unsigned)((co * C::H * A.T + c * 4) * 4);
    }

    // per-lane LDS read bases.  A (dZ): channel = lane_j, 4 consecutive pixels per ds_read_b128; the k index of MFMA s
    // in pixel group g is pixel 8g + 4*lane_h + s for BOTH operands.
    const int zrowbase = ((wmw * C::MT) * 32 + lane_j) * C::CW;
    const int zswz = lane_h ^ ((lane_j / C::ZKD) & (C::LPR - 1));
    int xoff[C::NS];
#pragma unroll
    for (int s = 0; s < C::NS; ++s) {
        int q = q0 + wq * C::NS + s;
        if (q >= C::Q) q = C::Q - 1;   // padded shift: computed, never stored
        const int rr = q / C::KW, ss = q - rr * C::KW;
        xoff[s] = (wci * 32 + lane_j) * ps + (rr - r0) * pcw + (C::DILMAX ? ss * A.dil : ss) + 4 * lane_h;
    }

    if (t_begin < t_end)
        wg_load_tile<C>(A, br, cob, cit, r0, t_begin, zbuf(0), xbuf(0), wid, lane, xvoff, xrc, zvoff, pcw);
    int cur = 0;
    // One workgroup per CU (8 or 12 waves, 2-3 per SIMD): if every wave issued its share of the next tile's DMA pieces
    // right after the barrier, the matrix pipes would idle for that long in every tile.  The waves of a SIMD take turns
    // instead: group `wid / 4` issues its pieces at wg_stagger_at(group) of the tile's CW / 8 pixel groups, while its SIMD
    // partners multiply.  (Two 4-wave workgroups per CU: the partner workgroup plays that role.)
    constexpr int NG = C::NW >= 8 ? C::NW / 4 : 1;
    constexpr int GSTEPS = C::CW / 8;
    const int grp = NG > 1 ? wid / 4 : 0;
    for (int tile = t_begin; tile < t_end; ++tile, cur ^= 1) {
        wait_vm0();
        __syncthreads();
        if constexpr (NG == 1) {
            if (tile + 1 < t_end)
                wg_load_tile<C>(A, br, cob, cit, r0, tile + 1, zbuf(cur ^ 1), xbuf(cur ^ 1), wid, lane, xvoff, xrc, zvoff, pcw);
        }
        const float* __restrict__ zp = zbuf(cur) + zrowbase;
        // B operand: one LDS byte address per shift and tile (opaque, so that it IS the base register of the reads); the
        // k position is then the instruction's immediate offset.  Left to itself the compiler bases every read on one
        // per-tile register plus a constant beyond the immediate range: a v_add per read, on the pipe the MFMAs need.
        unsigned xa[C::NS];
#pragma unroll
        for (int s = 0; s < C::NS; ++s) {
            xa[s] = lds_addr(xbuf(cur)) + xoff[s] * 4;
            asm volatile("" : "+v"(xa[s]));
        }
#pragma unroll
        for (int g = 0; g < GSTEPS; ++g) {
            if constexpr (NG > 1) {
                bool mine = false;
#pragma unroll
                for (int q = 0; q < NG; ++q) mine = mine || (g == wg_stagger_at(q, GSTEPS) && grp == q);
                if (mine && tile + 1 < t_end)
                    wg_load_tile<C>(A, br, cob, cit, r0, tile + 1, zbuf(cur ^ 1), xbuf(cur ^ 1), wid, lane, xvoff, xrc, zvoff, pcw);
            }
            f32x4 a4[C::MT];
            const int chunk = (2 * g) ^ zswz;
#pragma unroll
            for (int m = 0; m < C::MT; ++m) a4[m] = *reinterpret_cast<const f32x4*>(zp + m * 32 * C::CW + chunk * 4);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                float b[C::NS];
#pragma unroll
                for (int s = 0; s < C::NS; ++s) b[s] = lds_ld(xa[s] + (8 * g + s4) * 4);
#pragma unroll
                for (int m = 0; m < C::MT; ++m)
#pragma unroll
                    for (int s = 0; s < C::NS; ++s)
                        acc[m][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[m][s4], b[s], acc[m][s], 0, 0, 0);
            }
        }
    }

    // partial slab: [br][ks][q][co][ci]
    float* sl = A.slab + ((size_t)br * A.KS + ks) * C::Q * C::COUT * C::CIN;
    if constexpr (C::LDS_BYTES >= C::NW * 32 * ETS * 4 && C::CIN % 4 == 0) {
        if ((reinterpret_cast<uintptr_t>(sl) & 15) == 0) {
            __syncthreads();                               // every wave is done with the last K-tile's operands
            float* tb = smem + wid * (32 * ETS);
#pragma unroll
            for (int s = 0; s < C::NS; ++s) {
                const int q = q0 + wq * C::NS + s;
                if (q >= C::Q) continue;
#pragma unroll
                for (int m = 0; m < C::MT; ++m) {
                    const int co0 = cob * C::COB + (wmw * C::MT + m) * 32;
                    store_tile_t(tb, acc[m][s], sl + ((size_t)q * C::COUT + co0) * C::CIN, (size_t)C::CIN,
                                 (cit * C::WCI + wci) * 32, C::CIN, lane);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int s = 0; s < C::NS; ++s) {
        const int q = q0 + wq * C::NS + s;
        if (q >= C::Q) continue;
#pragma unroll
        for (int m = 0; m < C::MT; ++m) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int co = cob * C::COB + (wmw * C::MT + m) * 32 + acc_row(k, lane_h);
                const int ci = (cit * C::WCI + wci) * 32 + lane_j;
                sl[((size_t)q * C::COUT + co) * C::CIN + ci] = acc[m][s][k];
            }
        }
    }
}

// dw[br][co][ci][r][s] (+)= sum_ks slab[br][ks][q=r*KW+s][co][ci], fixed summation order
static __global__ void wgrad_reduce_kernel(const float* __restrict__ slab, MPtr2 dw, int KS, int Q, int cout, int cin,
                                    int accumulate, int nbr) {
    const long long per = (long long)Q * cout * cin;
    const long long total = per * nbr;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total;
         o += (long long)gridDim.x * blockDim.x) {
        const int br = (int)(o / per);
        const long long e = o % per;
        const int ci = (int)(e % cin);
        const int co = (int)((e / cin) % cout);
        const int q = (int)(e / ((long long)cin * cout));
        const float* sp = slab + (size_t)br * KS * per + e;
        float s = 0.f;
        for (int k = 0; k < KS; ++k) s += sp[(size_t)k * per];
        float* d = dw.p[br] + ((size_t)co * cin + ci) * Q + q;
        *d = accumulate ? *d + s : s;
    }
}


static int device_cus() {
    static int cus[ZNS_MAX_DEVICES] = {};
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 256;   // no device visible (workspace queries on a CPU box)
    int& c = cus[dev % ZNS_MAX_DEVICES];
    if (c == 0) c = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    return c;
}

// Split-K factor for backward-weights.  One workgroup per CU is resident (LDS), every workgroup does the same
// work, so the launch runs in ceil(WGs / CUs) rounds: pick KS that wastes the least of the last round and of the
// last (ragged) split, among launches of >= 2 rounds' worth of workgroups, preferring the smaller slab on ties.
template <class C>
static int wg_pick_ks(const ConvArgs& a, int* tiles_total, int* tps) {
    const int ncb = ceil_div(a.T, C::CW);
    const int total = a.B * C::H * ncb;
    const int base = C::NQG * (C::CIN / C::CIB) * (C::COUT / C::COB) * a.nbr;
    const int cus = device_cus() * C::WPE;    // resident workgroup slots
    int best_ks = 1;
    double best_eff = -1.0;
    const int ks_max = total < 256 ? total : 256;
    for (int ks = 1; ks <= ks_max; ++ks) {
        const int per = ceil_div(total, ks);
        if (ceil_div(total, per) != ks) continue;              // not a distinct split
        const long long wgs = (long long)base * ks;
        const long long rounds = (wgs + cus - 1) / cus;
        double eff = (double)wgs / (double)(rounds * cus) * ((double)total / ((double)ks * per));
        // a workgroup's first K-tile load, its partial-slab store and the reducer's extra read cost about two K-tiles of time:
        // without this term a short K axis (the batch-of-one fine-tune step on 192 slots: 54 tiles) was split 54 ways for a
        // last 2 % of grid fill -- one tile per workgroup, a 2.4 GB slab, 0.7 ms of reduce
        eff *= (double)per / ((double)per + 2.0);
        if (wgs < cus) eff *= 0.5;                               // do not leave CUs without a workgroup
        if (eff > best_eff + 0.02 || (eff > best_eff - 1e-9 && wgs >= 2LL * cus && (long long)base * best_ks < 2LL * cus)) {
            best_eff = eff;
            best_ks = ks;
        }
        if (best_eff > 0.985 && (long long)base * best_ks >= 2LL * cus) break;
    }
    const int per = ceil_div(total, best_ks);
    *tiles_total = total;
    *tps = per;
    return ceil_div(total, per);
}


}  // namespace zns
