// generate_XQT (vqt) in ONE launch: xqt_stream_kernel.  input_rep.py:41-54 -> librosa.vqt, see zns_xqt.hip for the arithmetic.
//
// The per-stage launches move every pyramid level to HBM and back: 13-14 GB for 4096 ten-second clips whose algorithmic
// traffic is 3.6 GB (2.6 in, 1.0 out).  Here a persistent workgroup (1024 threads: xqt_stream16_kernel, the default since
// round 5 -- eight pyramid waves + eight filter waves, see there; 768 threads: xqt_stream_kernel, the round-4 layout with four
// filter waves described below, kept as the second implementation the tests compare) STREAMS through its clips in time order,
// 4096 level-0 samples (16 frames) per step, ONE barrier per step, and no pyramid sample ever leaves LDS:
//   * waves 0..7, the pyramid: thread = (level j, 8 consecutive outputs).  EVERY level advances in EVERY step, level j one
//     step behind level j - 1 (a systolic pipeline, so nothing is recomputed and all eight waves carry the same load):
//     2048 + 1024 + ... + 32 outputs = 508 threads.  Same taps, same order, one v_fma_f32 per tap as
//     decimate_chain_f32_kernel: the same bits.  The same 512 threads bring in the next 4096 input samples (global ->
//     registers at the top of a step, registers -> LDS at its end).
//   * Level j's samples live in `window j`: double buffered, written in one step, read in the next by BOTH consumers --
//     level j + 1's decimator and octave j's filter bank.  Layout: 4-sample chunk c in plane (c & 3) at column (c >> 2), so
//     the 20 chunks a decimator thread needs (16 g .. 16 g + 79) are 5 ds_read_b128 per plane with consecutive lanes on
//     consecutive columns.  A window holds H_j samples of history before the step's own s_j = 4096 >> j (slid over from the
//     other buffer): what the next level's 63 taps reach back plus what the filter bank's 16-frame tile needs.
//   * waves 8..11, the filter bank on the fp32 matrix pipe: v_mfma_f32_16x16x4_f32 (bit for bit the fmaf chain over its
//     four taps, tools/ubench_mfma16_order.hip, so the results equal xqt_filter_kernel's), 16 frames x 16 rows per
//     instruction, the 24 real rows of an octave = two row tiles.  Wave w owns octave w (128 taps) plus a share of the short
//     octaves (2688 / 2560 / 2560 / 2688 matrix-pipe cycles per step); its A operands stay in registers for the whole kernel.  A tile of octave i = frames 16 u - FO_i .. + 15 of
//     step u (FO_i: the few frames octave i runs behind so that its last window is complete), staged by the wave itself
//     from window i into a private image with two pad words per hop (frame stride = 2 x odd: conflict-free B reads),
//     element-wise with librosa's reflect padding for the tiles that touch a signal end.  No workgroup barrier inside a step.
// A workgroup takes items b, b + G, ... back to back as ONE stream (the pipeline drains once per launch).  An item is a
// whole clip, or -- small batches, so that the launch still covers the chip -- a run of `chs` steps of a clip behind two
// warm-up steps of real samples (every output depends on a bounded span of input: 4961 samples; results are bit-identical
// however a clip is cut).  A whole-clip item needs no warm-up: the two steps that follow a clip's last frame produce
// zeros at every level, which is exactly the history resampy gives the next clip's first outputs.
#include <algorithm>
#include <cstdlib>

#include "zns_common.h"
#include "zns_xqt.h"

namespace zns {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace sq {

#define SQ_STAMP(P, s, k)                                                                                   \
    do {                                                                                                    \
        if ((P).stamps && blockIdx.x == 0 && (s) >= 64 && (s) < 96 && (threadIdx.x & 63) == 0)             \
            (P).stamps[(((s) - 64) * 12 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime();   \
    } while (0)

constexpr int S0 = 4096;                  // level-0 samples per step = 16 frames
constexpr int NPYR = 512, NTHR = 768;     // 8 pyramid waves + 4 filter-bank waves
constexpr int LAG = XQ_OCT + 1;           // steps behind the loader at which the last role (octave 7's tile) runs, plus one

__host__ __device__ constexpr int s_(int j) { return S0 >> j; }
// level j runs D_j outputs behind (the 31 samples of look-ahead of its right wing must exist): 2 D_j >= D_{j-1} + 30
__host__ __device__ constexpr int D_(int j) { return j == 0 ? 0 : j == 1 ? 16 : j == 2 ? 24 : j == 3 ? 28 : 32; }
// window j at consumer step u holds level-j samples [u s_j - H_j, (u + 1) s_j - D_j)
__host__ __device__ constexpr int H_(int j) { return j == 0 ? 64 : j == 1 ? 80 : j == 2 ? 136 : j == 3 ? 144 : j == 7 ? 48 : 96; }
__host__ __device__ constexpr int HS_(int j) { return H_(j) - D_(j); }           // samples slid over from the previous window
__host__ __device__ constexpr int WL_(int j) { return H_(j) + s_(j) - D_(j); }   // samples per window
// floats per plane (>= 4 x columns; 16 (mod 64) for the three big windows: a wave copying consecutive chunks -- planes
// 0,1,2,3,0,.. -- then hits 64 different banks per 16 lanes)
__host__ __device__ constexpr int PL_(int j) {
    return j == 0 ? 1104 : j == 1 ? 592 : j == 2 ? 336 : j == 3 ? 164 : j == 4 ? 84 : j == 5 ? 52 : j == 6 ? 36 : 16;
}
// float offset of window j: two buffers x four planes each (a closed form: a recursive constexpr is CALLED at run time)
__host__ __device__ constexpr int WOFF_(int j) {
    return 8 * ((j > 0 ? PL_(0) : 0) + (j > 1 ? PL_(1) : 0) + (j > 2 ? PL_(2) : 0) + (j > 3 ? PL_(3) : 0) + (j > 4 ? PL_(4) : 0) +
                (j > 5 ? PL_(5) : 0) + (j > 6 ? PL_(6) : 0) + (j > 7 ? PL_(7) : 0));
}
constexpr int WIN_FLOATS = WOFF_(8);
// level j + 1's thread g reads chunks 4 (DCOL_j + g) .. + 19 of window j
__host__ __device__ constexpr int DCOL_(int j) { return (H_(j) - (2 * D_(j + 1) + 32)) / 16; }
// the VQT bank (checked against the host-built bank at launch)
__host__ __device__ constexpr int NFFT_(int i) { return i < 4 ? 128 : i == 4 ? 64 : i == 7 ? 16 : 32; }
__host__ __device__ constexpr int HOP_(int i) { return 256 >> i; }
__host__ __device__ constexpr int LGH_(int i) { return 8 - i; }
// octave i's tile of step u = frames 16 u - FO_i .. + 15: the last one's window must end inside window i
__host__ __device__ constexpr int FO_(int i) { return i < 2 ? 0 : i == 2 ? 1 : i == 3 ? 2 : i == 4 ? 3 : i == 5 ? 5 : i == 6 ? 11 : 19; }
__host__ __device__ constexpr int M0_(int i) { return H_(i) - FO_(i) * HOP_(i) - NFFT_(i) / 2; }   // window-local start of the tile's first frame
// staged tile image: octaves 0 / 1 one 130-float slot per frame (128 taps), the others the contiguous segment of
// 15 hop + n_fft samples with two pad words per hop (none for hop 2: its stride is 2 x odd already)
__host__ __device__ constexpr int SS_(int i) { return i < 2 ? 130 : i == 7 ? 2 : HOP_(i) + 2; }
__host__ __device__ constexpr int NEL_(int i) { return i == 0 ? 16 * 128 : 15 * HOP_(i) + NFFT_(i); }
__host__ __device__ constexpr int STG_(int i) { return i < 2 ? 16 * 130 : ((16 * SS_(i) + NFFT_(i) + 8 + 7) / 8) * 8; }
// wave w's staging region: [octave w][its short octaves]
constexpr int STG_W0 = STG_(0) + STG_(7) + STG_(4), STG_W1 = STG_(1) + STG_(6), STG_W2 = STG_(2) + STG_(5), STG_W3 = STG_(3) + STG_(4) + STG_(7);
constexpr int STG_FLOATS = STG_W0 + STG_W1 + STG_W2 + STG_W3;
constexpr int LDS_FLOATS = WIN_FLOATS + STG_FLOATS;

template <int J>
constexpr bool window_ok() {
    // the decimator of level J + 1 stays inside window J, on whole columns, and finds its look-ahead
    constexpr bool dec = J == 7 || ((H_(J) - (2 * D_(J + 1) + 32)) % 16 == 0 && H_(J) >= 2 * D_(J + 1) + 32 &&
                                    16 * DCOL_(J) + s_(J) + 64 <= WL_(J) + 1 && 2 * D_(J + 1) >= D_(J) + 30 &&
                                    4 * (DCOL_(J) + s_(J) / 16 + 4) <= PL_(J));
    // the tile of octave J lies inside window J; one sample before it exists (right-edge reflection, J >= 1)
    constexpr bool fb = M0_(J) >= (J == 0 ? 0 : 1) && M0_(J) + 15 * HOP_(J) + NFFT_(J) <= WL_(J) && (J == 7 || M0_(J) % 4 == 0);
    return dec && fb && HS_(J) % 4 == 0 && HS_(J) <= s_(J) + (J == 7 ? 0 : 64) && 4 * ((WL_(J) + 3) / 4 + 3) / 4 <= PL_(J) && PL_(J) % 4 == 0;
}
static_assert(window_ok<0>() && window_ok<1>() && window_ok<2>() && window_ok<3>() && window_ok<4>() && window_ok<5>() &&
              window_ok<6>() && window_ok<7>(), "window geometry");
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS");


__device__ __forceinline__ int waddr(int woff, int pl, int buf, int c) { return woff + (4 * buf + (c & 3)) * pl + (c >> 2) * 4; }

// 63 taps x 8 outputs from 20 chunks of a window: p0 = plane 0 at the thread's first column, pl = floats per plane (the
// packed tap chain itself: taps63_pairs, zns_xqt.h).
__device__ __forceinline__ void taps63x8(const float* __restrict__ p0, int pl, const TapPairs& w, float (&out)[8]) {
    const float* __restrict__ p1 = p0 + pl;
    const float* __restrict__ p2 = p1 + pl;
    const float* __restrict__ p3 = p2 + pl;
    f32x2 xp[40];                // xp[q] = samples (2 q, 2 q + 1) of the thread's 80-sample span; column h = samples 16 h .. 16 h + 15
    auto column = [&](int h) {
        float4 a = *reinterpret_cast<const float4*>(p0 + 4 * h);
        const float4 b = *reinterpret_cast<const float4*>(p1 + 4 * h);
        const float4 c = *reinterpret_cast<const float4*>(p2 + 4 * h);
        float4 d = *reinterpret_cast<const float4*>(p3 + 4 * h);
        if (h == 0) asm volatile("" : "+v"(a.x));     // samples 0, 78, 79 are never used: keep the 16-byte reads whole
        if (h == 4) asm volatile("" : "+v"(d.z), "+v"(d.w));
        xp[8 * h] = f32x2{a.x, a.y}; xp[8 * h + 1] = f32x2{a.z, a.w};
        xp[8 * h + 2] = f32x2{b.x, b.y}; xp[8 * h + 3] = f32x2{b.z, b.w};
        xp[8 * h + 4] = f32x2{c.x, c.y}; xp[8 * h + 5] = f32x2{c.z, c.w};
        xp[8 * h + 6] = f32x2{d.x, d.y}; xp[8 * h + 7] = f32x2{d.z, d.w};
    };
    // All eight pyramid waves ask for their windows right after the step's barrier: 160 ds_read_b128 queue up, and a wave
    // that waited for all 20 of its own would start ~600 cycles late.  The columns are requested in the order of first use,
    // each two phases ahead of the taps that need it (the fences keep the compiler from hoisting every read to the top).
    Taps63<4> t;
    column(2); column(1);
    __builtin_amdgcn_sched_barrier(0);
    column(0);
    __builtin_amdgcn_sched_barrier(0);
    t.open(xp, w);                       // samples 33 .. 46
    t.left<0, 14>(xp, w);                // 19 .. 44 (+ 12)
    __builtin_amdgcn_sched_barrier(0);
    column(3);
    __builtin_amdgcn_sched_barrier(0);
    t.left<14, 30>(xp, w);               // 3 .. 30 (+ 12)
    t.turn(xp, w);
    __builtin_amdgcn_sched_barrier(0);
    column(4);
    __builtin_amdgcn_sched_barrier(0);
    t.right<2, 15>(xp, w);               // 35 .. 47 (+ 12)
    t.right<15, NR>(xp, w);              // 48 .. 63 (+ 12)
    t.close(xp, w, out);
}

struct Args {
    const float* wav;            // [n_clips][n]
    float* out;                  // [n_clips][96][T]
    const float* fira16;         // Bank::fira16
    int offa16[XQ_OCT];
    TapPairs taps;
    int n, T, vec0;              // samples and frames per clip; rows of wav are 16-byte aligned
    int len[XQ_OCT];             // pyramid row lengths (len[0] = n)
    int nout[XQ_OCT];            // resampy's output count of level j >= 1: len[j - 1] / 2
    int nitems, nck, chs, W, K, G;   // items = n_clips * nck; chunks per clip; steps per chunk; warm-up steps; steps per item; workgroups
    int dbg;                     // tuning aid (ZNS_XQT_DBG): 1 skip the taps, 2 skip the multiplies, 4 skip the staging, 8 skip the stores
    unsigned long long* stamps;  // tuning aid (ZNS_XQT_STAMPS): [step 64..95][wave 0..11][8] s_memtime stamps of workgroup 0
};

// item `it` of workgroup b -> (clip, chunk)
__device__ __forceinline__ void item_of(const Args& P, int b, int it, int& clip, int& ck) {
    const int m = b + it * P.G;
    clip = m / P.nck;
    ck = m - clip * P.nck;
}

// ------------------------------------------------------------------------------------------------ pyramid role
__device__ __forceinline__ void pyramid_role(const Args& P, float* __restrict__ lds, int b, int nmine, int steps) {
    const int tid = threadIdx.x;
    // level j and group g: tid 0..255 level 1, 256..383 level 2, ... 504..507 level 7, 508..511 idle
    const int r = 511 - tid;
    const bool idle = r < 4;
    const int j = idle ? 7 : 9 - (31 - __builtin_clz(r));
    const int g = tid - 512 + (512 >> (j - 1));
    int sj = 0, nout = 0, obase = 0;
    // every LDS address of the step loop is (a per-thread constant) + (the buffer's parity) x (4 planes of that window)
    int src_a = 0, src_pl = 0;                            // window j - 1: plane 0 at this thread's first column
    int dst_a0 = 0, dst_a1 = 0, dst_4pl = 0;              // window j: the two chunks of this thread's outputs
#pragma unroll
    for (int l = 1; l < XQ_OCT; ++l)
        if (j == l) {
            sj = s_(l); nout = P.nout[l]; obase = 8 * g - D_(l);
            src_pl = PL_(l - 1); src_a = WOFF_(l - 1) + 4 * (DCOL_(l - 1) + g);
            dst_4pl = 4 * PL_(l);
            dst_a0 = waddr(WOFF_(l), PL_(l), 0, HS_(l) / 4 + 2 * g);
            dst_a1 = waddr(WOFF_(l), PL_(l), 0, HS_(l) / 4 + 2 * g + 1);
        }
    // slide role: chunk ci of window jj's history
    int sl_src = 0, sl_dst = 0, sl_4pl = 0;
    bool slider = false;
    {
        int base = 0;
#pragma unroll
        for (int l = 0; l < XQ_OCT; ++l) {
            const int nc = HS_(l) / 4;
            if (tid >= base && tid < base + nc) {
                slider = true; sl_4pl = 4 * PL_(l);
                sl_dst = waddr(WOFF_(l), PL_(l), 0, tid - base);
                sl_src = waddr(WOFF_(l), PL_(l), 0, s_(l) / 4 + tid - base);
            }
            base += nc;
        }
    }
    const int ld_a0 = waddr(WOFF_(0), PL_(0), 0, HS_(0) / 4 + 2 * tid), ld_a1 = waddr(WOFF_(0), PL_(0), 0, HS_(0) / 4 + 2 * tid + 1);
    // stream positions: loader (lag 0) and own level (lag j); the item's first sample / output index, per level
    int u0 = 0, it0 = 0, clip0, ck0;
    int uj = -j, itj = 0, clipj, ckj;
    item_of(P, b, 0, clip0, ck0);
    clipj = clip0; ckj = ck0;
    // Per item, not per step: the clip's row, this thread's sample index at step 0 of the item, and the range of steps in
    // which its eight samples lie wholly inside the clip (two 16-byte loads, nothing to check); the level's first output
    // index at step 0 likewise.  The step loop then adds u * S0 (a scalar) and compares u with two registers.
    const float* __restrict__ xrow = P.wav;
    int ld_i0 = 0, ld_ulo = 1, ld_uhi = 0, oj0 = 0;
    auto loader_item = [&]() {
        xrow = P.wav + (long long)clip0 * P.n;
        ld_i0 = (ck0 * P.chs - P.W) * S0 + 8 * tid;
        // idx(u) = ld_i0 + u * S0 >= 0  and  idx(u) + 7 < n
        ld_ulo = ld_i0 >= 0 ? 0 : (-ld_i0 + S0 - 1) / S0;
        ld_uhi = P.vec0 && it0 < nmine ? (P.n - 8 - ld_i0 >= 0 ? (P.n - 8 - ld_i0) / S0 : -1) : -1;
    };
    auto level_item = [&]() { oj0 = (((ckj * P.chs - P.W) * S0) >> j) + obase; };
    loader_item();
    level_item();
    const TapPairs& W = P.taps;
    for (int s = 0; s < steps; ++s) {
        const int par = s & 1;
        SQ_STAMP(P, s, 0);
        // ---- level 0: the step's 8 samples of this thread, global -> registers (stored at the end of the step)
        float4 fa = make_float4(0.f, 0.f, 0.f, 0.f), fb = fa;
        if (u0 >= ld_ulo && u0 <= ld_uhi) {
            const float* __restrict__ xp = xrow + u0 * S0;               // a scalar base; the lane part is ld_i0
            fa = *reinterpret_cast<const float4*>(xp + ld_i0);
            fb = *reinterpret_cast<const float4*>(xp + ld_i0 + 4);
        } else if (it0 < nmine) {
            const int idx = ld_i0 + u0 * S0;                             // a clip end, or rows off 16-byte alignment
            if (idx + 7 >= 0 && idx < P.n) {
                const float* __restrict__ x = xrow;
                if (idx >= 0) fa.x = x[idx];
                if (idx + 1 >= 0 && idx + 1 < P.n) fa.y = x[idx + 1];
                if (idx + 2 >= 0 && idx + 2 < P.n) fa.z = x[idx + 2];
                if (idx + 3 >= 0 && idx + 3 < P.n) fa.w = x[idx + 3];
                if (idx + 4 >= 0 && idx + 4 < P.n) fb.x = x[idx + 4];
                if (idx + 5 >= 0 && idx + 5 < P.n) fb.y = x[idx + 5];
                if (idx + 6 >= 0 && idx + 6 < P.n) fb.z = x[idx + 6];
                if (idx + 7 >= 0 && idx + 7 < P.n) fb.w = x[idx + 7];
            }
        }
        SQ_STAMP(P, s, 1);
        // ---- own level: eight outputs from the window the level below filled in the previous step
        if (!idle) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (uj >= 0 && itj < nmine) {
                const int o0 = oj0 + uj * sj;                             // first output index in the clip
                if (o0 + 7 >= 0 && o0 < nout && !(P.dbg & 1)) {
                    float acc[8];
                    taps63x8(lds + src_a + (par ? 4 * src_pl : 0), src_pl, W, acc);
                    if (o0 >= 0 && o0 + 7 < nout) {          // the common case: all eight inside the row
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = dc_div_sqrt_half((double)acc[k]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = (o0 + k >= 0 && o0 + k < nout) ? dc_div_sqrt_half((double)acc[k]) : 0.f;
                    }
                }
            }
            const int flip = par ? 0 : dst_4pl;            // the other buffer
            *reinterpret_cast<float4*>(lds + dst_a0 + flip) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(lds + dst_a1 + flip) = make_float4(v[4], v[5], v[6], v[7]);
        }
        SQ_STAMP(P, s, 2);
        // ---- slide: the last HS samples of a window open the next one
        if (slider)
            *reinterpret_cast<float4*>(lds + sl_dst + (par ? 0 : sl_4pl)) = *reinterpret_cast<const float4*>(lds + sl_src + (par ? sl_4pl : 0));
        // ---- level 0: registers -> window 0 of the next step
        *reinterpret_cast<float4*>(lds + ld_a0 + (par ? 0 : 4 * PL_(0))) = fa;
        *reinterpret_cast<float4*>(lds + ld_a1 + (par ? 0 : 4 * PL_(0))) = fb;
        if (++u0 == P.K) { u0 = 0; ++it0; item_of(P, b, it0, clip0, ck0); loader_item(); }
        if (++uj == P.K) { uj = 0; ++itj; item_of(P, b, itj, clipj, ckj); level_item(); }
        SQ_STAMP(P, s, 3);
        __syncthreads();
        SQ_STAMP(P, s, 4);
    }
}

// ------------------------------------------------------------------------------------------------ filter-bank role
template <int NK>
__device__ __forceinline__ void load_operand(const float* __restrict__ ga, int lane, float (&a)[NK]) {
#pragma unroll
    for (int s = 0; s < NK; ++s) a[s] = ga[s * 64 + lane];
    // opaque from here on: otherwise the compiler re-loads the operands from global memory in every step instead of keeping
    // them in registers (a read-only load is cheaper to redo than to keep, by its cost model) and every tile waits for L2
#pragma unroll
    for (int s = 0; s < NK; ++s) asm volatile("" : "+v"(a[s]));
}

__device__ __forceinline__ int reflect_idx(int m, int len) {
    m = m < 0 ? -m : m;
    m = m >= len ? 2 * (len - 1) - m : m;
    return m < 0 ? 0 : (m >= len ? len - 1 : m);      // frames past the last one are never written out
}

// A signal end inside the tile of octave I: the image element by element with librosa's center=True reflect padding, from
// window I (octave 0: from the clip itself -- its reflection reaches one sample outside the window).  Rare (two tiles per
// clip and octave): out of line, arguments by value, so that nothing of the step loop's state has to live in memory.
template <int I>
__device__ __noinline__ void stage_edge(const float* x, int len, int worig, int Ff, const float* lds, float* stg, int par, int lane) {
    constexpr int hop = HOP_(I), nfft = NFFT_(I), lg = LGH_(I);
    for (int e = lane; e < NEL_(I); e += 64) {
        int abs_i, d;
        if (I == 0) {
            abs_i = (Ff + (e >> 7)) * 256 - 64 + (e & 127);
            d = (e >> 7) * 130 + (e & 127);
        } else {
            abs_i = Ff * hop - nfft / 2 + e;
            d = I == 7 ? e : e + 2 * (e >> (I == 1 ? 7 : lg));
        }
        const int idx = reflect_idx(abs_i, len);
        float val;
        if (I == 0) {
            val = x[idx];
        } else {
            int m = idx - worig;
            m = m < 0 ? 0 : (m >= WL_(I) ? WL_(I) - 1 : m);
            val = lds[waddr(WOFF_(I), PL_(I), par, m >> 2) + (m & 3)];
        }
        stg[d] = val;
    }
}

// One (wave, octave) slot: the per-lane constants of octave I's tile (everything the step loop adds to them is a scalar or an
// immediate) and its position in the stream.
template <int I>
struct Oct {
    static constexpr int hop = HOP_(I), nfft = NFFT_(I), lg = LGH_(I);
    static constexpr int NCH = I == 0 ? 512 : NEL_(I) / 4;          // 16-byte chunks of the tile image (octave 7: element-wise)
    static constexpr int NIT = (NCH + 63) / 64;
    // chunk lane + 64 it of the image: source = window chunk (M0 / 4 + lane + 64 it) -- its plane is a lane constant, its
    // column advances 16 per it; destination = image element 4 (lane + 64 it) plus two pad words per hop: 256 it is a whole
    // number of hops.  Octave 0: 32 chunks per frame slot, two frames per it.
    static constexpr int SRC_STEP = I == 0 ? 128 : 64;
    static constexpr int DST_STEP = I == 0 ? 260 : 256 + 2 * (256 >> (I == 1 ? 7 : lg));
    int src, dst;          // float offsets: inside window I (buffer 0) / inside the staging region
    int ko0, ko1;          // output offsets (floats) of the lane's two filters of row tile 0: kbin * T + n (row tile 1: + 8 T)
    int u, it, clip, ck;   // stream position
    bool tile, edge;       // this step: a tile to compute; it touches a signal end
    int Ff, flo, fhi;

    __device__ __forceinline__ void init(const Args& P, int lane, int b) {
        if (I == 0) {
            src = waddr(WOFF_(0), PL_(0), 0, 64 * (lane >> 5) + (lane & 31));
            dst = (lane >> 5) * 130 + 4 * (lane & 31);
        } else if (I == 7) {
            const int m = M0_(7) + lane;
            src = waddr(WOFF_(7), PL_(7), 0, m >> 2) + (m & 3);
            dst = lane;
        } else {
            src = waddr(WOFF_(I), PL_(I), 0, M0_(I) / 4 + lane);
            dst = 4 * lane + 2 * ((4 * lane) >> (I == 1 ? 7 : lg));
        }
        const int n = lane & 15, kk = lane >> 4;
        ko0 = (XQ_BINS - XQ_BPO * (I + 1) + 2 * kk) * P.T + n;
        ko1 = ko0 + P.T;
        u = -(I + 1); it = 0;
        item_of(P, b, 0, clip, ck);
        tile = edge = false; Ff = flo = fhi = 0;
    }
    // phase 0 of a step: is there a tile, and where
    __device__ __forceinline__ void begin(const Args& P, int nmine) {
        tile = false;
        if (u >= 0 && it < nmine) {
            flo = ck * P.chs * 16;
            Ff = flo + 16 * (u - P.W) - FO_(I);
            fhi = flo + P.chs * 16 < P.T ? flo + P.chs * 16 : P.T;
            tile = Ff + 15 >= flo && Ff < fhi;
            const int last = Ff + 15 < P.T - 1 ? Ff + 15 : P.T - 1;
            edge = Ff * hop - nfft / 2 < 0 || last * hop + nfft / 2 > P.len[I];
        }
    }
    // clip index of window-local sample 0
    __device__ __forceinline__ int worig(const Args& P) const { return (((ck * P.chs - P.W) * S0) >> I) + u * s_(I) - H_(I); }
    __device__ __forceinline__ void advance(const Args& P, int b) {
        if (++u == P.K) { u = 0; ++it; item_of(P, b, it, clip, ck); }
    }
    // phase 1: window I (buffer par) -> registers (an interior tile); phase 2: registers -> the staged image
    __device__ __forceinline__ void stage_load(const float* __restrict__ lds, int par, int lane, float4 (&v)[NIT]) const {
        const float* __restrict__ sp = lds + src + (par ? 4 * PL_(I) : 0);
        if (I == 7) {
            v[0].x = lane < NEL_(7) ? *sp : 0.f;
        } else {
#pragma unroll
            for (int t = 0; t < NIT; ++t)
                if ((t + 1) * 64 <= NCH || lane + 64 * t < NCH) v[t] = *reinterpret_cast<const float4*>(sp + SRC_STEP * t);
        }
    }
    __device__ __forceinline__ void stage_store(float* __restrict__ stg, int lane, const float4 (&v)[NIT]) const {
        float* __restrict__ dp = stg + dst;
        if (I == 7) {
            if (lane < NEL_(7)) *dp = v[0].x;
        } else {
#pragma unroll
            for (int t = 0; t < NIT; ++t)
                if ((t + 1) * 64 <= NCH || lane + 64 * t < NCH) {
                    *reinterpret_cast<float2*>(dp + DST_STEP * t) = make_float2(v[t].x, v[t].y);
                    *reinterpret_cast<float2*>(dp + DST_STEP * t + 2) = make_float2(v[t].z, v[t].w);
                }
        }
    }
    // rows 4 kk + r of a row tile are registers r of lane (n, kk): (re, im) of filters 8 rt + 2 kk and 8 rt + 2 kk + 1
    template <int RT>
    __device__ __forceinline__ void store(const Args& P, const f32x4& acc, int lane) const {
        const int F = Ff + (lane & 15);
        if (F < flo || F >= fhi || (RT == 1 && lane >= 32)) return;
        float* __restrict__ o = P.out + ((size_t)clip * XQ_BINS * P.T + Ff) + (RT ? 8 * P.T : 0);
        if (P.dbg & 16) { o[ko0] = acc[0]; o[ko1] = acc[2]; return; }
        if (P.dbg & 32) { const float q = xqt_log_magnitude(acc[0], acc[1]) + xqt_log_magnitude(acc[2], acc[3]); if (q == 1234.5f) o[0] = q; return; }
        o[ko0] = xqt_log_magnitude(acc[0], acc[1]);
        o[ko1] = xqt_log_magnitude(acc[2], acc[3]);
    }
};

// D[16 rt + row][frame] += A[.][tap] B[tap][frame] over the NFFT taps: one v_mfma_f32_16x16x4_f32 per four taps and row
// tile.  bp = stg + frame * SS + (lane >> 4); tap 4 ks + kk of frame n sits at n SS + 4 ks + kk + 2 (4 ks >> LG).
template <int NFFT, int LG, int RTM>
__device__ __forceinline__ void tile_mma(const float* __restrict__ bp, const float (&a0)[NFFT / 4], const float (&a1)[NFFT / 4],
                                         f32x4& acc0, f32x4& acc1) {
    constexpr int NK = NFFT / 4, BB = 4;
    auto off = [](int ks) { return 4 * ks + (LG >= 2 ? 2 * ((4 * ks) >> (LG >= 2 ? LG : 0)) : 0); };
    float bv[2][BB];
#pragma unroll
    for (int u = 0; u < BB; ++u) bv[0][u] = bp[off(u)];
#pragma unroll
    for (int s0 = 0; s0 < NK; s0 += BB) {
        const int cur = (s0 / BB) & 1;
        if (s0 + BB < NK) {
#pragma unroll
            for (int u = 0; u < BB; ++u) bv[cur ^ 1][u] = bp[off(s0 + BB + u)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < BB; ++u) {
            if (RTM & 1) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s0 + u], bv[cur][u], acc0, 0, 0, 0);
            if (RTM & 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s0 + u], bv[cur][u], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int I, int RTM>
__device__ __forceinline__ void octave_mma(const Args& P, const Oct<I>& O, const float* __restrict__ stg, int lane,
                                           const float (&a0)[NFFT_(I) / 4], const float (&a1)[NFFT_(I) / 4]) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // same-wave LDS traffic is ordered: no barrier between the staging writes and the operand reads
    if (!(P.dbg & 2)) tile_mma<NFFT_(I), (I < 2 ? 7 : I == 7 ? 0 : LGH_(I)), RTM>(stg + (lane & 15) * SS_(I) + (lane >> 4), a0, a1, acc0, acc1);
    if (!(P.dbg & 8)) {
        if (RTM & 1) O.template store<0>(P, acc0, lane);
        if (RTM & 2) O.template store<1>(P, acc1, lane);
    }
}

template <int WV>
__device__ __forceinline__ void filter_role(const Args& P, float* __restrict__ lds, int b, int nmine, int steps) {
    const int lane = threadIdx.x & 63;
    float* __restrict__ stg = lds + WIN_FLOATS + (WV == 0 ? 0 : WV == 1 ? STG_W0 : WV == 2 ? STG_W0 + STG_W1 : STG_W0 + STG_W1 + STG_W2);
    // The octaves of this wave: WV (128 taps, both row tiles), S1 and -- waves 0 and 3 -- one row tile of E:
    //   wave 0: octave 0 + row tile 0 of octave 7 + row tile 1 of octave 4      2048 + 128 + 512 = 2688 matrix-pipe cycles per step
    //   wave 1: octave 1 + octave 6                                              2048 + 512       = 2560
    //   wave 2: octave 2 + octave 5                                              2048 + 512       = 2560
    //   wave 3: octave 3 + row tile 0 of octave 4 + row tile 1 of octave 7      2048 + 512 + 128 = 2688
    // (both waves that share an octave stage it for themselves: 46 or 304 floats, cheaper than a barrier).  Measured
    // against it (cycle stamps, the younger pyramid wave of each SIMD reaching the barrier): all of octave 7 and row tile 1 of
    // octave 4 on wave 0: 8850 / 8120 / 8140 / 7920; this split: 8580 / 7940 / 7890 / 8500; octave 4's row tiles on waves 0 / 3
    // and octave 7's on waves 1 / 2: 7760 / 8670 / 8770 / 7670 -- a slot costs ~1200 cycles of its wave whatever its size.
    constexpr int S1 = WV == 0 ? 7 : WV == 1 ? 6 : WV == 2 ? 5 : 4;
    constexpr int S1_RTM = (WV == 0 || WV == 3) ? 1 : 3;
    constexpr bool HAS_E = WV == 0 || WV == 3;
    constexpr int EO = WV == 0 ? 4 : 7;
    constexpr int E_RTM = 2;
    float* __restrict__ stgB = stg, * __restrict__ stgS = stg + STG_(WV), * __restrict__ stgE = stg + STG_(WV) + STG_(S1);
    float a0[32], a1[32];
    load_operand<32>(P.fira16 + P.offa16[WV], lane, a0);
    load_operand<32>(P.fira16 + P.offa16[WV] + 32 * 64, lane, a1);
    float c0[NFFT_(S1) / 4], c1[NFFT_(S1) / 4];
    if constexpr (S1_RTM & 1) load_operand<NFFT_(S1) / 4>(P.fira16 + P.offa16[S1], lane, c0);
    if constexpr (S1_RTM & 2) load_operand<NFFT_(S1) / 4>(P.fira16 + P.offa16[S1] + (NFFT_(S1) / 4) * 64, lane, c1);
    float e1[NFFT_(EO) / 4];
    if constexpr (HAS_E) load_operand<NFFT_(EO) / 4>(P.fira16 + P.offa16[EO] + (E_RTM == 2 ? (NFFT_(EO) / 4) * 64 : 0), lane, e1);
    Oct<WV> OB; Oct<S1> OS; Oct<EO> OE;
    OB.init(P, lane, b); OS.init(P, lane, b); OE.init(P, lane, b);
    // the one filter wave of a SIMD is its youngest wave and loses issue arbitration to the two pyramid waves: give it the
    // issue slots it asks for (3.25 -> 2.90 ms per 4096 clips)
    if (!(P.dbg & 64)) __builtin_amdgcn_s_setprio(3);
    for (int s = 0; s < steps; ++s) {
        const int par = s & 1;
        SQ_STAMP(P, s, 0);
        OB.begin(P, nmine); OS.begin(P, nmine);
        if constexpr (HAS_E) OE.begin(P, nmine);
        const bool stage = !(P.dbg & 4);
        // all of the step's interior tiles: window -> registers together (one LDS latency), then registers -> images
        float4 vb[Oct<WV>::NIT], vs[Oct<S1>::NIT], ve[Oct<EO>::NIT];
        if (stage && OB.tile && !OB.edge) OB.stage_load(lds, par, lane, vb);
        if (stage && OS.tile && !OS.edge) OS.stage_load(lds, par, lane, vs);
        if constexpr (HAS_E) { if (stage && OE.tile && !OE.edge) OE.stage_load(lds, par, lane, ve); }
        if (stage && OB.tile) { if (OB.edge) stage_edge<WV>(P.wav + (long long)OB.clip * P.n, P.len[WV], OB.worig(P), OB.Ff, lds, stgB, par, lane); else OB.stage_store(stgB, lane, vb); }
        if (stage && OS.tile) { if (OS.edge) stage_edge<S1>(P.wav + (long long)OS.clip * P.n, P.len[S1], OS.worig(P), OS.Ff, lds, stgS, par, lane); else OS.stage_store(stgS, lane, vs); }
        if constexpr (HAS_E) { if (stage && OE.tile) { if (OE.edge) stage_edge<EO>(P.wav + (long long)OE.clip * P.n, P.len[EO], OE.worig(P), OE.Ff, lds, stgE, par, lane); else OE.stage_store(stgE, lane, ve); } }
        SQ_STAMP(P, s, 1);
        if (OB.tile) octave_mma<WV, 3>(P, OB, stgB, lane, a0, a1);
        SQ_STAMP(P, s, 2);
        if (OS.tile) octave_mma<S1, S1_RTM>(P, OS, stgS, lane, c0, c1);
        if constexpr (HAS_E) { if (OE.tile) octave_mma<EO, E_RTM>(P, OE, stgE, lane, e1, e1); }
        OB.advance(P, b); OS.advance(P, b);
        if constexpr (HAS_E) OE.advance(P, b);
        SQ_STAMP(P, s, 3);
        __syncthreads();
        SQ_STAMP(P, s, 4);
    }
}

// ------------------------------------------------------------------------------------------------ 16-wave variant
// TWO filter waves per SIMD (waves 8 + s and 12 + s sit on SIMD s): wave (WV, HALF) multiplies row tile HALF of octave WV
// (32 of its 64 MFMAs) and row tile HALF of one short octave.  Why: with one filter wave per SIMD its chain -- stage the tile
// image, 80 MFMAs, log epilogue, stores -- is the critical path of a step (phase-skipping bounds, DESIGN.md section 5:
// without the multiplies 2.08 of 2.88 ms per 4096 clips, without the staging 2.52); two waves run the two halves of that
// chain side by side and cover each other's LDS and store latencies.  Both stage the octave's image for themselves (a
// barrier between a staging wave and a multiplying wave costs more than the 8 KB of LDS traffic).
//   wave (0, 0): octave 0 rt 0 + octave 7 rt 0      wave (0, 1): octave 0 rt 1 + octave 4 rt 1
//   wave (1, h): octave 1 rt h + octave 6 rt h      wave (2, h): octave 2 rt h + octave 5 rt h
//   wave (3, 0): octave 3 rt 0 + octave 4 rt 0      wave (3, 1): octave 3 rt 1 + octave 7 rt 1
constexpr int NTHR2 = 1024;
__host__ __device__ constexpr int S2_(int wv, int half) { return wv == 0 ? (half ? 4 : 7) : wv == 1 ? 6 : wv == 2 ? 5 : (half ? 7 : 4); }
__host__ __device__ constexpr int STG2_SLOT(int wv, int half) { return STG_(wv) + STG_(S2_(wv, half)); }
__host__ __device__ constexpr int STG2_OFF(int wv, int half) {
    int o = 0;
    for (int w = 0; w < 4; ++w)
        for (int h = 0; h < 2; ++h) {
            if (w == wv && h == half) return o;
            o += STG2_SLOT(w, h);
        }
    return o;
}
constexpr int STG2_FLOATS = STG2_OFF(3, 1) + STG2_SLOT(3, 1);
constexpr int LDS_FLOATS2 = WIN_FLOATS + STG2_FLOATS;
static_assert(LDS_FLOATS2 * 4 <= 160 * 1024, "LDS (16-wave variant)");

template <int WV, int HALF>
__device__ __forceinline__ void filter_role2(const Args& P, float* __restrict__ lds, int b, int nmine, int steps) {
    const int lane = threadIdx.x & 63;
    constexpr int S2 = S2_(WV, HALF);
    constexpr int RTM = HALF ? 2 : 1;
    float* __restrict__ stgB = lds + WIN_FLOATS + STG2_OFF(WV, HALF);
    float* __restrict__ stgS = stgB + STG_(WV);
    float a[32], c[NFFT_(S2) / 4];
    load_operand<32>(P.fira16 + P.offa16[WV] + HALF * 32 * 64, lane, a);
    load_operand<NFFT_(S2) / 4>(P.fira16 + P.offa16[S2] + HALF * (NFFT_(S2) / 4) * 64, lane, c);
    Oct<WV> OB; Oct<S2> OS;
    OB.init(P, lane, b); OS.init(P, lane, b);
    if (!(P.dbg & 64)) __builtin_amdgcn_s_setprio(3);
    for (int s = 0; s < steps; ++s) {
        const int par = s & 1;
        OB.begin(P, nmine); OS.begin(P, nmine);
        const bool stage = !(P.dbg & 4);
        float4 vb[Oct<WV>::NIT], vs[Oct<S2>::NIT];
        if (stage && OB.tile && !OB.edge) OB.stage_load(lds, par, lane, vb);
        if (stage && OS.tile && !OS.edge) OS.stage_load(lds, par, lane, vs);
        if (stage && OB.tile) { if (OB.edge) stage_edge<WV>(P.wav + (long long)OB.clip * P.n, P.len[WV], OB.worig(P), OB.Ff, lds, stgB, par, lane); else OB.stage_store(stgB, lane, vb); }
        if (stage && OS.tile) { if (OS.edge) stage_edge<S2>(P.wav + (long long)OS.clip * P.n, P.len[S2], OS.worig(P), OS.Ff, lds, stgS, par, lane); else OS.stage_store(stgS, lane, vs); }
        if (OB.tile) octave_mma<WV, RTM>(P, OB, stgB, lane, a, a);
        if (OS.tile) octave_mma<S2, RTM>(P, OS, stgS, lane, c, c);
        OB.advance(P, b); OS.advance(P, b);
        __syncthreads();
    }
}

__global__ void __launch_bounds__(NTHR2) xqt_stream16_kernel(Args P) {
    extern __shared__ __attribute__((aligned(16))) float sq_lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int nmine = (P.nitems - b + P.G - 1) / P.G;
    const int steps = nmine * P.K + LAG;
    for (int i = tid; i < LDS_FLOATS2 / 4; i += NTHR2) reinterpret_cast<float4*>(sq_lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wv < NPYR / 64) pyramid_role(P, sq_lds, b, nmine, steps);
    else if (wv == 8) filter_role2<0, 0>(P, sq_lds, b, nmine, steps);
    else if (wv == 9) filter_role2<1, 0>(P, sq_lds, b, nmine, steps);
    else if (wv == 10) filter_role2<2, 0>(P, sq_lds, b, nmine, steps);
    else if (wv == 11) filter_role2<3, 0>(P, sq_lds, b, nmine, steps);
    else if (wv == 12) filter_role2<0, 1>(P, sq_lds, b, nmine, steps);
    else if (wv == 13) filter_role2<1, 1>(P, sq_lds, b, nmine, steps);
    else if (wv == 14) filter_role2<2, 1>(P, sq_lds, b, nmine, steps);
    else filter_role2<3, 1>(P, sq_lds, b, nmine, steps);
}

__global__ void __launch_bounds__(NTHR) xqt_stream_kernel(Args P) {
    extern __shared__ __attribute__((aligned(16))) float sq_lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int nmine = (P.nitems - b + P.G - 1) / P.G;        // items b, b + G, ... of this workgroup
    const int steps = nmine * P.K + LAG;
    for (int i = tid; i < LDS_FLOATS / 4; i += NTHR) reinterpret_cast<float4*>(sq_lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // wave-uniform role (a scalar branch: every wave runs exactly one of the step loops, same number of barriers)
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wv < NPYR / 64) pyramid_role(P, sq_lds, b, nmine, steps);
    else if (wv == 8) filter_role<0>(P, sq_lds, b, nmine, steps);
    else if (wv == 9) filter_role<1>(P, sq_lds, b, nmine, steps);
    else if (wv == 10) filter_role<2>(P, sq_lds, b, nmine, steps);
    else filter_role<3>(P, sq_lds, b, nmine, steps);
}

}  // namespace sq

static unsigned long long* g_last_stamps = nullptr;
unsigned long long* xqt_stream_stamps() { return g_last_stamps; }

bool xqt_stream_supported(const Bank* b, int n_clips, int n_samples) {
    if (n_clips < 1 || n_samples < 1) return false;
    for (int i = 0; i < XQ_OCT; ++i)
        if (b->n_fft[i] != sq::NFFT_(i) || b->hop[i] != sq::HOP_(i)) return false;     // the vqt bank only
    // index arithmetic of the step loop is 32-bit: the padded timeline of a clip must fit
    return (long long)n_samples + 4LL * sq::S0 < (1LL << 30);
}

int xqt_stream_launch(const Bank* b, const float* wav, int n_clips, int n_samples, float* out, hipStream_t st) {
    using namespace sq;
    ZNS_CHECK_ARG(xqt_stream_supported(b, n_clips, n_samples), "xqt (fused kernel): vqt bank only");
    Args A{};
    A.wav = wav;
    A.out = out;
    A.fira16 = b->fira16;
    for (int i = 0; i < XQ_OCT; ++i) A.offa16[i] = b->offa16[i];
    for (int i = 0; i < 30; ++i) A.taps.p[i] = f32x2{b->tapsf[i], b->tapsf[i + 2]};
    A.n = n_samples;
    A.T = zns_xqt_frames(n_samples);
    A.vec0 = (n_samples % 4 == 0) && ((uintptr_t)wav % 16 == 0);
    A.len[0] = n_samples;
    for (int i = 1; i < XQ_OCT; ++i) { A.len[i] = (A.len[i - 1] + 1) / 2; A.nout[i] = A.len[i - 1] / 2; }
    int ncu = 256;
    {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            ncu = v;
    }
    // items: whole clips when there are enough of them to cover the chip, else each clip in nck runs of chs steps
    const int tiles = (A.T + 15) / 16;
    int nck = 1;
    if (n_clips < ncu) nck = std::max(1, std::min(tiles / 2, ncu / n_clips));
    if (const int ck = tuning(ZNS_TUNE_XQT_CHUNKS)) nck = std::max(1, std::min(tiles, ck));     // tuning aid
    A.chs = (tiles + nck - 1) / nck;
    A.nck = (tiles + A.chs - 1) / A.chs;
    A.W = A.nck > 1 ? 2 : 0;
    A.K = A.W + A.chs + 2;
    A.nitems = n_clips * A.nck;
    A.G = std::min(A.nitems, ncu);
    A.dbg = tuning(ZNS_TUNE_XQT_DBG);      // phase-skipping bits: timing diagnosis only (include/zns_test.h)
    static unsigned long long* g_stamps[ZNS_MAX_DEVICES] = {};
    if (tuning(ZNS_TUNE_XQT_STAMPS)) {      // tuning aid: cycle stamps of workgroup 0, dumped by zns_xqt_debug_stamps
        unsigned long long*& sb = g_stamps[current_device_slot()];
        if (!sb) ZNS_HIP(hipMalloc(&sb, 32 * 12 * 8 * sizeof(unsigned long long)));
        ZNS_HIP(hipMemsetAsync(sb, 0, 32 * 12 * 8 * sizeof(unsigned long long), st));
        A.stamps = sb;
        g_last_stamps = sb;
    }
    static bool lds_set[ZNS_MAX_DEVICES] = {};
    if (!lds_set[current_device_slot()]) {
        ZNS_HIP(hipFuncSetAttribute((const void*)xqt_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ZNS_HIP(hipFuncSetAttribute((const void*)xqt_stream16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lds_set[current_device_slot()] = true;
    }
    // 16 waves (two filter waves per SIMD) unless a test asks for the 12-wave layout or the cycle stamps (12-wave only) are on:
    // same bits; per 4096 / 2048 / 1024 / 192 clips 2.77 / 1.26 / 0.68 / 0.182 ms against 2.87 / 1.41 / 0.76 / 0.193
    if (tuning(ZNS_TUNE_XQT_WAVES16) >= 0 && !A.stamps)
        hipLaunchKernelGGL(xqt_stream16_kernel, dim3((unsigned)A.G), dim3(NTHR2), (size_t)LDS_FLOATS2 * sizeof(float), st, A);
    else
        hipLaunchKernelGGL(xqt_stream_kernel, dim3((unsigned)A.G), dim3(NTHR), (size_t)LDS_FLOATS * sizeof(float), st, A);
    ZNS_LAUNCH_CHECK();
    return ZNS_OK;
}

}  // namespace zns
