// Ablation of the conv_core stage structure: MFMA rate with (a) LDS operand reads, (b) one barrier per stage,
// (c) the stage's LDS-DMA traffic (weights slab + patch, double buffered), on cv4's spectral shape
// (12 waves, 4 accumulator tiles, 36 x 4 MFMAs per stage).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

constexpr int SLAB = 9216, PATCH = 4096;   // floats, as FF4

template <int MODE>   // 0: LDS reads only, 1: + barrier per stage, 2: + DMA per stage, 3: 2 + the HET wave mix (8 x 4x1, 4 x 1x3 tiles)
__global__ void __launch_bounds__(768) k(const float* __restrict__ g, float* out, int stages) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * (SLAB + PATCH); i += 768) smem[i] = 0.001f * (i & 15);
    __syncthreads();
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    const float* gb = g + (size_t)blockIdx.x * 4096;
    for (int st = 0; st < stages; ++st) {
        const float* slab = smem + (st & 1) * SLAB;
        const float* patch = smem + 2 * SLAB + (st & 1) * PATCH;
        if (MODE >= 1) {
            if (MODE >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (MODE >= 2) {
            float* ns = smem + ((st + 1) & 1) * SLAB;
            for (int q = wid; q < SLAB / 256; q += 12)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(gb + ((q * 256 + lane * 4) & 4095)), (lds_void_t*)(ns + q * 256), 16, 0, 0);
            float* np = smem + 2 * SLAB + ((st + 1) & 1) * PATCH;
            for (int q = wid; q < PATCH / 256; q += 12)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(gb + ((q * 256 + lane * 4) & 4095)), (lds_void_t*)(np + q * 256), 16, 0, 0);
        }
        if (MODE == 3 && wid >= 8) {
#pragma unroll
            for (int cp = 0; cp < 4; ++cp)
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    const float a = slab[(cp * 9 + s) * 256 + lane * 4 + (wid - 8)];
                    float b[3];
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) b[ct] = patch[(2 * cp + (lane >> 5)) * 456 + (8 + ct) * 32 + (lane & 31) + s * 11];
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[ct], acc[ct], 0, 0, 0);
                }
            continue;
        }
#pragma unroll
        for (int cp = 0; cp < 4; ++cp)
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(slab + (cp * 9 + s) * 256 + lane * 4);
                const float b = patch[(2 * cp + (lane >> 5)) * 456 + wid * 32 + (lane & 31) + s * 11];
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b, acc[m], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) s += acc[k][i];
    out[blockIdx.x * 768 + threadIdx.x] = s;
}

template <int MODE>
static void run(const float* g, float* out, int stages, int blocks) {
    const int lds = 2 * (SLAB + PATCH) * 4;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(768), lds, 0, g, out, stages);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(768), lds, 0, g, out, stages);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (MODE == 3 ? 11 : 12) * stages * 36 * 4 * 4096;
    printf("mode %d, %d WGs x %d stages: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)  %s\n", MODE, blocks, stages, ms, flop / ms / 1e9,
           flop / ms / 1e9 / 157.3 * 100, hipGetErrorString(hipGetLastError()));
}

int main() {
    float *g, *out;
    hipMalloc(&g, (size_t)3072 * 4096 * 4 + 65536); hipMalloc(&out, (size_t)3072 * 768 * 4);
    hipMemset(g, 0, (size_t)3072 * 4096 * 4 + 65536);
    run<0>(g, out, 16 * 12, 256);      // persistent: one WG per CU doing 12 WGs' worth of stages
    run<1>(g, out, 16 * 12, 256);
    run<2>(g, out, 16 * 12, 256);
    run<2>(g, out, 16, 3072);          // the real grid: 3072 WGs of 16 stages (12 rounds)
    run<3>(g, out, 16, 3072);          // ... with the 12-wave heterogeneous layout (44 tiles)
    run<3>(g, out, 16 * 12, 256);
    run<0>(g, out, 16, 3072);
    return 0;
}
