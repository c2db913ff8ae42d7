// Raw issue rate of v_mfma_f32_32x32x2_f32 (no memory traffic): what fraction of the 157.3 TFLOP/s datasheet peak the
// matrix pipes sustain on this box, as the practical ceiling for the conv kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o tools/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(768) mfma_loop(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k)
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < NACC; ++k)
        for (int i = 0; i < 16; ++i) s += acc[k][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void run(int wg_per_cu, int waves, int iters) {
    float* out;
    const int blocks = 256 * wg_per_cu;
    hipMalloc(&out, (size_t)blocks * waves * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(waves * 64), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(waves * 64), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * waves * iters * NACC * 32 * 32 * 2 * 2;
    printf("acc tiles/wave %d, %d WG/CU x %d waves: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", NACC, wg_per_cu, waves, ms,
           flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
    hipFree(out);
}

int main() {
    run<4>(1, 4, 20000);     // 1 wave per SIMD
    run<4>(1, 8, 10000);     // 2 waves per SIMD
    run<4>(1, 12, 8000);     // 3 waves per SIMD
    run<2>(1, 12, 16000);
    run<6>(2, 4, 8000);
    run<4>(1, 12, 80000);    // long run: sustained clocks
    return 0;
}
