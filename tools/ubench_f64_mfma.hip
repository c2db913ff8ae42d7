// Do the float64 vector pipe (the decimation tap: v_fma_f64 + v_cvt_f32_f64 + v_cvt_f64_f32) and the fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32) of ONE SIMD work at the same time?  768-thread workgroups, one per CU: waves 0..7 (two per
// SIMD) run tap chains, waves 8..11 (one per SIMD) run a dependent MFMA chain; each role alone, then both.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_f64_mfma.hip -o tools/ubench_f64_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int F32>
__global__ void __launch_bounds__(768) both(double* out, int taps, int mfmas, int prio, double w0, double x0) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double s = 0;
    if (wave < 8) {
        if (F32) {      // control: a float32 FMA chain (three per "triple" so the instruction count matches)
            float acc[4], w = (float)w0 + threadIdx.x * 1e-6f, x = (float)x0;
            for (int k = 0; k < 4; ++k) acc[k] = k * 1e-3f;
            for (int it = 0; it < taps; ++it)
#pragma unroll
                for (int u = 0; u < 24; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = __builtin_fmaf(w, x, acc[k]);
            for (int k = 0; k < 4; ++k) s += acc[k];
        } else {
            double acc[4], w = w0 + threadIdx.x * 1e-9, x = x0;
            for (int k = 0; k < 4; ++k) acc[k] = k * 1e-3;
            for (int it = 0; it < taps; ++it)
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = (double)(float)__builtin_fma(w, x, acc[k]);
            for (int k = 0; k < 4; ++k) s += acc[k];
        }
    } else {
        if (prio) __builtin_amdgcn_s_setprio(3);
        f32x16 acc;
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        float a = (float)w0 + threadIdx.x * 1e-6f, b = (float)x0;
        for (int it = 0; it < mfmas; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        for (int k = 0; k < 16; ++k) s += acc[k];
    }
    out[(size_t)blockIdx.x * 768 + threadIdx.x] = s;
}

template <int F32>
static float run(int taps, int mfmas, int prio) {
    double* out;
    hipMalloc(&out, (size_t)256 * 768 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(both<F32>, dim3(256), dim3(768), 0, 0, out, taps, mfmas, prio, 0.999, 1.0001);
    hipEventRecord(e0);
    hipLaunchKernelGGL(both<F32>, dim3(256), dim3(768), 0, 0, out, taps, mfmas, prio, 0.999, 1.0001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}

template <int F32>
static void sweep(const char* what) {
    const int T = 20000;
    float t_tap = run<F32>(T, 0, 0);
    printf("%s chains alone (2 waves/SIMD, %d x 32 triples each): %.3f ms = %.2f cycles@2.4GHz per triple per SIMD\n", what, T, t_tap,
           t_tap * 1e-3 * 2.4e9 / (2.0 * T * 32));
    for (int m : {5000, 20000}) {
        float t_m = run<F32>(0, m, 0);
        printf("  MFMA chain alone (%d x 8): %.3f ms = %.1f cycles@2.4GHz per MFMA\n", m, t_m, t_m * 1e-3 * 2.4e9 / (8.0 * m));
        for (int prio : {0, 1}) {
            float t_b = run<F32>(T, m, prio);
            printf("    both, MFMA wave priority %d: %.3f ms   (sum %.3f, max %.3f)\n", prio * 3, t_b, t_tap + t_m, t_tap > t_m ? t_tap : t_m);
        }
    }
}

int main() {
    sweep<0>("float64 tap (fma_f64 + cvt + cvt)");
    sweep<1>("float32 fma (control)");
    return 0;
}
