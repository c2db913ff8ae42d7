// fp32 vector FMA issue rate on gfx950: v_fmac_f32 (one FMA per lane) vs v_pk_fma_f32 (two), 1..8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O2 tools/ubench_valu.hip -o tools/ubench_valu && ./tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>

template <int PK>
__global__ void k(float* out, int iters, float w) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    float x0 = 0.5f, x1 = 0.25f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, xx = {x0, x1}, ww = {w, w};
    for (int i = 0; i < iters; ++i) {
        if (PK == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                             "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(w), "v"(x0));
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                             "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ww), "v"(xx));
        }
    }
    if (PK == 0) out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    else out[blockIdx.x * blockDim.x + threadIdx.x] = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 2048 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int pk = 0; pk < 2; ++pk)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int threads = 64 * 4 * wps;
            auto launch = [&]() {
                if (pk) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0001f);
                else hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0001f);
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)iters * 64 * wps;          // wave-instructions per SIMD
            printf("%s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz), %.1f TFLOP/s\n",
                   pk ? "v_pk_fma_f32" : "v_fmac_f32  ", wps, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4,
                   instr_per_simd * 1024 * 64 * (pk ? 4 : 2) / (ms * 1e-3) / 1e12);
        }
    return 0;
}
