// Issue rates of the three float64-path instructions the bit-exact decimation chain is made of
//   acc = (double)(float)(acc + w * x)   ->  v_fma_f64, v_cvt_f32_f64, v_cvt_f64_f32
// measured as dependent chains (NCH independent chains per lane, as decimate2_kernel has) with no memory traffic:
// the ceiling of generate_XQT's decimation pyramid in tap-operations per second.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o tools/ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int NCH>
__global__ void __launch_bounds__(256) chain(double* out, int iters, double w0, double x0) {
    double acc[NCH], w = w0 + threadIdx.x * 1e-9, x = x0;
    float accf[NCH];
    for (int k = 0; k < NCH; ++k) { acc[k] = k * 1e-3; accf[k] = k * 1e-3f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                if (MODE == 0) acc[k] = __builtin_fma(w, x, acc[k]);                       // v_fma_f64 only
                if (MODE == 1) acc[k] = (double)(float)__builtin_fma(w, x, acc[k]);        // the decimation tap
                if (MODE == 2) acc[k] = (double)(float)acc[k] * 1.0000001;                 // cvt pair (+ a mul to keep it live)
                if (MODE == 3) accf[k] = __builtin_fmaf((float)w, (float)x, accf[k]);      // v_fma_f32 for scale
                if (MODE == 4) {                                                            // round-to-f32 by integer ops on the f64 bits
                    double v = __builtin_fma(w, x, acc[k]);
                    unsigned long long b = __builtin_bit_cast(unsigned long long, v);
                    b += 0x0FFFFFFFull + ((b >> 29) & 1ull);
                    b &= ~0x1FFFFFFFull;
                    acc[k] = __builtin_bit_cast(double, b);
                }
            }
        }
    }
    double s = 0;
    for (int k = 0; k < NCH; ++k) s += acc[k] + accf[k];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NCH>
static void run(const char* what, int wg_per_cu, int iters) {
    double* out;
    const int blocks = 256 * wg_per_cu;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((chain<MODE, NCH>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.999, 1.0001);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain<MODE, NCH>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.999, 1.0001);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters * 8 * NCH;      // lane-operations (one "tap" each)
    // cycles per wave-instruction-group on one SIMD at 2.4 GHz: ops / 64 lanes / (1024 SIMDs) ...
    const double per_simd_wave_ops = ops / 64.0 / 1024.0;
    printf("%-44s chains %d, %d WG/CU: %8.3f ms  %7.2f Gtap/s/chip  %.2f cycles@2.4GHz per wave-op per SIMD\n", what, NCH,
           wg_per_cu, ms, ops / ms / 1e6, ms * 1e-3 * 2.4e9 / per_simd_wave_ops);
    hipFree(out);
}

int main() {
    for (int wg : {2, 4, 8}) {
        run<0, 4>("v_fma_f64", wg, 4000);
        run<1, 4>("fma_f64 + cvt_f32_f64 + cvt_f64_f32 (tap)", wg, 4000);
        run<2, 4>("cvt_f32_f64 + cvt_f64_f32 + mul_f64", wg, 4000);
        run<3, 4>("v_fma_f32", wg, 4000);
        run<4, 4>("fma_f64 + integer round-to-24-bit", wg, 4000);
    }
    run<1, 8>("tap, 8 chains", 4, 2000);
    run<1, 2>("tap, 2 chains", 8, 8000);
    run<1, 1>("tap, 1 chain", 8, 16000);
    return 0;
}
