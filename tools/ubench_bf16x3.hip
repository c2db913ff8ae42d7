// What a bf16 x 3 split of the fp32 operands would buy and cost (NOT used by the library).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_bf16x3.hip -o tools/ubench_bf16x3 && ./tools/ubench_bf16x3
// A [32 x K] . B [K x 32] with K = 16384 (the depth of the path's largest reductions), standard-normal operands, against a
// float64 product on the host:
//   fp32    v_mfma_f32_32x32x2_f32 (what the library runs: an fmaf chain, 64 cycles per K = 2)
//   x3/6    a = a1 + a2 + a3 in bf16 (24 mantissa bits), the six products down to 2^-16 a1 b1, v_mfma_f32_32x32x16_bf16
//           (32 cycles per K = 16: 192 cycles per K = 16 against fp32's 512 -- 2.67x the matrix rate)
//   x2/3    a = a1 + a2, three products (1.33x .. 5.3x the rate; 16 mantissa bits)
// and the issue rate of both instructions, measured.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int acc_row(int reg, int lane_h) { return (reg & 3) + 8 * (reg >> 2) + 4 * lane_h; }

__global__ void gemm_fp32(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc = {0};
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) C[acc_row(i, h) * 32 + r] = acc[i];
}

__device__ __forceinline__ void split3(float v, __bf16& p1, __bf16& p2, __bf16& p3) {
    p1 = (__bf16)v;
    const float r1 = v - (float)p1;
    p2 = (__bf16)r1;
    p3 = (__bf16)(r1 - (float)p2);
}

// terms: 6 = all products down to 2^-16, 3 = the bf16 x 2 form
__global__ void gemm_bf16x3(const float* A, const float* B, float* C, int K, int terms) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 hi = {0}, lo = {0};           // small terms in their own accumulator, added at the end
    for (int k = 0; k < K; k += 16) {
        bf16x8 a1, a2, a3, b1, b2, b3;
        for (int j = 0; j < 8; ++j) {
            __bf16 p1, p2, p3;
            split3(A[r * K + k + 8 * h + j], p1, p2, p3);
            a1[j] = p1; a2[j] = p2; a3[j] = p3;
            split3(B[(k + 8 * h + j) * 32 + r], p1, p2, p3);
            b1[j] = p1; b2[j] = p2; b3[j] = p3;
        }
        hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, hi, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, lo, 0, 0, 0);
        if (terms == 6) {
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, lo, 0, 0, 0);
        }
    }
    for (int i = 0; i < 16; ++i) C[acc_row(i, h) * 32 + r] = hi[i] + lo[i];
}

template <int BF>
__global__ void rate_kernel(float* out, int iters) {
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)1.0f; }
    const float af = (float)threadIdx.x, bf = 1.0f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (BF) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
            else acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[q], 0, 0, 0);
        }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
    if (s == 12345.f) out[0] = s;
}

int main() {
    const int K = 16384;
    std::vector<float> A(32 * K), B(K * 32);
    srand(7);
    auto nrm = [] { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return (float)(sqrt(-2 * log(u)) * cos(6.283185307179586 * v)); };
    for (auto& v : A) v = nrm();
    for (auto& v : B) v = nrm();
    std::vector<double> ref(1024, 0.0);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * 32 + j]; ref[i * 32 + j] = s; }
    double rms = 0; for (double v : ref) rms += v * v; rms = sqrt(rms / 1024);
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 4096));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> C(1024);
    auto report = [&](const char* name) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
        double e2 = 0, emax = 0;
        for (int i = 0; i < 1024; ++i) { const double e = C[i] - ref[i]; e2 += e * e; emax = fmax(emax, fabs(e)); }
        printf("%-28s rms error %.3e, max %.3e of the result's rms (K = %d)\n", name, sqrt(e2 / 1024) / rms, emax / rms, K);
    };
    hipLaunchKernelGGL(gemm_fp32, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); report("fp32 MFMA (fmaf chain)");
    hipLaunchKernelGGL(gemm_bf16x3, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 6); report("bf16 x 3, six products");
    hipLaunchKernelGGL(gemm_bf16x3, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, 3); report("bf16 x 2, three products");
    // issue rates: 4 waves per SIMD on every CU
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int bf = 0; bf < 2; ++bf) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (bf) hipLaunchKernelGGL(rate_kernel<1>, dim3(256 * 4), dim3(256), 0, 0, dC, iters);
            else hipLaunchKernelGGL(rate_kernel<0>, dim3(256 * 4), dim3(256), 0, 0, dC, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = 2.0 * 32 * 32 * (bf ? 16 : 2) * 4.0 * iters * (256.0 * 4 * 4);
        printf("%s: %.1f TFLOP/s%s\n", bf ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32  ", flop / ms / 1e9,
               bf ? "  (/ 6 products = the fp32-equivalent rate of the split form, operand splitting not counted)" : "");
    }
    return 0;
}
