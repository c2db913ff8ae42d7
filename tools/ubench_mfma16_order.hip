// Is v_mfma_f32_16x16x4_f32 bit for bit the fmaf chain k = 0, 1, 2, 3 per output (as 32x32x2 is for k = 0, 1)?
// hipcc --offload-arch=gfx950 -O2 tools/ubench_mfma16_order.hip -o tools/ubench_mfma16_order && ./tools/ubench_mfma16_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k16(const float* A, const float* B, float* D, int ksteps) {   // A [16][4*ksteps], B [4*ksteps][16]
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < ksteps; ++s) {
        const float a = A[(l & 15) * 4 * ksteps + 4 * s + (l >> 4)];
        const float b = B[(4 * s + (l >> 4)) * 16 + (l & 15)];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}

int main() {
    const int ks = 32, K = 4 * ks;
    std::vector<float> A(16 * K), B(K * 16), D(256), R(256);
    srand(1);
    for (auto& v : A) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : B) v = ((float)rand() / RAND_MAX - 0.5f) * 3.f;
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD, ks);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad_chain = 0, bad_pair = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float c = 0.f;
            for (int k = 0; k < K; ++k) c = fmaf(A[i * K + k], B[k * 16 + j], c);
            if (c != D[i * 16 + j]) ++bad_chain;
            // alternative: exact 4-term dot added once per instruction
            double p = 0.0; float c2 = 0.f;
            for (int s = 0; s < ks; ++s) {
                p = (double)c2;
                for (int k = 0; k < 4; ++k) p += (double)A[i * K + 4 * s + k] * (double)B[(4 * s + k) * 16 + j];
                c2 = (float)p;
            }
            if (c2 != D[i * 16 + j]) ++bad_pair;
        }
    printf("16x16x4 f32: mismatches vs fmaf chain k=0..3: %d / 256; vs one-rounding-per-instruction: %d / 256\n", bad_chain, bad_pair);
    return 0;
}
