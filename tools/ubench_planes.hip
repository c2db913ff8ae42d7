// Microbenchmark: how fast can a thread block read 96 planes that are C*P*4 bytes apart (256 B per wave per plane)
// versus the same bytes laid out contiguously per wave (24 KB runs)?  Decides the spectrum layout.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void __launch_bounds__(256) rd(const float* __restrict__ S, float* __restrict__ out, int C, int P, int chunks, int mode) {
    const int chunk = blockIdx.x % chunks, plane = blockIdx.x / chunks;
    const int c = plane % C, n = plane / C;
    const int p = chunk * 256 + threadIdx.x;
    if (p >= P) return;
    float acc = 0.f;
    if (mode == 0) {   // [n][slot][c][p]
        const float* in = S + (((size_t)n * 96) * C + c) * P + p;
        const size_t ts = (size_t)C * P;
#pragma unroll
        for (int s = 0; s < 96; ++s) acc += in[s * ts];
    } else {           // [n][c][p/64][slot][64]
        const int npc = (P + 63) / 64;
        const float* in = S + ((((size_t)n * C + c) * npc + (p >> 6)) * 96) * 64 + (p & 63);
#pragma unroll
        for (int s = 0; s < 96; ++s) acc += in[s * 64];
    }
    out[((size_t)plane) * P + p] = acc;
}
int main() {
    const int nb = 32, C = 128, P = 352;
    const size_t nfl = (size_t)nb * C * ((P + 63) / 64) * 96 * 64;
    float *S, *out;
    hipMalloc(&S, nfl * 4); hipMalloc(&out, (size_t)nb * C * P * 4);
    hipMemset(S, 0, nfl * 4);
    const int chunks = (P + 255) / 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; ++mode) {
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(rd, dim3(nb * C * chunks), dim3(256), 0, 0, S, out, C, P, chunks, mode);
        hipEventRecord(a);
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(rd, dim3(nb * C * chunks), dim3(256), 0, 0, S, out, C, P, chunks, mode);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
        const double bytes = (double)nb * C * P * 96 * 4;
        printf("mode %d: %.3f ms  %.2f TB/s (of %.0f MB)\n", mode, ms, bytes / ms / 1e9, bytes / 1e6);
    }
    return 0;
}
