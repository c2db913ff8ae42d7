// Microbenchmark for the "2-D frequency domain" idea: the GEMM stage becomes
// pointwise per (time frequency, row frequency): Y[slot][co][col] = sum_ci G[slot][co][ci] X[slot][ci][col] with
// K = Cin only, i.e. ~27 FLOP/byte.  Runs the shipped conv_core template with one tap on cv4's shape
// (2 x 6144 slots of 128 x 128 x 352) to see how close to the HBM roofline it streams.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zeronotesamba_amd/csrc tools/ubench_pointwise.hip -o /tmp/ubench_pointwise
#include <cstdio>
#include <vector>
#include "zns_conv_core.h"

namespace zns {
thread_local char g_err[512];
void set_error(const char*, ...) {}
}  // namespace zns
using namespace zns;

template <class C>
static float run(int slots, int P, int iters) {
    const size_t nx = (size_t)slots * C::CIN * P, ny = (size_t)slots * C::COUT * P;
    const size_t nw = (size_t)slots * C::COUT * C::CIN;
    float *x, *y, *w;
    hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4); hipMalloc(&w, nw * 4);
    hipMemset(x, 0, nx * 4); hipMemset(w, 0, nw * 4);
    hipFuncSetAttribute((const void*)conv_core<C>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    CoreArgs A{};
    A.x = x; A.wpk = w; A.dx = y; A.xmask = nullptr; A.inv_keep = 1.f; A.prev_pool = 1;
    A.B = 1; A.T = P; A.N = slots; A.F = slots; A.dil = 1; A.ncb = (P + C::WB - 1) / C::WB;
    const unsigned blocks = (unsigned)((size_t)A.N * A.ncb * (C::COUT / C::COB));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(conv_core<C>, dim3(blocks), dim3(C::NT), C::LDS_BYTES, 0, A);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(conv_core<C>, dim3(blocks), dim3(C::NT), C::LDS_BYTES, 0, A);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    const double bytes = (double)(nx + ny + nw) * 4, flop = 2.0 * slots * C::COUT * C::CIN * P;
    printf("slots %d  %dx%dx%d  LDS %d KB  waves %d: %.3f ms  %.2f TB/s  %.1f TFLOP/s  (%s)\n", slots, C::COUT, C::CIN, P,
           C::LDS_BYTES / 1024, C::NW, ms, bytes / ms / 1e9, flop / ms / 1e9, hipGetErrorString(hipGetLastError()));
    hipFree(x); hipFree(y); hipFree(w);
    return ms;
}

int main() {
    //                 CIN COUT KH KW H MT RT CT WM WR WC CK POOL EPI DILMAX HET
    run<ConvCfg<128, 128, 1, 1, 1, 4, 1, 1, 1, 1, 12, 32, 1, 1, 12, 1>>(2 * 6144, 352, 5);   // cv4, N_h = 64
    run<ConvCfg<128, 128, 1, 1, 1, 4, 1, 1, 1, 1, 12, 16, 1, 1, 12, 1>>(2 * 6144, 352, 5);
    run<ConvCfg<64, 64, 1, 1, 1, 2, 1, 2, 1, 1, 4, 32, 1, 1, 12, 0>>(2 * 6144, 640, 5);       // cv2, two row windows
    run<ConvCfg<256, 256, 1, 1, 1, 2, 1, 1, 4, 1, 3, 32, 1, 1, 12, 0>>(2 * 1536, 384, 5);     // cv6, N_h = 16
    return 0;
}
