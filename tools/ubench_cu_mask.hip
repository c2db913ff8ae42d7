// Which XCD does bit n of a hipExtStreamCreateWithCUMask mask belong to?  A kernel of many one-wave blocks records the
// XCC_ID / HW_ID registers of where it ran under a few test masks; per mask: the number of distinct CUs used in each XCD.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_cu_mask.hip -o tools/ubench_cu_mask && ./tools/ubench_cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

__global__ void where(uint32_t* out) {
    uint32_t xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    uint64_t t0 = clock64();
    while (clock64() - t0 < 200000) {}          // long enough that the blocks spread over every CU the mask allows
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 8) | ((hwid >> 8) & 0xff);   // (xcc, se / sh / cu)
}

static void run(const char* name, const std::vector<uint32_t>& m, uint32_t* d) {
    const int NB = 4096;
    std::vector<uint32_t> h(NB);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    hipLaunchKernelGGL(where, dim3(NB), dim3(64), 0, s, d);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), d, NB * 4, hipMemcpyDeviceToHost);
    (void)hipStreamDestroy(s);
    std::set<uint32_t> seen(h.begin(), h.end());
    int per[16] = {0};
    for (uint32_t v : seen) per[(v >> 8) & 15]++;
    printf("%-44s distinct CUs per XCD:", name);
    for (int x = 0; x < 8; ++x) printf(" %2d", per[x]);
    printf("   total %zu\n", seen.size());
}

int main() {
    int ncu = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int words = (ncu + 31) / 32;
    uint32_t* d;
    (void)hipMalloc(&d, 4096 * 4);
    printf("CUs %d, mask words %d\n", ncu, words);
    std::vector<uint32_t> all(words, 0xffffffffu);
    run("all bits set", all, d);
    { auto m = all; for (int w = 0; w < words; ++w) m[w] &= ~(1u << 31); run("bits 31, 63, .. cleared (bench.py round 4)", m, d); }
    { auto m = all; m[0] &= ~0xffu; run("bits 0..7 cleared", m, d); }
    { std::vector<uint32_t> m(words, 0u); m[0] = 0xffffffffu; run("only bits 0..31 set", m, d); }
    { std::vector<uint32_t> m(words, 0u); for (int b = 0; b < ncu; b += 8) m[b / 32] |= 1u << (b % 32); run("only bits n % 8 == 0 set", m, d); }
    { std::vector<uint32_t> m(words, 0u); m[0] = 0xffu; run("only bits 0..7 set", m, d); }
    return 0;
}
