// Shared device/host helpers for libzns_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/zns.h"
#include "../../include/zns_test.h"

namespace zns {

void set_error(const char* fmt, ...);

#define ZNS_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            zns::set_error(__VA_ARGS__);         \
            return ZNS_ERR_ARG;                  \
        }                                        \
    } while (0)

#define ZNS_HIP(expr)                                                                  \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess) {                                                       \
            zns::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return (int)e__;                                                           \
        }                                                                              \
    } while (0)

#define ZNS_LAUNCH_CHECK()                                                             \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            zns::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
            return (int)e__;                                                           \
        }                                                                              \
    } while (0)

// ---- stateless dropout hash (restated in oracle/dropout_ref.py) ----
__host__ __device__ inline uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint32_t hash_u32(uint32_t seed, uint64_t idx) {
    uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    return lowbias32((lo * 0x9E3779B1u) ^ seed ^ (hi * 0x7FEB352Du));
}
__host__ __device__ inline uint32_t layer_seed(uint32_t seed, int layer, int branch) {
    uint32_t s = lowbias32(seed + (uint32_t)(layer + 1) * 0x9E3779B1u);
    return lowbias32(s ^ ((uint32_t)(branch + 1) * 0x846CA68Bu));
}
__host__ __device__ inline uint32_t drop_threshold(float p) { return (uint32_t)(p * 16777216.0f); }
__device__ inline bool drop_keep(uint32_t seed, uint64_t idx, uint32_t thresh) {
    return (hash_u32(seed, idx) >> 8) >= thresh;
}

// Non-finite values go through the fused epilogues as they go through ATen's ops (models.py:37-70 on the reference's side):
//   max_pool2d   a NaN in the window wins (`val > max || isnan(val)`; the LAST NaN's position is the arg-max)
//   relu         relu(NaN) = NaN (threshold keeps what is not <= 0), relu(+inf) = +inf
//   dropout      input * mask / (1 - p): a dropped NaN or inf is NaN (x * 0), not 0
// A diverged run must look diverged here too: a plain `v > best` / `v > 0 ? v : 0` / `keep ? v : 0` would swallow the NaN.
__device__ __forceinline__ bool pool_takes(float v, float best) { return v > best || v != v; }
__device__ __forceinline__ float relu_nan(float v) { return v <= 0.f ? 0.f : v; }
__device__ __forceinline__ float drop_apply(bool keep, float v, float inv_keep) { return keep ? v * inv_keep : v * 0.f; }

// Per-branch pointer pack passed by value to kernels (nbr <= 2).
struct Ptr2 {
    const float* p[2];
};
struct MPtr2 {
    float* p[2];
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Tuning / diagnostic knobs of tests and tools: set ONLY through zns_test_set_tuning (include/zns_test.h), never read from
// the environment -- a stray variable in a training job must not change sums or skip stores.  Defaults = the product path.
struct Tuning {
    int v[ZNS_TUNE_COUNT];
};
extern Tuning g_tuning;
static inline int tuning(int key) { return g_tuning.v[key]; }

// Function attributes (MaxDynamicSharedMemorySize) and device allocations are PER DEVICE: "done once" state is keyed
// by the calling thread's current device, so a process that drives several GPUs (e.g. the reference's branch-per-device
// placement, pretext.py:199-200) configures each of them.
constexpr int ZNS_MAX_DEVICES = 64;
static inline int current_device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
    return d % ZNS_MAX_DEVICES;
}

}  // namespace zns
