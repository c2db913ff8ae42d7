// Internal launcher declarations shared by the translation units of libzns_hip.so.
#pragma once
#include <thread>

#include "zns_common.h"

namespace zns {

// bench.py measurement hook (zns_probe_arm): events around the main kernel of matching launches
// ONE probe per process: arming while another probe is armed is refused, only the arming thread may disarm
// (zns_probe_arm).  Launches from any thread match (autograd's backward thread included).
struct Probe {
    int kind = -1, cin = 0, cout = 0, kh = 0, n = 0, count = 0;
    hipEvent_t* start = nullptr;
    hipEvent_t* stop = nullptr;
    std::thread::id owner;
};
extern Probe g_probe;
inline bool probe_match(int kind, int cin, int cout, int kh) {
    return g_probe.n > 0 && g_probe.kind == kind && g_probe.cin == cin && g_probe.cout == cout && g_probe.kh == kh;
}

struct ConvArgs {
    int nbr, B, T, cin, cout, kh, kw, H, pool;
};

// zns_naive.hip
// x: per-branch input rows (x.p[br] = the branch's B samples; one stacked tensor is {x, x + B * cin * H * T})
int naive_fwd(const ConvArgs& a, Ptr2 x, Ptr2 w, Ptr2 bias, float* y, uint8_t* argmax, float drop_p,
              uint32_t seed, int layer, hipStream_t st);
int naive_dgrad(const ConvArgs& a, const float* dz, Ptr2 w, float* dx, const float* x, float inv_keep,
                const uint8_t* prev_argmax, int prev_pool, hipStream_t st);
int naive_wgrad(const ConvArgs& a, const float* dz, Ptr2 x, MPtr2 dw, int accumulate, hipStream_t st);
int channel_sum(int nbr, int B, int C, long long HT, const float* dz, MPtr2 db, int accumulate, hipStream_t st);

// zns_conv_mfma.hip -- return ZNS_ERR_ARG (without setting an error) when the shape has no instance
bool mfma_supported(const ConvArgs& a);
size_t mfma_workspace_bytes(const ConvArgs& a);
// wpk (nullable): operand already packed by mfma_pack for the same geometry -- the call then skips its own packing
int mfma_fwd(const ConvArgs& a, const float* x, Ptr2 w, Ptr2 bias, float* y, uint8_t* argmax, float drop_p,
             uint32_t seed, int layer, const float* wpk, void* ws, size_t ws_bytes, hipStream_t st);
int mfma_dgrad(const ConvArgs& a, const float* dz, Ptr2 w, float* dx, const float* x, float inv_keep,
               const uint8_t* prev_argmax, int prev_pool, const float* wpk, void* ws, size_t ws_bytes, hipStream_t st);
int mfma_pack_layout(const ConvArgs& a, int transposed);     // > 0: id of the operand layout this geometry's kernel reads
size_t mfma_pack_bytes(const ConvArgs& a, int transposed);
int mfma_pack(const ConvArgs& a, Ptr2 w, int transposed, float* dst, hipStream_t st);
int mfma_wgrad(const ConvArgs& a, const float* dz, const float* x, MPtr2 dw, int accumulate, void* ws,
               size_t ws_bytes, hipStream_t st);

// zns_conv_fft.hip -- frequency-domain (overlap-save) form of cv2..cv6, same results, ~6x fewer MFMA FLOPs
bool fft_supported(const ConvArgs& a);
bool fft_wgrad_supported(const ConvArgs& a);
size_t fft_workspace_bytes(const ConvArgs& a);
size_t fft_spectrum_bytes(const ConvArgs& a);   // input spectrum shared by forward and backward-weights
size_t fft_dz_spectrum_bytes(const ConvArgs& a);   // dZ spectrum shared by backward-weights and backward-data
int fft_fwd(const ConvArgs& a, const float* x, Ptr2 w, Ptr2 bias, float* y, uint8_t* argmax, float drop_p,
            uint32_t seed, int layer, float* xspec, const float* wpk, void* ws, size_t ws_bytes, hipStream_t st);
int fft_dgrad(const ConvArgs& a, const float* dz, Ptr2 w, float* dx, const float* x, float inv_keep,
              const uint8_t* prev_argmax, int prev_pool, const float* dzspec, const float* wpk, void* ws,
              size_t ws_bytes, hipStream_t st);
int fft_pack_layout(const ConvArgs& a, int transposed);      // > 0: id of the frequency-weight layout (wide / narrow tile variant)
size_t fft_pack_bytes(const ConvArgs& a, int transposed);
int fft_pack(const ConvArgs& a, Ptr2 w, int transposed, float* dst, hipStream_t st);
int fft_wgrad(const ConvArgs& a, const float* dz, const float* x, MPtr2 dw, MPtr2 db, int accumulate,
              const float* xspec, float* dzspec_out, void* ws, size_t ws_bytes, hipStream_t st);   // also writes the bias gradient

// zns_cv1.hip -- direct (HBM-bound) kernels for the cin==1 first layer
int cv1_fwd(const ConvArgs& a, Ptr2 x, Ptr2 w, Ptr2 bias, float* y, float drop_p, uint32_t seed,
            int layer, hipStream_t st);
int cv1_wgrad(const ConvArgs& a, const float* dz, Ptr2 x, MPtr2 dw, MPtr2 db, int accumulate, void* ws,
              size_t ws_bytes, hipStream_t st);
size_t cv1_workspace_bytes(const ConvArgs& a);

}  // namespace zns
