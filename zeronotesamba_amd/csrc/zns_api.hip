// extern "C" entry points for the conv layers + error plumbing (see include/zns.h).
#include <stdarg.h>

#include <atomic>
#include <mutex>
#include <string.h>

#include "zns_common.h"
#include "zns_kernels.h"

namespace zns {

static thread_local char g_err[512] = "";
static std::atomic<int> g_impl{ZNS_IMPL_AUTO};   // process-wide ON PURPOSE: autograd runs backward() on its own thread, which
                                                  // must launch the family the test selected on the Python thread
Probe g_probe;
Tuning g_tuning = [] {
    Tuning t{};
    t.v[ZNS_TUNE_KSPLIT] = -1;   // automatic
    return t;
}();

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static bool is_cv1(const ConvArgs& a) { return a.cin == 1 && a.cout == 64 && a.kh == 3 && a.kw == 11; }

static int check_conv(const ConvArgs& a, const char* who) {
    ZNS_CHECK_ARG(a.nbr >= 1 && a.nbr <= 2, "%s: nbr=%d not in {1,2}", who, a.nbr);
    ZNS_CHECK_ARG(a.B >= 1 && a.T >= 1 && a.H >= 1, "%s: empty batch/time axis (B=%d T=%d H=%d)", who, a.B, a.T, a.H);
    ZNS_CHECK_ARG(a.cin >= 1 && a.cout >= 1 && a.kh >= 1 && a.kw >= 1 && (a.kh & 1) && (a.kw & 1),
                  "%s: bad conv geometry (cin=%d cout=%d k=%dx%d: odd kernels only)", who, a.cin, a.cout, a.kh, a.kw);
    ZNS_CHECK_ARG(a.pool >= 1 && a.H % a.pool == 0, "%s: pool=%d does not divide H=%d", who, a.pool, a.H);
    return ZNS_OK;
}

}  // namespace zns

using namespace zns;

extern "C" {

const char* zns_last_error(void) { return g_err; }
int zns_version(void) { return 105; }   // 105: separate branch inputs (x1 / demb1), zns_scale_f32, float64 NT-Xent tables   // 101: dzspec_out / dzspec, 102: stem mix + RMS gate, 103: NT-Xent workspace (any batch size), 104: packed-weight hand-off, staged xqt, multi-tensor Adam
int zns_set_impl(int impl) {
    return g_impl.exchange(impl);
}

int zns_test_set_tuning(int key, int value) {
    if (key < 0 || key >= ZNS_TUNE_COUNT) {
        zns::set_error("zns_test_set_tuning: unknown key %d", key);
        return ZNS_ERR_ARG;
    }
    zns::g_tuning.v[key] = value;
    return ZNS_OK;
}
int zns_test_get_tuning(int key) { return key >= 0 && key < ZNS_TUNE_COUNT ? zns::g_tuning.v[key] : 0; }

static std::mutex g_probe_mu;
int zns_probe_arm(int kind, int cin, int cout, int kh, void* const* ev_start, void* const* ev_stop, int n) {
    std::lock_guard<std::mutex> lock(g_probe_mu);
    if (g_probe.n > 0 && (n > 0 || g_probe.owner != std::this_thread::get_id())) {
        set_error("probe_arm: %s", n > 0 ? "a probe is already armed (disarm it first: n = 0)"
                                         : "only the thread that armed the probe may disarm it");
        return ZNS_ERR_ARG;
    }
    delete[] g_probe.start;
    delete[] g_probe.stop;
    g_probe = Probe{};
    if (n <= 0) return ZNS_OK;
    if (ev_start == nullptr || ev_stop == nullptr) { set_error("probe_arm: null event arrays"); return ZNS_ERR_ARG; }
    g_probe.kind = kind; g_probe.cin = cin; g_probe.cout = cout; g_probe.kh = kh; g_probe.n = n;
    g_probe.owner = std::this_thread::get_id();
    g_probe.start = new hipEvent_t[n];
    g_probe.stop = new hipEvent_t[n];
    for (int i = 0; i < n; ++i) { g_probe.start[i] = (hipEvent_t)ev_start[i]; g_probe.stop[i] = (hipEvent_t)ev_stop[i]; }
    return ZNS_OK;
}
int zns_probe_count(void) { return g_probe.count; }

// the size queries take the same arguments as the calls they serve and must survive what those calls refuse (an empty
// batch or time axis would otherwise reach the tile pickers' divisions)
static bool degenerate(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H) {
    return nbr < 1 || B < 1 || T < 1 || cin < 1 || cout < 1 || kh < 1 || kw < 1 || H < 1;
}

size_t zns_conv2d_workspace_bytes(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H) {
    if (degenerate(nbr, B, T, cin, cout, kh, kw, H)) return 256;
    ConvArgs a{nbr, B, T, cin, cout, kh, kw, H, 1};
    size_t n = 256;
    if (is_cv1(a)) n = cv1_workspace_bytes(a);
    else if (mfma_supported(a)) n = mfma_workspace_bytes(a);
    if (fft_supported(a) && fft_workspace_bytes(a) > n) n = fft_workspace_bytes(a);
    return (n + 255) & ~(size_t)255;
}

size_t zns_conv2d_spectrum_bytes(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H) {
    if (degenerate(nbr, B, T, cin, cout, kh, kw, H)) return 0;
    ConvArgs a{nbr, B, T, cin, cout, kh, kw, H, 1};
    return (g_impl == ZNS_IMPL_AUTO && fft_wgrad_supported(a)) ? fft_spectrum_bytes(a) : 0;
}

size_t zns_conv2d_dz_spectrum_bytes(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H) {
    if (degenerate(nbr, B, T, cin, cout, kh, kw, H)) return 0;
    ConvArgs a{nbr, B, T, cin, cout, kh, kw, H, 1};
    return (g_impl == ZNS_IMPL_AUTO && fft_wgrad_supported(a)) ? fft_dz_spectrum_bytes(a) : 0;
}

// which kernel family a (forward / backward-data) call of this geometry runs on: 0 naive, 1 cv1, 2 frequency domain, 3 direct
static int conv_family(const ConvArgs& a, bool forward) {
    if (g_impl == ZNS_IMPL_NAIVE) return 0;
    if (forward && is_cv1(a) && a.pool == 1) return 1;
    if (g_impl == ZNS_IMPL_AUTO && fft_supported(a)) return 2;
    if (mfma_supported(a)) return 3;
    return -1;
}

size_t zns_conv2d_packed_bytes(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H, int transposed,
                               int* layout_id) {
    if (degenerate(nbr, B, T, cin, cout, kh, kw, H)) {
        if (layout_id) *layout_id = 0;
        return 0;
    }
    ConvArgs a{nbr, B, T, cin, cout, kh, kw, H, 1};
    const int fam = conv_family(a, !transposed);
    int id = 0;
    size_t n = 0;
    if (fam == 2) { id = fft_pack_layout(a, transposed); n = fft_pack_bytes(a, transposed); }
    else if (fam == 3) { id = mfma_pack_layout(a, transposed); n = mfma_pack_bytes(a, transposed); }
    if (layout_id) *layout_id = id;
    return id ? n : 0;
}

int zns_conv2d_pack_f32(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H, int transposed,
                        const float* const* w, float* dst, size_t dst_bytes, void* stream) {
    ConvArgs a{nbr, B, T, cin, cout, kh, kw, H, 1};
    int rc = check_conv(a, "conv2d_pack");
    if (rc) return rc;
    int id = 0;
    const size_t need = zns_conv2d_packed_bytes(nbr, B, T, cin, cout, kh, kw, H, transposed, &id);
    ZNS_CHECK_ARG(id != 0, "conv2d_pack: this geometry's kernels take the weights as they are (nothing to pack)");
    if (dst == nullptr || dst_bytes < need) { set_error("conv2d_pack: destination %zu < %zu", dst_bytes, need); return ZNS_ERR_WORKSPACE; }
    Ptr2 pw{};
    for (int i = 0; i < nbr; ++i) pw.p[i] = w[i];
    hipStream_t st = (hipStream_t)stream;
    return conv_family(a, !transposed) == 2 ? fft_pack(a, pw, transposed, dst, st) : mfma_pack(a, pw, transposed, dst, st);
}

// the per-branch input rows of a call: one stacked tensor, or (first layer only) the two branches' own tensors
static int branch_inputs(const ConvArgs& a, const float* x, const float* x1, const char* who, Ptr2* out) {
    ZNS_CHECK_ARG(x1 == nullptr || (a.nbr == 2 && a.cin == 1),
                  "%s: separate branch inputs (x1) are taken by the first layer of a two-branch stack only", who);
    out->p[0] = x;
    out->p[1] = x1 ? x1 : x + (size_t)a.B * a.cin * a.H * a.T;
    return ZNS_OK;
}

int zns_conv2d_fwd_f32(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H, int pool, const float* x,
                       const float* x1,
                       const float* const* w, const float* const* bias, float* y, uint8_t* argmax, float drop_p,
                       uint32_t seed, int layer, float* xspec, const float* wpk, void* ws, size_t ws_bytes,
                       void* stream) {
    ConvArgs a{nbr, B, T, cin, cout, kh, kw, H, pool};
    int rc = check_conv(a, "conv2d_fwd");
    if (rc) return rc;
    ZNS_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "conv2d_fwd: drop_p=%f", drop_p);
    Ptr2 pw{}, pb{};
    for (int i = 0; i < nbr; ++i) { pw.p[i] = w[i]; pb.p[i] = bias[i]; }
    hipStream_t st = (hipStream_t)stream;
    Ptr2 px{};
    rc = branch_inputs(a, x, x1, "conv2d_fwd", &px);
    if (rc) return rc;
    if (g_impl == ZNS_IMPL_NAIVE) return naive_fwd(a, px, pw, pb, y, argmax, drop_p, seed, layer, st);
    if (is_cv1(a) && pool == 1) return cv1_fwd(a, px, pw, pb, y, drop_p, seed, layer, st);
    ZNS_CHECK_ARG(x1 == nullptr, "conv2d_fwd: separate branch inputs are taken by the cv1 kernels only");
    if (g_impl == ZNS_IMPL_AUTO && fft_supported(a))
        return fft_fwd(a, x, pw, pb, y, argmax, drop_p, seed, layer, xspec, wpk, ws, ws_bytes, st);
    if (mfma_supported(a)) return mfma_fwd(a, x, pw, pb, y, argmax, drop_p, seed, layer, wpk, ws, ws_bytes, st);
    set_error("conv2d_fwd: no gfx950 kernel instance for cin=%d cout=%d k=%dx%d H=%d", cin, cout, kh, kw, H);
    return ZNS_ERR_ARG;
}

int zns_conv2d_dgrad_f32(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H, const float* dz,
                         const float* const* w, float* dx, const float* x, float inv_keep,
                         const uint8_t* prev_argmax, int prev_pool, const float* dzspec, const float* wpk, void* ws,
                         size_t ws_bytes, void* stream) {
    ConvArgs a{nbr, B, T, cin, cout, kh, kw, H, 1};
    int rc = check_conv(a, "conv2d_dgrad");
    if (rc) return rc;
    ZNS_CHECK_ARG(prev_pool >= 1 && (prev_pool == 1 || prev_argmax != nullptr), "conv2d_dgrad: prev_pool/argmax");
    Ptr2 pw{};
    for (int i = 0; i < nbr; ++i) pw.p[i] = w[i];
    hipStream_t st = (hipStream_t)stream;
    if (g_impl == ZNS_IMPL_NAIVE) return naive_dgrad(a, dz, pw, dx, x, inv_keep, prev_argmax, prev_pool, st);
    if (g_impl == ZNS_IMPL_AUTO && fft_supported(a))
        return fft_dgrad(a, dz, pw, dx, x, inv_keep, prev_argmax, prev_pool, dzspec, wpk, ws, ws_bytes, st);
    if (mfma_supported(a)) return mfma_dgrad(a, dz, pw, dx, x, inv_keep, prev_argmax, prev_pool, wpk, ws, ws_bytes, st);
    set_error("conv2d_dgrad: no gfx950 kernel instance for cin=%d cout=%d k=%dx%d H=%d", cin, cout, kh, kw, H);
    return ZNS_ERR_ARG;
}

int zns_conv2d_wgrad_f32(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H, const float* dz,
                         const float* x, const float* x1, float* const* dw, float* const* db, int accumulate, const float* xspec,
                         float* dzspec_out, void* ws, size_t ws_bytes, void* stream) {
    ConvArgs a{nbr, B, T, cin, cout, kh, kw, H, 1};
    int rc = check_conv(a, "conv2d_wgrad");
    if (rc) return rc;
    MPtr2 pdw{}, pdb{};
    for (int i = 0; i < nbr; ++i) { pdw.p[i] = dw[i]; pdb.p[i] = db[i]; }
    hipStream_t st = (hipStream_t)stream;
    Ptr2 px{};
    rc = branch_inputs(a, x, x1, "conv2d_wgrad", &px);
    if (rc) return rc;
    if (g_impl == ZNS_IMPL_NAIVE) {
        rc = naive_wgrad(a, dz, px, pdw, accumulate, st);
        if (rc) return rc;
        return channel_sum(nbr, B, cout, (long long)H * T, dz, pdb, accumulate, st);
    }
    if (is_cv1(a)) return cv1_wgrad(a, dz, px, pdw, pdb, accumulate, ws, ws_bytes, st);
    ZNS_CHECK_ARG(x1 == nullptr, "conv2d_wgrad: separate branch inputs are taken by the cv1 kernels only");
    if (g_impl == ZNS_IMPL_AUTO && fft_wgrad_supported(a))
        return fft_wgrad(a, dz, x, pdw, pdb, accumulate, xspec, dzspec_out, ws, ws_bytes, st);
    if (mfma_supported(a)) {
        rc = mfma_wgrad(a, dz, x, pdw, accumulate, ws, ws_bytes, st);
        if (rc) return rc;
        return channel_sum(nbr, B, cout, (long long)H * T, dz, pdb, accumulate, st);
    }
    set_error("conv2d_wgrad: no gfx950 kernel instance for cin=%d cout=%d k=%dx%d H=%d", cin, cout, kh, kw, H);
    return ZNS_ERR_ARG;
}

}  // extern "C"
