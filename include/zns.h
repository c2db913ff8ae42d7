/*
 * zns.h -- C ABI of libzns_hip.so: the MI355X (gfx950) implementation of the
 * zeroNoteSamba self-supervised pretext hot path.
 *
 * The reference (deezer/zeroNoteSamba) is pure Python: its "FFI" for this path is
 * the call signature of generate_XQT() and the torch.nn.Module / autograd protocol
 * of _CNN / DS_CNN / Pretext_CNN / Down_CNN / NTXent.  Each
 * entry point below names the reference lines it replaces.  INTEGRATION.md shows
 * the ctypes binding a maintainer would add on the reference side.  This header is the
 * PRODUCT surface; the process-global switches and self-tests that the test suite
 * and bench.py use live in include/zns_test.h (same library) and are not for binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter says "host";
 *     `const float* const* w` is a HOST array of device pointers (one per branch)
 *   - tensors are contiguous fp32, NCHW, exactly as the reference modules hold them
 *   - `nbr` = number of independent-weight branches processed by one launch
 *     (2 for Pretext_CNN: anchor + postve, 1 for a lone DS_CNN); activations are
 *     stacked branch-major: [nbr*B, C, H, T]
 *   - `stream` is a hipStream_t (NULL = default stream); all work is enqueued, no
 *     host synchronisation, no allocation: the caller owns outputs and workspace
 *   - return value: 0 on success, otherwise a negative zns error or a positive
 *     hipError_t; zns_last_error() gives the message (thread local)
 */
#ifndef ZNS_H
#define ZNS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZNS_OK 0
#define ZNS_ERR_ARG (-1)         /* bad argument / unsupported shape */
#define ZNS_ERR_WORKSPACE (-2)   /* workspace too small */
#define ZNS_ERR_MODE (-3)        /* generate_XQT: "Mode can only be vqt or cqt!" (input_rep.py:56-57) */

#define ZNS_IMPL_AUTO 0          /* fastest kernels: frequency-domain MFMA path where it applies, else direct MFMA */
#define ZNS_IMPL_NAIVE 1         /* one-thread-per-output HIP kernels, for on-device validation only */
#define ZNS_IMPL_DIRECT 2        /* direct (time-domain) implicit-GEMM MFMA kernels only */

#define ZNS_REDUCE_MAX 0         /* Down_CNN(reduction="max"), models.py:147-148 */
#define ZNS_REDUCE_MEAN 1        /* Down_CNN(reduction="mean"), models.py:144-145 */

#define ZNS_MODE_VQT 0           /* input_rep.py:41-54 */
#define ZNS_MODE_CQT 1           /* input_rep.py:26-39 */

const char* zns_last_error(void);
int zns_version(void);
/* ------------------------------------------------------------------ conv layers
 * One call = one `nn.Conv2d` (+ the MaxPool2d/ReLU/Dropout that follow it) of
 * _CNN.forward, models.py:37-70, for `nbr` branches at once.
 *   y = dropout(relu(maxpool_{(pool,1)}(conv2d(x, w, b, padding=(kh/2, kw/2)))))
 * argmax (may be NULL when pool==1 or for inference) receives, per output element,
 * the row offset 0..pool-1 of the pooled maximum (first maximum wins, as ATen).
 * drop_p == 0 disables dropout (module.eval()).  The keep mask is a stateless hash
 * of (seed, layer, branch, linear index of y within its branch); see csrc/zns_common.h.
 */
size_t zns_conv2d_workspace_bytes(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H);
/* Bytes of the input spectrum the forward pass can hand to backward-weights (0 when this geometry does not use
 * the frequency-domain kernels): pass a buffer of that size as `xspec` to BOTH calls to skip recomputing it. */
size_t zns_conv2d_spectrum_bytes(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H);
/* Bytes of the dZ spectrum backward-weights can hand to backward-data of the same layer (0 when the geometry does
 * not use the frequency-domain kernels): pass a buffer of that size as `dzspec_out` to zns_conv2d_wgrad_f32 and
 * then as `dzspec` to zns_conv2d_dgrad_f32 -- dZ is then read and transformed once for both. */
size_t zns_conv2d_dz_spectrum_bytes(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H);

/* Packed-weight hand-off.  The MFMA kernels stream the weights in a kernel-specific order (K-stage slabs; for the
 * frequency-domain kernels: 96 real operators per branch built from the 64-point DFT of the time taps, 5-7x the size of
 * the weights).  By default every forward / backward-data call builds that operand itself.  A caller whose weights do
 * not change between calls (module.eval(), val_epoch pretext.py:527-592, inference sample_script.py:46-48) packs once:
 *   bytes = zns_conv2d_packed_bytes(geometry, transposed, &layout_id)    (0 bytes / id 0: nothing to pack)
 *   zns_conv2d_pack_f32(geometry, transposed, w, dst, bytes, stream)
 * and passes `wpk = dst` to zns_conv2d_fwd_f32 (transposed = 0) / zns_conv2d_dgrad_f32 (transposed = 1).  `wpk` is valid
 * for every call whose geometry reports the SAME layout_id (the id depends on the layer, on the tile variant the
 * dispatcher picks from T, and on the kernel family a test harness may have selected) -- not only for the T it was packed at. */
size_t zns_conv2d_packed_bytes(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H, int transposed,
                               int* layout_id);
int zns_conv2d_pack_f32(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H, int transposed,
                        const float* const* w, float* dst, size_t dst_bytes, void* stream);

/* x: [nbr*B, cin, H, T], the branches' samples back to back.  x1 (nullable, first layer -- cin == 1 -- of a two-branch
 * stack only): the reference hands Pretext_CNN two separate tensors (models.py:119-122, pretext.py:482); with x1 given, x
 * holds branch 0's B samples and x1 branch 1's, and nothing is concatenated. */
int zns_conv2d_fwd_f32(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H, int pool,
                       const float* x, const float* x1, const float* const* w, const float* const* bias,
                       float* y, uint8_t* argmax,
                       float drop_p, uint32_t seed, int layer,
                       float* xspec /* nullable: receives the input spectrum, see zns_conv2d_spectrum_bytes */,
                       const float* wpk /* nullable: weights packed by zns_conv2d_pack_f32(transposed = 0) */,
                       void* ws, size_t ws_bytes, void* stream);

/* Backward-data of the same conv (autograd of models.py:37-70):
 *   dx_raw = conv_transpose(dz, w)                      [nbr*B, cin, H, T]
 * fused epilogue (all optional):
 *   g  = dx_raw * (x != 0) * inv_keep    -- ReLU+Dropout backward of the PREVIOUS layer
 *                                           (x is this conv's input = previous layer's output)
 *   if prev_pool > 1: dx[.., h*prev_pool + a, ..] = (a == prev_argmax) ? g : 0   -- MaxPool backward,
 *        dx is then [nbr*B, cin, H*prev_pool, T] = grad w.r.t. the previous conv's raw output
 *   else dx = g
 * dz is the gradient w.r.t. this conv's RAW output [nbr*B, cout, H, T].
 */
int zns_conv2d_dgrad_f32(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H,
                         const float* dz, const float* const* w, float* dx,
                         const float* x, float inv_keep, const uint8_t* prev_argmax, int prev_pool,
                         const float* dzspec /* nullable: dZ spectrum left by zns_conv2d_wgrad_f32 */,
                         const float* wpk /* nullable: weights packed by zns_conv2d_pack_f32(transposed = 1) */,
                         void* ws, size_t ws_bytes, void* stream);

/* Backward-weights: dw[br] (+)= sum_{b,h,t} dz (x) shifted x ; db[br] (+)= sum dz.
 * accumulate != 0 adds into dw/db (gradient accumulation), else overwrites. */
int zns_conv2d_wgrad_f32(int nbr, int B, int T, int cin, int cout, int kh, int kw, int H,
                         const float* dz, const float* x, const float* x1 /* as in zns_conv2d_fwd_f32 */,
                         float* const* dw, float* const* db,
                         int accumulate, const float* xspec /* nullable: spectrum saved by the forward call */,
                         float* dzspec_out /* nullable: receives the dZ spectrum, see zns_conv2d_dz_spectrum_bytes */,
                         void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ DS_CNN head
 * emb[n, t] = sigmoid(sum_c w[c] * h[n, c, t] + b)   (fc1 + sig + reshape, models.py:99-101) */
int zns_head_fwd_f32(int nbr, int B, int T, int C, const float* h, const float* const* w,
                     const float* const* b, float* emb, void* stream);
/* Given demb: dz8 = demb*emb*(1-emb) * w[c] * (h != 0) * inv_keep  (grad w.r.t. cv8 raw output),
 * dw[c] (+)= sum demb*emb*(1-emb)*h, db (+)= sum demb*emb*(1-emb).
 * demb1 (nullable, nbr == 2): branch 1's gradient rows when autograd hands the two embeddings' gradients as separate
 * tensors (demb then holds branch 0's B rows); NULL: demb is [nbr*B, T]. */
int zns_head_bwd_f32(int nbr, int B, int T, int C, const float* h, const float* emb, const float* demb,
                     const float* demb1, const float* const* w, float inv_keep, float* dz, float* const* dw, float* const* db,
                     int accumulate, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ Down_CNN combine (models.py:144-148) */
int zns_down_fwd_f32(const float* a, const float* p, size_t n, int reduction, float* out, void* stream);
int zns_down_bwd_f32(const float* a, const float* p, const float* dout, size_t n, int reduction,
                     float* da, float* dp, void* stream);

/* ------------------------------------------------------------------ NT-Xent (loss_functions.py:24-55)
 * a, p: [rows, D] (rows <= batch_len, 1 <= rows <= 2048: config.yaml's batch_size is 16).  out3 (device, 3 floats): loss,
 * mean anchor/positive cosine, mean anchor/negative cosine -- all divided by batch_len as the reference does.
 * loss_out (nullable): the loss once more, as a device scalar of its own (a framework that needs the loss as a separate
 * tensor gets it without a copy kernel).  ga/gp (nullable): d(loss*grad_scale)/da, /dp.
 * ws: zns_ntxent_workspace_bytes(rows) bytes of device scratch (float64 tables: see csrc/zns_small.hip). */
size_t zns_ntxent_workspace_bytes(int rows);
int zns_ntxent_f32(const float* a, const float* p, int rows, int D, int batch_len, float tau,
                   float* out3, float* loss_out, float* ga, float* gp, float grad_scale, void* ws, size_t ws_bytes,
                   void* stream);

/* y = x * scale[0] with `scale` a DEVICE scalar: the factor autograd hands a loss node's backward (`loss.backward()`:
 * ones; `(2 * loss).backward()`: 2), applied without a host read-back.  Two arrays per launch (NT-Xent: ga, gp); n1 = 0
 * for one.  In place (y == x) is allowed. */
int zns_scale_f32(const float* scale, const float* x0, float* y0, size_t n0, const float* x1, float* y1, size_t n1,
                  void* stream);

/* BCELoss(mean) forward + backward (loader.py:16, epochs.py:61): loss (device scalar), dout nullable */
int zns_bce_f32(const float* out, const float* target, size_t n, float* loss, float* dout,
                float grad_scale, void* stream);

/* ------------------------------------------------------------------ Adam (pretext.py:202, loader.py:43)
 * torch.optim.Adam semantics: no weight decay, no amsgrad.  `step` is 1-based.
 * g is multiplied by grad_scale first (1/world_size after an all-reduce(sum)). */
int zns_adam_step_f32(float* p, const float* g, float* m, float* v, size_t n,
                      float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                      void* stream);
/* The same update for a list of tensors in ONE launch (torch.optim.Adam over model.parameters(): 36 tensors for
 * Pretext_CNN).  table: DEVICE array of n_tensors records {float* p; const float* g; float* m; float* v; uint64 n}
 * (five 8-byte fields); max_n = the largest n.  All tensors share lr, betas, eps and the step count. */
int zns_adam_multi_f32(const void* table, int n_tensors, size_t max_n, float lr, float beta1, float beta2,
                       float eps, int step, float grad_scale, void* stream);

/* ------------------------------------------------------------------ generate_XQT (input_rep.py:11-57)
 * wav: [n_clips, n_samples] fp32 @16 kHz; out: [n_clips, 96, 1 + n_samples/256] = log(|XQT| + 1e-9).
 * mode: ZNS_MODE_VQT / ZNS_MODE_CQT, anything else -> ZNS_ERR_MODE.
 * The filter bank (host, float64) is built once per (mode) and cached on the device.
 * sample_rate: the reference's signature carries it (input_rep.py:11) and every reference caller passes 16000
 * (sample_script.py:28, pretext.py:143-147, config.yaml); this library is built for that rate only -- the octave plan
 * (hop 256 >> i, n_fft per octave, the 2:1 decimation count, the fused kernel's LDS geometry) is a compile-time table of
 * 16 kHz -- and any other value returns ZNS_ERR_ARG ("only sample_rate=16000 is built") instead of a silently different
 * transform.  A deliberate narrowing of the signature, not of any call the reference makes. */
size_t zns_xqt_workspace_bytes(int n_clips, int n_samples, int mode);
int zns_xqt_frames(int n_samples);
int zns_xqt_f32(const float* wav, int n_clips, int n_samples, int sample_rate, int mode,
                float* out, void* ws, size_t ws_bytes, void* stream);
/* The same, one stage at a time (measurement and tests): stages = 1 builds the decimation pyramid in `ws`, 2 applies
 * the filter bank to the pyramid already in `ws`, 3 = both = zns_xqt_f32.  With both stages the vqt transform runs as ONE
 * fused launch that keeps the pyramid in LDS (`ws` untouched); OR ZNS_XQT_SEPARATE into `stages` for the per-stage launches
 * instead (same arithmetic, bit-identical results, four times the HBM traffic: DESIGN.md section 5).  cqt always takes the
 * per-stage launches. */
#define ZNS_XQT_SEPARATE 4       /* per-stage launches (decimation chains, then the filter bank) */
#define ZNS_XQT_FUSED 8          /* insist on the fused kernel (error where it does not apply: cqt, ZNS_XQT_EXACT) */
#define ZNS_XQT_EXACT 16         /* validation implementation of the pyramid: resampy's tap-by-tap float32 rounding of
                                    float64 products emulated exactly (three half-rate instructions per tap, per-stage
                                    launches); the default accumulates the same taps in the same order with one float32
                                    fused multiply-add each */
int zns_xqt_stages_f32(const float* wav, int n_clips, int n_samples, int sample_rate, int mode,
                       float* out, void* ws, size_t ws_bytes, int stages, void* stream);

/* ------------------------------------------------------------------ shift batching (pretext.py:285-292,308-321)
 * bank: [n_bank, 2, 96, bank_T]; for item i of the batch: out[i] = bank[clip[i], :, :, start[i]:start[i]+T]
 * out is laid out branch-major for the twin launch: [2, n_items, 96, T]. clip/start are device int32. */
int zns_shift_gather_f32(const float* bank, int bank_T, const int32_t* clip, const int32_t* start,
                         int n_items, int T, float* out, void* stream);

/* ------------------------------------------------------------------ online clip selection (pretext.py:30-86)
 * zns_stem_mix_f32: anchor = stems[0] + stems[1] + ... in that order, float32 (pretext.py:50-61: every stem
 * except "drums", in dict order).  stems: host array of n_stems (<= 8) device pointers, n samples each.
 *
 * zns_rms_gate_f32: processing/stem_check.py:22-51 (check_CL_clips) with compute_rms (:8-19) =
 * librosa.feature.rms(frame_length=2048, hop_length=512, center=True, pad_mode="reflect").  Evaluates n_cand
 * candidate crops of n_samples each in one launch: candidate c starts at anchor + start[c] / positive + start[c]
 * (start: device int64[n_cand]; NULL = consecutive clips c * n_samples).  pass_count[c] (device int32) = number of
 * frames with  rms_a > rms_p / 2  and  rms_a < rms_p * 4;  the caller divides by zns_rms_frames(n_samples) and
 * applies lower_p < perc <= upper_p on the host, exactly as the reference does.  rms_a / rms_p: optional device
 * outputs [n_cand][n_frames]. */
int zns_stem_mix_f32(const float* const* stems, int n_stems, long long n, float* out, void* stream);
/* The same mix for MANY files in one launch, cropped (the bank fill of pretext.py:143-170): `table` is a device array of
 * n_files rows of 3 + max_other int64: [n_other (1..max_other), start, drums pointer, other-stem pointers in dict order];
 * out[f][0][0..n) = the float32 sum of file f's other stems over [start, start + n) (same order of additions as
 * zns_stem_mix_f32), out[f][1][0..n) = its drums stem over the same range.  out: [n_files, 2, n_samples].  The RMS gate
 * then reads row f at start offset f * 2 * n_samples (anchor = out, positive = out + n_samples). */
int zns_stage_crops_f32(const long long* table, int n_files, int max_other, int n_samples, float* out, void* stream);
int zns_rms_frames(int n_samples);
int zns_rms_gate_f32(const float* anchor, const float* positive, const long long* start, int n_cand, int n_samples,
                     float* rms_a, float* rms_p, int* pass_count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ZNS_H */
