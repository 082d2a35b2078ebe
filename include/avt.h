/*
 * avt.h — C ABI of the MI355X-native contrastive video-texture hot path.
 *
 * The reference (medhini/audio-video-textures) has no FFI: its boundary is the
 * Python operator `ContrastivePredictionTemporal.forward`
 * (contrastive_video_textures/models/models.py:307-467) driven by `validate()`
 * (contrastive_video_textures/validate.py:324-685).  This header is the boundary
 * the MI355X build puts underneath that operator; each entry point names the
 * reference lines it replaces.  The Python host side binds it with ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every function returns 0 on success, a negative avt_status otherwise;
 *     avt_last_error() returns a thread-local message for the last failure.
 *   - the CALLER owns every buffer.  Device pointers are row-major, contiguous
 *     unless a leading dimension is passed, 16-byte aligned.  Nothing is
 *     allocated, nothing is retained after the call returns.
 *   - `stream` is a hipStream_t passed as void* (0 = null stream).  Calls are
 *     asynchronous on that stream; the caller has already selected the device.
 *   - functions marked HOST touch no GPU state.
 */
#ifndef AVT_H
#define AVT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVT_ABI_VERSION 8  /* 8: avt_res2_x3, avt_weight_planes_gather_f32, avt_weight_planes_multi (round 6); 7: BatchNorm statistics on the producing convolution's epilogue: avt_conv3d_igemm_x3_f32_stats, avt_bn_train_fwd_pre (round 5); 2: avt_bn_train_fwd gained num_batches_tracked (round 2); 3: avt_bn_train_* take groups (+ beta, relu in bwd), avt_stem_conv_x3 takes frames_per_tile (round 3); 4: avt_stem_conv_pool_x3 removed, avt_stem_conv_x3 / avt_maxpool_hw3s2_ndhwc_x3 take a frame index, avt_stem_conv_x3_merged, avt_lateral_x3 (round 4); 5: avt_bn_train_fwd / _bwd and avt_maxpool_train_fwd / _bwd take the leading dimension of y / dy (round 4); 6: avt_pw_x3_f32 (round 4) */

typedef enum {
  AVT_OK = 0,
  AVT_ERR_ARG = -1,      /* bad argument (null pointer, size, alignment)   */
  AVT_ERR_UNSUPPORTED = -2,
  AVT_ERR_LAUNCH = -3,   /* hipGetLastError() after launch was not success */
  AVT_ERR_DEVICE = -4    /* not a gfx950 device / no device               */
} avt_status;

/* similarity precision (avt_sim_gemm_nt) */
#define AVT_SIM_BF16    0 /* 1 bf16 MFMA pass:  hi*hi                        */
#define AVT_SIM_BF16X3  1 /* 3 bf16 MFMA passes: hi*hi + hi*lo + lo*hi       */
#define AVT_SIM_F32     2 /* f32-input MFMA, exact fmaf chain (canonical)    */

/* clip_pack output element type */
#define AVT_DT_F32   0
#define AVT_DT_BF16  1

/* SlowFast-8x8 packing constants (reference: process_cv2_inputs call sites
 * models.py:365, validate.py:333; Appendix A of SURVEY.md) */
#define AVT_FAST_T 32
#define AVT_SLOW_T 8
#define AVT_SLOTS  (AVT_FAST_T + AVT_SLOW_T)

/* HOST. ABI version of the loaded library. */
int avt_abi_version(void);
/* HOST. Thread-local message of the last failing call on this thread. */
const char* avt_last_error(void);
/* Checks that the current HIP device is gfx950; fills `name` (may be NULL). */
int avt_device_check(char* name, size_t name_len);

/* ------------------------------------------------------------------------
 * clip_pack — replaces the per-window Python preprocessing of
 *   models.py:364-383 / validate.py:333-344 / dataset.py:145-154:
 *   frames (uint8 RGB, [F,H,W,3]) -> per window: /255, RGB->BGR, (x-mean)/std,
 *   temporal resample to 32 (fast) and 8 (slow) frames by
 *   linspace(0,win_len-1,32).long() and linspace(0,31,8).long(), per-plane
 *   bilinear resize to out_hw x out_hw (align_corners=False).
 *   slow: [n_win,3,8,out_hw,out_hw]   fast: [n_win,3,32,out_hw,out_hw]
 *
 * The kernel is organised by SOURCE frame: every (frame, channel) plane is
 * resized once and stored to every (window, slot) that samples it.  The plan
 * (a CSR list frame -> destination planes) is built on the host.
 * ---------------------------------------------------------------------- */

/* HOST. Temporal sample table: fast_idx[32], slow_idx[8] (indices into the
 * window), exactly torch.linspace(0,win_len-1,32).long() and
 * fast_idx[linspace(0,31,8).long()]. */
int avt_clip_sample_table(int win_len, int32_t* fast_idx, int32_t* slow_idx);

/* HOST. Builds the CSR plan.  dst_off has n_frames+1 entries; dst_slot holds
 * n_win*AVT_SLOTS entries, each = window*AVT_SLOTS + slot, where slot < 8 is
 * slow frame `slot` and slot >= 8 is fast frame `slot-8`.  Windows are
 * [win_start[i], win_start[i]+win_len).  Returns AVT_ERR_ARG if a window
 * leaves [0, n_frames). */
int avt_clip_pack_plan(const int32_t* win_start, int n_win, int win_len,
                       int n_frames, int32_t* dst_off, int32_t* dst_slot);

/* DEVICE. frames/dst_off/dst_slot/slow/fast are device pointers. */
int avt_clip_pack_u8(const uint8_t* frames, int n_frames, int height, int width,
                     const int32_t* dst_off, const int32_t* dst_slot, int n_win,
                     int out_hw, float mean, float std, int bgr,
                     void* slow, void* fast, int out_dtype, void* stream);

/* DEVICE. Channels-last variant feeding the MFMA stem: slow [n_win,8,hw,hw,4],
 * fast [n_win,32,hw,hw,4] bf16 (NDHWC with C padded 3 -> 4 by a zero channel). */
int avt_clip_pack_u8_ndhwc4(const uint8_t* frames, int n_frames, int height,
                            int width, const int32_t* dst_off,
                            const int32_t* dst_slot, int n_win, int out_hw,
                            float mean, float std, int bgr, void* slow,
                            void* fast, void* stream);

/* ------------------------------------------------------------------------
 * l2norm_rows — replaces torch.cat + F.normalize (models.py:347-351, 408-412,
 *   433-436): y = [x0|x1] / max(||[x0|x1]||_2, eps), row-wise.
 *   x1 may be NULL (d1 = 0).  Row norm is accumulated in fp64 and rounded to
 *   fp32 once (canonical; see oracle/avt_oracle.c).  Outputs (each optional):
 *   y_f32 [n, d0+d1] fp32; y_hi / y_lo [n, d0+d1] bf16 with
 *   hi = bf16_rne(y), lo = bf16_rne(y - hi).
 * ---------------------------------------------------------------------- */
int avt_l2norm_rows(const float* x0, int d0, const float* x1, int d1, int64_t n,
                    float eps, float* y_f32, void* y_hi, void* y_lo,
                    void* stream);

/* ------------------------------------------------------------------------
 * sim_gemm_nt — replaces torch.bmm(q, t) and `output /= temp`
 *   (models.py:416-417; audio branch :439, :455-457), for ALL queries at once:
 *   out[i,j] = <q_i, t_j> / temp,  i < nq, j < nt.
 *   precision AVT_SIM_F32   : q, t are fp32 [.,d]; q_lo/t_lo ignored.
 *       The dot product is the fp32 fmaf chain the f32-input MFMA executes:
 *       k visited in the order 0,4,1,5,2,6,3,7 inside each group of 8
 *       (d is zero-extended to a multiple of 8).  Bit-identical to
 *       oracle/avt_oracle.c: avt_oracle_sim_f32.
 *   precision AVT_SIM_BF16  : q, t are bf16 [.,d] (the hi outputs of l2norm).
 *   precision AVT_SIM_BF16X3: q,q_lo,t,t_lo are bf16 hi/lo pairs.
 *   ldo = leading dimension of out in elements (>= nt).
 * ---------------------------------------------------------------------- */
int avt_sim_gemm_nt(const void* q, const void* q_lo, const void* t,
                    const void* t_lo, int64_t nq, int64_t nt, int d, float temp,
                    int precision, float* out, int64_t ldo, void* stream);
/* out[m][n] = (sum_k A[m][k] * B[n][k]) / divisor with both operands as split-plane pairs (hi + lo, bf16 or fp16) and an fp32 result:
 * the contract-grade encoder's 256 x 256 LDS-DMA tile (csrc/conv_x3.hip: operands staged by `buffer_load ... lds` into two 64 KB LDS
 * stages, three MFMA passes per product, one correctly rounded division on the way out) as a plain NT GEMM.  Round 5: the bf16x3
 * similarity Q_hat T_hat^T / temp (models/models.py:416-417) runs on it when n % 256 == 0 and 256 <= k <= 8192 (k % 32 == 0) —
 * 0.35 -> 0.5 of the 833 TFLOP/s x3 roof; avt_sim_gemm_nt's own bf16x3 tile stays for every other shape.  ktab: the K-chunk table of
 * avt_conv3d_ktab(k, 1, 1, 1, 1, m, lda). */
int avt_gemm_nt_x3_f32out(const void* a_hi, const void* a_lo, int lda, const void* b_hi, const void* b_lo, float* out, int64_t ldo,
                          int m, int n, int k, float divisor, const int32_t* ktab, int plane_dtype, void* stream);

/* ------------------------------------------------------------------------
 * row_transition — replaces the CPU row post-process of validate.py:524-572.
 *   For each of the nq rows:
 *     p = x / sum(x);  [p = alpha*p + (1-alpha)*(xa/sum(xa))]
 *     ce = logsumexp(p) - p[0]                       (validate.py:531, report)
 *     cut = max(p) - threshold*max(p);  p[p < cut] = 0
 *     p[nz] /= sum(p);  entropy = |mean(log p[nz])|
 *     survivors = nonzero(p) in ascending position order
 *   Column order.  If q_ids == NULL the row is used as given (nt entries).
 *   Otherwise row r belongs to query segment q = q_ids[r] of n_seg segments
 *   and holds one score per TARGET SEGMENT id (nt == n_seg); positions are the
 *   reference's target order (validate.py:369-378): position 0 is
 *   pos = min(q+1, n_seg-1), then every id except q and pos ascending (row
 *   length n_seg-1, or n_seg when q == n_seg-1).  surv_idx holds POSITIONS;
 *   surv_seg (optional) holds the segment ids os_ids_t[position].
 *   sim_a (optional, same layout) is the driving-audio row (validate.py:525-527).
 *   cap = capacity of surv_* per row; surv_cnt[r] is the true count even if it
 *   exceeds cap.  stats (optional) [nq,4] = {row_sum, row_max, ce, entropy}.
 * ---------------------------------------------------------------------- */
int avt_row_transition(const float* sim, int64_t nq, int64_t nt, int64_t ld,
                       const int64_t* q_ids, int64_t n_seg, const float* sim_a,
                       int64_t ld_a, float alpha, float threshold, int cap,
                       int32_t* surv_idx, int32_t* surv_seg, float* surv_p,
                       int32_t* surv_cnt, float* stats, void* stream);

/* top-k select used by the sharded N=16384 configuration: per row the k
 * largest scores (ties: lower column first), descending.  self_col (optional)
 * [nq] is a column to exclude (the query itself). */
int avt_row_topk(const float* sim, int64_t nq, int64_t nt, int64_t ld,
                 const int64_t* self_col, int k, int32_t* top_idx,
                 float* top_val, void* stream);

/* ------------------------------------------------------------------------
 * softmax cross-entropy — replaces nn.CrossEntropyLoss on the InfoNCE logits
 *   (train.py:129-135; label 0 = positive).  fwd writes per-row loss and the
 *   softmax; bwd writes dlogits = scale * (prob - onehot(label)).
 * ---------------------------------------------------------------------- */
int avt_softmax_ce_fwd(const float* logits, int64_t b, int64_t c,
                       const int64_t* label, float* loss, float* prob,
                       void* stream);
int avt_softmax_ce_bwd(const float* prob, const int64_t* label, int64_t b,
                       int64_t c, float scale, float* dlogits, void* stream);

/* ------------------------------------------------------------------------
 * infonce_fwd / infonce_bwd — the operator's TRAINING branch fused: replaces F.normalize(q,1),
 *   F.normalize(t,2), torch.bmm and `/= temp` (models.py:351, 412-417 under :385-417) and their
 *   autograd backward.  q [b,d], t [b,n,d] fp32 (unnormalised embeddings, audio already
 *   concatenated by the caller), logits [b,n] = cos(q_b, t_bj)/temp with F.normalize's eps clamp.
 *   fwd also writes the inverse norms inv_q [b], inv_t [b,n] that bwd consumes;
 *   bwd: dlogits [b,n] -> dq [b,d], dt [b,n,d].
 * ---------------------------------------------------------------------- */
int avt_infonce_fwd(const float* q, const float* t, int64_t b, int n, int d,
                    float temp, float eps, float* logits, float* inv_q,
                    float* inv_t, void* stream);
int avt_infonce_bwd(const float* q, const float* t, const float* logits,
                    const float* dlogits, const float* inv_q, const float* inv_t,
                    int64_t b, int n, int d, float temp, float* dq, float* dt,
                    void* stream);

/* ------------------------------------------------------------------------
 * conv3d_igemm_bf16 — the encoder's convolutions on the matrix cores.  Replaces the
 *   Conv3d + BatchNorm3d + ReLU (+ residual add, + lateral-fusion concat) sequence the
 *   reference executes through the third-party SlowFast model for every clip window
 *   (models.py:335, :399; architecture: SURVEY.md Appendix A).
 *   Activations are NDHWC (channels-last-3d) bf16 rows of `ld*` elements; weights are
 *   BN-folded, packed [cout, kt*kh*kw*cin] bf16 with cin innermost; bias fp32 [cout] or NULL.
 *   out[m, 0:cout] = act(conv(in)[m] + bias (+ res[m, 0:cout])), act = ReLU when relu != 0.
 *   `out`/`res` may point INTO a wider row (channel slice of a concat buffer) via ldo/ldr.
 *   cin, cout, ld* multiples of 8; kernel extents 1..8.  ktab comes from avt_conv3d_ktab
 *   (HOST; 2*n_entries int32, n_entries = 8*ceil(kt*kh*kw*cin/64) + 2: the table plus 16 zero
 *   bytes the kernel fetches for out-of-bounds chunks), copied to the device.
 *   to/ho/wo: output extent; 0 = (x + 2p - k)/s + 1, a smaller value crops the far edge.
 * ---------------------------------------------------------------------- */
int avt_conv3d_ktab(int cin, int kt, int kh, int kw, int h, int w, int ldi,
                    int32_t* ktab, int n_entries);
int avt_conv3d_igemm_bf16(const void* in, const void* wt, const float* bias,
                          const void* res, void* out, const int32_t* ktab,
                          int batch, int t, int h, int w, int cin, int cout,
                          int kt, int kh, int kw, int st, int sh, int sw,
                          int pt, int ph, int pw, int to, int ho, int wo,
                          int ldi, int ldo, int ldr, int relu, void* stream);
/* The same convolution with its output rows REMAPPED: output position (frame f, ho, wo) is written to row
 * (f * out_h + out_row_stride * ho) * out_w + out_row_stride * wo of `out` (no residual).  This lets the stride-2
 * [1,3,3] conv of a slow-pathway stage's first bottleneck write behind the channels of its own INPUT's rows, so that
 * the block's c conv and its strided 1x1x1 shortcut conv run as ONE strided GEMM over K = [x | b-output]
 * (fused_slowfast._Block; same model, models.py:335, 399).  out_row_stride = 1 is avt_conv3d_igemm_bf16. */
int avt_conv3d_igemm_rows_bf16(const void* in, const void* wt, const float* bias,
                               const void* res, void* out, const int32_t* ktab,
                               int batch, int t, int h, int w, int cin, int cout,
                               int kt, int kh, int kw, int st, int sh, int sw,
                               int pt, int ph, int pw, int to, int ho, int wo,
                               int ldi, int ldo, int ldr, int relu,
                               int out_row_stride, int out_h, int out_w, void* stream);

/* The same convolution for the GEMM-like layers (Cout >= 256, K >= 1024, Cin % 32 == 0) with the weights ALSO given in
 * MFMA-fragment order along the kernel's K walk (taps innermost): wfrag [ceil(Cout/256)*8 tiles of 32 rows][nup units of
 * 32 K][2 k-slices][64 lanes][8] bf16 — lane = row (lane & 31), k-half (lane >> 5); unit i = tap i % taps of input-channel
 * chunk i / taps; element k = 16*slice + 8*half + e; zero beyond Cout / Cin / the last unit; nup >= 4*ceil(units/4) + 3
 * (fused_slowfast.pack_wfrag).  The weight operand then bypasses the LDS (csrc/conv_igemm.hip, XB tile).
 * wfrag = NULL is avt_conv3d_igemm_rows_bf16. */
int avt_conv3d_igemm_wfrag_supported(int cin, int cout, int kt, int kh, int kw);
int avt_conv3d_igemm_wfrag_bf16(const void* in, const void* wt, const float* bias,
                                const void* res, void* out, const int32_t* ktab,
                                int batch, int t, int h, int w, int cin, int cout,
                                int kt, int kh, int kw, int st, int sh, int sw,
                                int pt, int ph, int pw, int to, int ho, int wo,
                                int ldi, int ldo, int ldr, int relu,
                                int out_row_stride, int out_h, int out_w,
                                const void* wfrag, int nup, void* stream);

/* MaxPool3d((1,3,3), stride (1,2,2), pad (0,1,1)) of the SlowFast stems on NDHWC bf16 rows
 * (bt = batch*frames); out may be a channel slice of a wider row buffer (ldo).
 * tgroup > 1: each input row holds `tgroup` consecutive frames of c/tgroup channels (the
 * time-grouped fast stem); the output is un-grouped to bt*tgroup frames of c/tgroup channels. */
int avt_maxpool_hw3s2_ndhwc_bf16(const void* in, void* out, int bt, int h, int w,
                                 int c, int ldi, int ldo, int tgroup, void* stream);

/* Global average pool of the SlowFast head (AdaptiveAvgPool3d(1) per pathway, models/models.py:576-580 head
 * surgery): in [batch, p positions, c] bf16 rows (stride ldi) -> out[b, 0..c) fp32 (row stride ldo: a column
 * slice of the [batch, 2304] embedding table), mean in fp32, deterministic summation order. */
int avt_mean_positions_bf16(const void* in, int batch, int p, int c, int ldi,
                            float* out, int ldo, void* stream);

/* MaxPool2d(2, stride 2), floor mode, of VGGish (audio_models/vggish.py:15-33: the "M" entries of
 * its feature stack) on NHWC bf16 rows; bt = batch, out [bt, h/2, w/2, c] (row stride ldo). */
int avt_maxpool_hw2s2_ndhwc_bf16(const void* in, void* out, int bt, int h, int w,
                                 int c, int ldi, int ldo, void* stream);

/* SlowFast stem convolution in pixel-pair form with the input patch resident in LDS (csrc/stem_conv.hip):
 * the Conv3d(3, C, [kt,7,7], stride [1,2,2], pad [kt//2,3,3]) + BN + ReLU at the head of both pathways of
 * the third-party SlowFast model the reference runs per clip window (models/models.py:335, 399).
 * in  [batch, t, h, pw, 8] bf16 = the channels-last clip [.., w, 4] read as pixel pairs (pw = w/2),
 * wt  [cout/32, kt, 7, 2, 4, 16, 8] bf16: the packed stem weights W[n][dt][dh][dp][8] (BN folded; kt frame
 *     taps, temporal stride st, temporal pad pt) re-ordered per 32-channel group into the kernel's LDS
 *     image [dt][dh][tile][dp][row][8] with channel n = 32*group + 8*(row/4) + 4*tile + row%4, so that
 *     staging is a linear copy (cout % 32 == 0),
 * out [batch, (t+2pt-kt)/st+1, h/2, pw, cout] bf16.  Same results as avt_conv3d_igemm_bf16 on the same
 * weights up to fp32 summation order.  avt_stem_conv_supported() tells whether the shape is
 * covered (production 224^2 clips: pw = 112, (h/2) % 4 == 0); other shapes go to the generic entry. */
int avt_stem_conv_supported(int h, int pw, int cout);
int avt_stem_conv_bf16(const void* in, const void* wt, const float* bias, void* out,
                       int batch, int t, int h, int pw, int cout, int kt, int st, int pt,
                       int relu, void* stream);
/* The same convolution + ReLU with the stem's MaxPool3d((1,3,3),(1,2,2),(0,1,1)) fused: the convolution
 * output never reaches HBM.  out = pooled [batch, To*tgroup, h/4, pw/2, cout/tgroup] bf16 with row stride
 * ldo elements (a channel slice of a wider buffer is allowed); tgroup > 1 un-groups the time-grouped
 * fast stem exactly as avt_maxpool_hw3s2_ndhwc_bf16 does.  Results equal conv -> bf16 -> max-pool.
 * Needs (h/2) % 8 == 0 (a workgroup owns 8 conv rows = 4 pooled rows). */
int avt_stem_conv_pool_bf16(const void* in, const void* wt, const float* bias, void* out,
                            int batch, int t, int h, int pw, int cout, int kt, int st, int pt,
                            int tgroup, int ldo, void* stream);

/* One identity-shortcut bottleneck of the SlowFast FAST pathway in a single kernel (csrc/bottleneck_fused.hip):
 *   out = relu(c(relu(b(relu(a(x))))) + x),  a: Conv3d[3,1,1] C->Cm, b: Conv3d[1,3,3] Cm->Cm, c: Conv3d[1,1,1] Cm->C,
 * BatchNorms folded, stride 1 (blocks of the third-party SlowFast model the reference runs per clip window,
 * models/models.py:335, 399).  x, out [batch, t, h, w, c] bf16 (NDHWC, distinct buffers); the intermediates
 * stay in LDS.  Weights in MFMA fragment order, bf16, 1 KB per fragment (lane l: n = l & 15, k-group q = l >> 4,
 * 8 values each), the bottleneck width Cm zero-padded to CMP = 16 (Cm = 8, 16) or 32 (Cm = 32), NT = CMP/16:
 *   wa [3 dt][c/32][NT][64][8]: Wa[16*nt + n][dt][32*k + 8*q + e];   ba [CMP] fp32
 *   wb CMP 16: [5][1][64][8]: tap = 2*j + (q >> 1) (dh = tap/3, dw = tap%3; tap 9 = zeros), channel 8*(q & 1) + e
 *      CMP 32: [9][2][64][8]: tap = j, channel 8*q + e, row 16*nt + n;   bb [CMP]
 *   wc [c/16][64][8]: tile nt, row r = l & 15 -> output channel 32*(nt/2) + 8*(r/4) + 4*(nt%2) + r%4, k = 8*q + e;  bc [c]
 * tchunk = frames walked per workgroup (each re-reads one halo frame on either side).
 * avt_bottleneck_fused_supported(c, w): (32, 56), (64, 28), (128, 14) = the res2 / res3 / res4 blocks at 224^2 clips. */
int avt_bottleneck_fused_supported(int c, int w);
/* First block of the fast pathway's res2 stage (Cin = 8 -> C = 32, stride 1, 1x1x1 shortcut conv + BN instead of the
 * identity): out = relu(c(relu(b(relu(a(x))))) + shortcut(x)).  Same kernel; differences in the packing:
 *   wa [1][1][64][8]: k-group q = frame tap dt (q = 3: zeros), Wa[n][dt = q][e]  (all three taps in one MFMA k-step)
 *   wsc [c/16][64][8]: shortcut weights in c's row order, k-group 0 = the 8 input channels, others zero
 *   bc = c's bias + the shortcut's bias.  x [batch, t, h, w, 8] bf16.
 * Strided first blocks (res3: Cin 32 -> C 64 at w = 56; res4: 64 -> 128 at w = 28): b and the shortcut have spatial
 * stride 2, out [batch, t, h/2, w/2, c]; wa as for the identity form over Cin, wsc [c/16][Cin/32][64][8] with
 * k = 32*ks + 8*q + e over the input channels (h, w = INPUT extent, h even). */
int avt_bottleneck_first_supported(int cin, int c, int w);
int avt_bottleneck_first_bf16(const void* x, void* out, const void* wa, const float* ba,
                              const void* wb, const float* bb, const void* wc, const void* wsc,
                              const float* bc, int batch, int t, int h, int w, int cin, int c,
                              int tchunk, void* stream);
int avt_bottleneck_fused_bf16(const void* x, void* out, const void* wa, const float* ba,
                              const void* wb, const float* bb, const void* wc, const float* bc,
                              int batch, int t, int h, int w, int c, int tchunk, void* stream);

/* ---- contract-grade encoder: split-plane ("x3") kernels ------------------------------------------------------------
 * The reference's encoders compute in fp32 (models/models.py:335, 399) and north_star asks for scores within 1e-3 of
 * them ON THE SAME FRAMES; bf16 activations are ~100x away from that (profiles/r02/precision_*.json).  In this mode every
 * activation / weight tensor is a PAIR of 16-bit planes of identical geometry, x = hi + lo:
 *   AVT_X3_BF16: bf16 planes, |x - hi - lo| <= 2^-16 |x| at every magnitude
 *   AVT_X3_F16 : fp16 planes, <= 2^-22 |x| for |x| >= 2^-3 (absolute 2^-24 below); values clamped to +-65504; weights
 *                are stored pre-scaled by a power of two per output channel (wscale[n] undoes it on the accumulator)
 * and a product is three MFMA passes into one fp32 accumulator: wl*ah + wh*al + wh*ah (csrc/conv_x3.hip).
 * Arguments as avt_conv3d_igemm_rows_bf16, each tensor given as its two planes; res_hi/res_lo both NULL = no residual;
 * wscale [cout] fp32 or NULL.  * relu: 0 none, 1 ReLU, 2 LeakyReLU(0.1) (the SuperSloMo UNets, models/slowmo.py:69-71; this entry only). */
#define AVT_X3_BF16 0
#define AVT_X3_F16  1
int avt_conv3d_igemm_x3(const void* in_hi, const void* in_lo, const void* wt_hi, const void* wt_lo,
                        const float* bias, const void* res_hi, const void* res_lo, void* out_hi, void* out_lo,
                        const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout,
                        int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw,
                        int to, int ho, int wo, int ldi, int ldo, int ldr, int relu,
                        int out_row_stride, int out_h, int out_w, int plane_dtype, const float* wscale,
                        void* stream);
/* HOST. 1 when avt_conv3d_igemm_x3 runs this layer (cout output channels, k = taps * cin, m output positions) on its XL
 * tile (256 x 256 outputs per workgroup, half the L2 -> LDS operand bytes per flop of the 128 x 128 tile): cout % 256 == 0,
 * k >= 256, m >= 16384.  Same arithmetic, same results to the last bit as the other tiles is NOT promised (fp32 accumulation
 * order over K differs in the 32-wide steps); both are held to the fp32 reference by the same tolerance. */
int avt_conv3d_igemm_x3_xl_picked(int cout, int k, int m);
/* avt_conv3d_igemm_x3 for a layer its XL tile runs (avt_conv3d_igemm_x3_xl_picked(cout, k, m) == 1 and k % 32 == 0; an error
 * otherwise), with the weight planes in K-BLOCKED order wt[k / 32][cout][32] (element (n, kk) of the [cout, K] matrix at
 * ((kk / 32) * cout + n) * 32 + kk % 32): the 16 weight rows a wave stages per instruction are then 1 KB of consecutive bytes
 * (8 whole cache lines) instead of 16 half lines.  Same arithmetic and results as avt_conv3d_igemm_x3. */
int avt_conv3d_igemm_x3_wblk(const void* in_hi, const void* in_lo, const void* wt_hi, const void* wt_lo,
                             const float* bias, const void* res_hi, const void* res_lo, void* out_hi, void* out_lo,
                             const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt, int kh,
                             int kw, int st, int sh, int sw, int pt, int ph, int pw, int to, int ho, int wo, int ldi,
                             int ldo, int ldr, int relu, int out_row_stride, int out_h, int out_w, int plane_dtype,
                             const float* wscale, void* stream);
/* The [1,3,3] 64 -> 64 stride-1 convolution (+ BN + ReLU) of the slow pathway's res2 bottlenecks on plane pairs
 * (csrc/conv33_x3.hip; the contract-grade form of avt_conv33_c64_bf16): activations are MFMA operands as loaded from global
 * memory (no LDS staging, zero padding by out-of-range buffer offsets), the weights live in LDS as fragments.
 * x [batch*t*h*w, ldi], out [., ldo] plane pairs (channel slices of wider rows allowed).  wfrag [9 taps][4 k-slices of 16]
 * [2 n-tiles][2 planes: hi, lo][64 lanes][8] 16-bit: lane l of a fragment holds W[channel(l & 31)][tap][16 k + 8 (l >> 5) + e]
 * with channel(rho) = 32 n + (2 (r >> 3) + h) * 8 + (r & 7), h = (rho >> 2) & 1, r = (rho & 3) + 4 (rho >> 3) (so that a lane of
 * the 32 x 32 accumulator ends with runs of 8 consecutive channels); coef fp32 [scale 64 | bias 64] by output channel (scale =
 * the power of two that undoes the fp16 planes' weight scaling, 1 for bf16 planes). */
int avt_conv33_x3_supported(int cin, int cout);
int avt_conv33_x3(const void* x_hi, const void* x_lo, const void* wfrag, const float* coef, void* out_hi, void* out_lo,
                  int batch, int t, int h, int w, int ldi, int ldo, int relu, int plane_dtype, void* stream);

/* One bottleneck of the SlowFast FAST pathway in ONE kernel on plane pairs (csrc/bneck_x3.hip; the contract-grade form of
 * avt_bottleneck_fused_bf16 / avt_bottleneck_first_bf16 above — same blocks of the third-party SlowFast model,
 * models/models.py:335, 399):  out = relu(c(relu(b(relu(a(x))))) + x)  for cin == c (identity shortcut), or
 * out = relu(c(..) + shortcut(x)) for cin == 8, c == 32 (res2's first block, 1x1x1 shortcut conv).
 * x [batch, t, h, w, cin], out [batch, t, h, w, c] as hi / lo planes (NDHWC, distinct buffers); the a and b outputs stay in
 * LDS, the three frame taps of a and c's residual live in a register ring (x crosses HBM once + a 2-row halo per strip).
 * wfrag [NF][2 planes: hi, lo][64 lanes][8] 16-bit: the fragments of a ([3 dt][cin/32][NT]; first block: [NT], k-group q =
 * frame tap), b ([5][1] tap pairs for CMP 16, [9][2] for CMP 32), c ([c/16], rows permuted as for
 * avt_bottleneck_fused_bf16) and, for the first block, the shortcut ([c/16], weights at k-group 1 = frame t of the operand),
 * in that order, each fragment's two planes adjacent.  coef fp32 [sa CMP | ba CMP | sb CMP | bb CMP | sc c | bc c]: the
 * per-channel power-of-two factors that undo the fp16 planes' weight scaling (1 for bf16 planes) and the biases (bc includes
 * the shortcut's).  tchunk = frames walked per workgroup.  Supported: (cin, c, w) = (32, 32, 56), (64, 64, 28),
 * (128, 128, 14), (8, 32, 56) and small test shapes; anything else: the per-layer kernels. */
int avt_bneck_x3_supported(int cin, int c, int w);
int avt_bneck_x3(const void* x_hi, const void* x_lo, void* out_hi, void* out_lo, const void* wfrag, const float* coef,
                 int batch, int t, int h, int w, int cin, int c, int tchunk, int plane_dtype, void* stream);

/* One IDENTITY bottleneck of the SlowFast SLOW pathway's res2 stage in ONE kernel on plane pairs (csrc/res2_x3.hip, round 6; the
 * blocks s2.pathway0_res1 / _res2 of the third-party SlowFast model the reference runs per clip window, models/models.py:335, 399 —
 * round 5 ran them as three launches avt_pw_x3 / avt_conv33_x3 / avt_pw_chain_x3):
 *     out = relu(c(relu(b(relu(a(x))))) + x),  a: 1x1x1 256 -> 64, b: [1,3,3] 64 -> 64, c: 1x1x1 64 -> 256, BatchNorms folded.
 * x [batch*t*h*w, ldi], out [., ldo] plane pairs (channel slices of wider rows allowed; distinct buffers); fp16 planes only.
 * The a output lives in an LDS ring, b's output in registers, and the x tile stays in registers as the residual: x crosses HBM once,
 * out once; the weights STREAM from L2 by LDS-DMA.  wfrag = avt_res2_x3_wfrag_bytes() bytes: [17 chunks][8 pairs][2 planes: hi, lo]
 * [64 lanes][8] 16-bit, a pair = one 16-row x 32-k MFMA operand in avt_pw_x3's fragment form: lane l holds W[channel(nt, l & 15)]
 * [32 k + 8 (l >> 4) + e] with channel(nt, r) = 32 (nt / 2) + 8 (r / 4) + 4 (nt % 2) + r % 4 (two n-tiles give a lane of the
 * accumulators 8 consecutive channels).  Chunks 0-3: a, pair (kk, nt) = k-step 2 chunk + kk, n-tile nt, at index 4 kk + nt; chunks
 * 4-12: b, one tap each, pair (kk, nt) at 4 kk + nt; chunks 13-16: c, pair (nn, kk) = n-tile 4 (chunk - 13) + nn, k-step kk, at
 * 2 nn + kk.  coef fp32 [sa 64 | ba 64 | sb 64 | bb 64 | sc 256 | bc 256]: the power-of-two factors that undo the fp16 planes'
 * per-channel weight scaling, and the biases.  Supported: w = 56 (and 12: tests), any h / t / batch within 32-bit byte offsets. */
int avt_res2_x3_supported(int c, int cm, int w);
int avt_res2_x3_wfrag_bytes(void);
int avt_res2_x3(const void* x_hi, const void* x_lo, void* out_hi, void* out_lo, const void* wfrag, const float* coef, int batch,
                int t, int h, int w, int ldi, int ldo, int plane_dtype, void* stream);

/* Pointwise (1x1x1, stride 1) layers in the same arithmetic, streaming form (csrc/pw_x3.hip): a wave owns 16 rows from
 * load to store, weights are LDS-resident MFMA fragments of persistent workgroups.  y = act(W x + b [+ res]) on plane pairs;
 * x [m, ldx] (k valid channels), y / res [m, ldy / ldr] (n channels); w_hi / w_lo = fragments [n/16][ceil(k/32)][64 lanes][8]
 * with output rows permuted so a lane ends with 8 consecutive channels: tile nt, row r -> channel
 * 32*(nt/2) + 8*(r/4) + 4*(nt%2) + r%4, k = 32*ks + 8*(lane>>4) + e, zero beyond k (fused_slowfast.pack_pw_planes).
 * avt_pw_x3_supported(k, n): k in 32-steps {1,2,3,4,8,10,16}, n % 32 == 0, at most 8 channel chunks. */
int avt_pw_x3_supported(int k, int n);
int avt_pw_x3(const void* x_hi, const void* x_lo, int ldx, int k, const void* w_hi, const void* w_lo,
              const float* bias, const float* wscale, const void* res_hi, const void* res_lo, int ldr,
              void* y_hi, void* y_lo, int ldy, int n, int64_t m, int relu, int plane_dtype, void* stream);
/* The TRAINING form of avt_pw_x3 (round 4, ABI 6): fp32 rows in, fp32 rows out — the pointwise layers of train_ops.conv3d (forward and
 * stride-1 input gradient of the 1x1x1 convolutions of the SlowFast bottlenecks in train(), contrastive_video_textures/train.py:114-141),
 * which the 128 x 128 IO32 tile served at 1.1 TB/s when K is one to four K-steps (a workgroup's prologue + epilogue: 55 % of its cycles).
 * y [m, ldy] = (W x) * wscale [+ add]; x [m, ldx] fp32 (k valid channels), add [m, lda] fp32 or NULL; w_hi / w_lo = the PLAIN planes
 * [n][k] of avt_weight_planes_f32 / avt_weight_planes_t_f32 (no host packing: the weights change every optimizer step; the kernel's
 * prologue lays them out as fragments); wscale [n] or NULL.  avt_pw_x3_f32_supported(k, n): k % 8 == 0, ceil(k / 32) in {1, 2, 4, 8},
 * n % 32 == 0, at most 8 channel chunks. */
int avt_pw_x3_f32_supported(int k, int n);
int avt_pw_x3_f32(const float* x, int ldx, int k, const void* w_hi, const void* w_lo, const float* wscale, const float* add, int lda,
                  float* y, int ldy, int n, int64_t m, int plane_dtype, void* stream);
/* Two pointwise layers of consecutive slow-pathway bottlenecks in ONE pass over the rows, on plane pairs (csrc/pw_x3.hip; the
 * contract-grade form of avt_pw_chain_bf16):  y = act(W1 x + b1 [+ res]) (block i's expanding conv + residual + ReLU) and
 * z = relu(W2 y + b2) (block i + 1's reducing conv).  The first GEMM's result, already split into the planes it stores, is the
 * second GEMM's operand in registers: y is written once and never read back.  z is bit-identical to avt_pw_x3 applied to the
 * stored y.  w1 / w2 as for avt_pw_x3 (fragments over K1 and over K2 = n1).  Supported: (k1, n1, n2) = (64, 256, 64). */
int avt_pw_chain_x3_supported(int k1, int n1, int n2);
int avt_pw_chain_x3(const void* x_hi, const void* x_lo, int ldx, int k1, const void* w1_hi, const void* w1_lo, const float* bias1,
                    const float* wscale1, const void* res_hi, const void* res_lo, int ldr, void* y_hi, void* y_lo, int ldy,
                    int n1, int relu1, const void* w2_hi, const void* w2_lo, const float* bias2, const float* wscale2,
                    void* z_hi, void* z_lo, int ldz, int n2, int64_t m, int plane_dtype, void* stream);
/* Conv3d [kt,1,1] with temporal stride st and padding pt (+ bias, ReLU) on plane pairs, as a STREAMING pass (round 4): the
 * lateral fast -> slow connections of SlowFast (FuseFastToSlow: [7,1,1], stride 4; third-party model the reference runs per clip
 * window, models/models.py:335, 399).  x_* rows [batch * t * hw, ldx] (NDHWC), y_* rows [batch * to * hw, ldy], to = (t + 2 pt -
 * kt) / st + 1; w_* = fused_slowfast.pack_pw_planes of the convolution's own weight rows [cout rounded up to 32, kt * cin]
 * (tap-major K), bias / wscale [that many] or NULL.  avt_pw_x3's kernel with a gathered operand: chunk ch of output row
 * ((b, to), pos) is channels 8 (ch % (cin / 8)) .. + 7 of input frame to * st - pt + ch / (cin / 8) at pos; frames outside [0, t)
 * are zero operands.  Supported: kt * cin = 56 (8 channels), 224 (32), 448 (64); cout % 16 == 0. */
int avt_lateral_x3_supported(int cin, int cout, int kt);
int avt_lateral_x3(const void* x_hi, const void* x_lo, int ldx, int cin, const void* w_hi, const void* w_lo, const float* bias,
                   const float* wscale, void* y_hi, void* y_lo, int ldy, int cout, int batch, int t, int hw, int kt, int st,
                   int pt, int relu, int plane_dtype, void* stream);
/* avt_stem_conv_bf16 in the same arithmetic (csrc/stem_conv.hip, patch-resident): in / wt / out as plane pairs, wt_* in the
 * LDS image order of avt_stem_conv_bf16 (fused_slowfast.stem_lds_image of each plane), wscale [cout] or NULL. */
/* frames_per_tile (round 3, ABI 3): 0 = that image; 2 = its FRAME-MAJOR form for the time-grouped fast stem (st = 4 output
 * frames x 8 channels = cout 32 over kt = 5 + 3 frame taps): channel 16*tile + row, i.e. a 16-channel MFMA tile holds two output
 * frames and meets frame taps 2*tile .. 2*tile + 5 only — the other (tap, tile) pairs are structural zeros of the block-Toeplitz
 * weights and are skipped (4 of 16).  Same results. */
/* frame_idx (round 4, ABI 4): NULL = `in_*` is the dense clip tensor [batch, t, h, pw, 8].  Otherwise `in_*` is a TABLE of
 * n_table_frames distinct frames [n_table_frames, h, pw, 8] and input frame t of clip b is table frame frame_idx[b * t_total + t]
 * (int32, device): overlapping clip windows (W / S of their frames each; reference validate.py:188-195) and the fast pathway's
 * repeated frames (linspace(0, W-1, 32).long() with W < 32) are packed once per distinct frame instead of once per
 * (window, slot) — 0.55 GB instead of 5.3 GB per 166 windows at W = 20, S = 4.  Same results. */
int avt_stem_conv_x3(const void* in_hi, const void* in_lo, const void* wt_hi, const void* wt_lo, const float* bias,
                     const float* wscale, void* out_hi, void* out_lo, int batch, int t, int h, int pw, int cout,
                     int kt, int st, int pt, int relu, int plane_dtype, int frames_per_tile, const int32_t* frame_idx,
                     int n_table_frames, void* stream);
/* The time-grouped fast stem (st = 4 output frames x 8 channels, frame-major tiles) over a frame TABLE with the taps of one
 * source frame MERGED (round 4).  The fast pathway samples its 32 slots with linspace(0, W-1, 32).long() (reference:
 * models/models.py:365 via process_cv2_inputs): for W = 20 the 8 slots an output-frame group meets are only ~5 distinct frames.
 * A convolution is linear in its weights, so the caller sums the taps that read one source frame (fused_slowfast.
 * merged_stem_taps) and packs, per group `to` (To of them) and merged tap j < ktm, one weight slab: wt_* is the frame-major LDS
 * image over To * ktm slabs.  tap_frames int32 [batch * To * ktm] (device): table frame of (clip, group, tap), -1 ends a group's
 * list; tap_tiles int32 [To * ktm]: bit n set = tile n (output frames 2n, 2n + 1) has non-zero weights in that slab.  t, kt, st,
 * pt describe the unmerged geometry (To = (t + 2 pt - kt) / st + 1).  Same function as avt_stem_conv_x3 up to the rounding of
 * the summed weights (2^-22 relative per product). */
int avt_stem_conv_x3_merged(const void* in_hi, const void* in_lo, const void* wt_hi, const void* wt_lo, const float* bias,
                            const float* wscale, void* out_hi, void* out_lo, int batch, int t, int h, int pw, int cout,
                            int kt, int st, int pt, int relu, int plane_dtype, const int32_t* tap_frames,
                            const int32_t* tap_tiles, int ktm, int n_table_frames, void* stream);
/* avt_clip_pack_u8_ndhwc4 writing (hi, lo) planes: slow_* [n,8,hw,hw,4], fast_* [n,32,hw,hw,4]. */
int avt_clip_pack_u8_ndhwc4_x3(const uint8_t* frames, int n_frames, int height, int width,
                               const int32_t* dst_off, const int32_t* dst_slot, int n_win, int out_hw,
                               float mean, float std, int bgr, void* slow_hi, void* slow_lo,
                               void* fast_hi, void* fast_lo, int plane_dtype, void* stream);
/* MaxPool2d(2, 2), floor mode, on plane pairs (NHWC rows): the contract-grade VGGish (audio_models/vggish.py:15-33; the
 * plane-pair form of avt_maxpool_hw2s2_ndhwc_bf16).  The max is taken on hi + lo and split again: value-preserving. */
int avt_maxpool_hw2s2_ndhwc_x3(const void* in_hi, const void* in_lo, void* out_hi, void* out_lo, int bt, int h, int w, int c,
                               int ldi, int ldo, int plane_dtype, void* stream);
/* avt_maxpool_hw3s2_ndhwc_bf16 / avt_mean_positions_bf16 on plane pairs (max / sum of the fp32 values hi + lo).
 * frame_idx (round 4, ABI 4; tgroup == 1): NULL, or int32 [bt] on the device: output frame b pools INPUT frame frame_idx[b] —
 * the slow stem ([1,7,7]: no temporal taps) runs once per distinct source frame and the pool hands every (window, slot) its frame.
 * The entries are NOT range-checked on the device (plain global loads): the caller guarantees 0 <= frame_idx[b] < input frames
 * (ops.clip_pack_frames validates the window starts they are built from). */
int avt_maxpool_hw3s2_ndhwc_x3(const void* in_hi, const void* in_lo, void* out_hi, void* out_lo, int bt, int h,
                               int w, int c, int ldi, int ldo, int tgroup, int plane_dtype, const int32_t* frame_idx,
                               void* stream);
int avt_mean_positions_x3(const void* in_hi, const void* in_lo, int batch, int p, int c, int ldi,
                          float* out, int ldo, int plane_dtype, void* stream);

/* ---- training input path on the device (config 5; dataset/dataset.py:121-253) ----------------------------------------
 * avt_negative_sample_mt19937: the dataset's negative sampling (dataset.py:128-139, 181-190) from a DEVICE-resident
 * MT19937 state, stream-for-stream what NumPy's legacy np.random.choice(others, n_negs, replace=False) draws, then the
 * hard negatives idx-4..idx-1, idx+2..idx+5 (clipped to [0, len]) overwrite the head.  mt_state [625] uint32 = the 624
 * key words + position of np.random.get_state(), advanced in place; idx [batch] int64 query segment ids; n_len =
 * len(dataset) (train split); neg_out [batch, n_negs] int32.  Items are drawn in batch order.
 * avt_clip_pack_gather_u8: avt_clip_pack_u8 organised by destination, window starts read from a DEVICE array (no host
 * plan): slow [n_win,3,8,hw,hw], fast [n_win,3,32,hw,hw] in out_dtype. */
int avt_negative_sample_mt19937(uint32_t* mt_state, const int64_t* idx, int batch, int n_len, int n_negs,
                                int32_t* neg_out, void* stream);
int avt_clip_pack_gather_u8(const uint8_t* frames, int n_frames, int height, int width,
                            const int32_t* win_start, int n_win, int win_len, int out_hw, float mean,
                            float std, int bgr, void* slow, void* fast, int out_dtype, void* stream);

/* Train-mode BatchNorm3d fused with the shortcut add and the ReLU that follow it in the SlowFast blocks (csrc/bn_train.hip;
 * the model the reference trains, train.py:114-141 / models/models.py:385-417), on channels-last fp32 rows [m, c] (c a power of
 * two >= 8):   y = act((x - mean_c) * invstd_c * gamma_c + beta_c [+ res]),  batch statistics in fp64.
 * Statistics are summed in fp64 in a fixed order (no atomics): results are bitwise reproducible.
 * fwd: writes save_mean / save_invstd [c]; running_mean / running_var (both or neither) are updated with
 *      momentum and the unbiased batch variance as torch.nn.BatchNorm3d does.
 * bwd: relu = whether the forward applied one; its mask is the sign of y (the forward output) or, with y NULL — allowed when
 *      the forward had no shortcut — recomputed from x with the forward's own expression (needs beta; a third fewer bytes);
 *      dres (may be NULL) receives the shortcut's gradient; dgamma / dbeta [c]. */
int64_t avt_bn_train_ws_bytes(int64_t m, int c, int groups); /* workspace both calls need (16-byte aligned); -1 outside the domain */
/* groups (round 3, ABI 3): the m rows are `groups` equal consecutive slabs, each normalised with its OWN batch statistics — the
 * items of a batch as DataParallel replicas see them (main.py:420) in one launch; save_mean / save_invstd are [groups, c], the
 * running statistics take the groups' updates in order, dgamma / dbeta sum over all groups.  1 = plain BatchNorm. */
/* relu_mask (round 3; may be NULL): m * c / 4 bytes, 4 bits per float4 chunk of the rows — the forward writes where its output
 * is positive, the backward reads it INSTEAD of y (with a shortcut the mask cannot be recomputed from x: a sixteenth of y's bytes
 * in each of the backward's two passes). */
/* ldy / ld_dy (round 4, ABI 5; 0 = c): floats between consecutive rows of the forward's y / the backward's dy — a channel slice
 * of a wider row.  SlowFast's lateral fusion concatenates the slow pathway with the fast pathway's lateral convolution
 * (torch.cat([slow, lateral], 1), the third-party model under train.py:114-141): with the two producers writing their slices of
 * ONE buffer and their backward passes reading slices of its gradient, neither the concatenation nor the gradient's
 * .contiguous() copies exist (13 GB of strided copies per config-5 step).  Everything else stays contiguous [m, c]. */
int avt_bn_train_fwd(const float* x, const float* res, float* y, int64_t m, int c, const float* gamma,
                     const float* beta, float eps, float momentum, int relu, int groups, void* ws, int64_t ws_size,
                     float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked /* incremented (by groups) when not NULL */, void* relu_mask, int64_t ldy,
                     void* stream);
int avt_bn_train_bwd(const float* dy, const float* y, const float* x, int64_t m, int c, const float* gamma,
                     const float* beta, const float* save_mean, const float* save_invstd, int relu, int groups,
                     const void* relu_mask, void* ws, int64_t ws_size, float* dx, float* dres, float* dgamma, float* dbeta,
                     int64_t ld_dy, void* stream);

/* The training form of avt_conv3d_igemm_x3 (csrc/conv_x3.hip, IO32): fp32 NDHWC rows in [batch*t*h*w, ldi], fp32 rows out
 * [M, ldo], no bias / residual / activation — an fp32-grade Conv3d(bias=False) on channels-last tensors for the forward
 * and the stride-1 input gradient of the SlowFast convolutions in train() (contrastive_video_textures/train.py:114-141;
 * avtex/train_ops.py).  Activations are split into the two planes inside the kernel; wt_hi / wt_lo [cout, kt*kh*kw*cin]
 * are the weight planes (plane_dtype AVT_X3_*; wscale as in avt_conv3d_igemm_x3, or NULL).  add (may be NULL): fp32 rows
 * [M, lda] summed into the result (out = conv + add) — a gradient reaching the same tensor by another path. */
int avt_conv3d_igemm_x3_f32(const float* in, const void* wt_hi, const void* wt_lo, const float* wscale, const float* add,
                            float* out, const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt,
                            int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int ldi, int ldo, int lda,
                            int plane_dtype, void* stream);
/* BatchNorm statistics on the producing convolution's epilogue (round 5).  In train() every convolution of the
 * SlowFast blocks is followed by a train-mode BatchNorm (train.py:114-141; per-replica statistics, main.py:420), whose statistics
 * pass re-reads the whole activation the convolution has just written.  avt_conv3d_igemm_x3_f32_stats = avt_conv3d_igemm_x3_f32
 * (no add operand) whose epilogue also leaves, per M tile, the per-channel sum and sum of squares of the rows it stores: the m rows
 * are `groups` equal slabs with statistics of their own, the M tiles are laid per group (the last tile of a group is short: no tile
 * straddles two groups), and tile t of group g writes row (g * rows + t) of `stat_part` — doubles, in the layout the BatchNorm's
 * finalize kernel sums in a fixed order (bitwise reproducible; per-thread fp32 sums over at most 16 rows, everything above in fp64).
 * stat_c = the BatchNorm's channel count: cout, or cout / g for a pixel-grouped layer (columns n and n + stat_c are one channel).
 * avt_conv3d_igemm_x3_f32_stat_rows -> rows of partials per group that call writes (tile height 64, 128 or 256 by the layer's shape and batch).
 * avt_bn_train_fwd_pre = avt_bn_train_fwd without its statistics pass: `ws` holds pre_rows rows per group written by the producer
 * (pre_rows = stat_rows * max(1, c / 1024)), sized by avt_bn_train_ws_bytes_pre(c, groups, pre_rows). */
/* The IO32 tiles are 256 rows (the XL tile's layers at >= 65 536 rows), 128 rows, or — round 6 — 64 rows where a wide layer's 128-row
 * tiles would number fewer than two per CU (config 5 at one item per rank).  avt_conv_x3_set_small_tile(0) switches the 64-row form off
 * (A/Bs, tests of the 128-row form at small sizes); returns the previous setting.  The *_stat_rows / *_bwdstats_rows queries follow it. */
int avt_conv_x3_set_small_tile(int on);
int avt_conv3d_igemm_x3_f32_stat_rows(int cout, int k, int64_t m, int groups);
int avt_conv3d_igemm_x3_f32_stats(const float* in, const void* wt_hi, const void* wt_lo, const float* wscale, float* out,
                                  const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt, int kh, int kw,
                                  int st, int sh, int sw, int pt, int ph, int pw, int ldi, int ldo, int plane_dtype,
                                  void* stat_part, int groups, int stat_c, void* stream);
/* BatchNorm BACKWARD statistics on the epilogue of an input-gradient launch (round 5).  The stride-1 input
 * gradient of the convolution that CONSUMES a train-mode BatchNorm (+ ReLU)'s output computes that BatchNorm's output gradient dz;
 * the BatchNorm's backward then read dz and its own input x twice each (statistics pass, apply pass).  avt_conv3d_igemm_x3_f32_bwdstats
 * = avt_conv3d_igemm_x3_f32 (stride 1; `add` allowed: the shortcut's gradient, train_ops.conv3d_fork) whose epilogue reads the
 * BatchNorm's input rows bn_x [M, cout] (the geometry of the output), rebuilds the ReLU mask — bn_mask: the forward's 4 bits per
 * float4 chunk when the BatchNorm had a shortcut; else recomputed as (x - mean) * (invstd * gamma) + beta > 0, the forward's own
 * expression — stores g = mask * (dz + add) instead of dz, and leaves the per-tile sums of g and g * xhat in stat_part (layout,
 * groups and stat_c as in avt_conv3d_igemm_x3_f32_stats).  avt_bn_train_bwd_pre finishes the BatchNorm's backward from g and those
 * rows: finalize + apply, no statistics pass, and the shortcut's gradient IS g (nothing written for it).  bf16 planes; layers of the
 * 128-row tile (avt_conv3d_igemm_x3_f32_bwdstats_rows = rows of partials per group, -1 elsewhere). */
int avt_conv3d_igemm_x3_f32_bwdstats_rows(int cout, int k, int64_t m, int groups);
int avt_conv3d_igemm_x3_f32_bwdstats(const float* in, const void* wt_hi, const void* wt_lo, const float* wscale, const float* add,
                                     float* out, const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt,
                                     int kh, int kw, int pt, int ph, int pw, int ldi, int ldo, int lda, int plane_dtype,
                                     const float* bn_x, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                     const float* bn_beta, const void* bn_mask, int relu, void* stat_part, int groups, int stat_c,
                                     void* stream);
/* ... and its streaming pointwise counterpart (csrc/pw_x3.hip; rows of the partials as avt_pw_x3_f32_stats: one per wave and group;
 * output rows contiguous: ldy == n) */
int avt_pw_x3_f32_bwdstats_rows(int k, int n, int64_t m, int groups);
int avt_pw_x3_f32_bwdstats(const float* x, int ldx, int k, const void* w_hi, const void* w_lo, const float* add, int lda, float* y,
                           int ldy, int n, int64_t m, int plane_dtype, const float* bn_x, const float* bn_mean,
                           const float* bn_invstd, const float* bn_gamma, const float* bn_beta, const void* bn_mask, int relu,
                           void* stat_part, int groups, void* stream);
int avt_bn_train_bwd_pre(const float* g, const float* x, int64_t m, int c, const float* gamma, const float* save_mean,
                         const float* save_invstd, int groups, void* ws, int64_t ws_size, int pre_rows, float* dx, float* dgamma,
                         float* dbeta, void* stream);
/* The streaming pointwise counterpart (csrc/pw_x3.hip): avt_pw_x3_f32 without an add operand over `groups` equal slabs of the m rows
 * (blockIdx.y = group), every wave leaving ONE row of partials — the sums over all the 16-row tiles it stored, taken column-wise
 * through a per-wave LDS block (fp32 over the 16 rows of a tile, fp64 across tiles).  n must be a power of two (the BatchNorm's
 * domain); avt_pw_x3_f32_stat_rows -> rows of partials per group (8 x the launch's row groups), or -1. */
int avt_pw_x3_f32_stat_rows(int k, int n, int64_t m, int groups);
int avt_pw_x3_f32_stats(const float* x, int ldx, int k, const void* w_hi, const void* w_lo, const float* wscale, float* y, int ldy,
                        int n, int64_t m, int plane_dtype, void* stat_part, int groups, void* stream);
int64_t avt_bn_train_ws_bytes_pre(int c, int groups, int pre_rows);
int avt_bn_train_fwd_pre(const float* x, const float* res, float* y, int64_t m, int c, const float* gamma, const float* beta,
                         float eps, float momentum, int relu, int groups, void* ws, int64_t ws_size, float* save_mean,
                         float* save_invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                         void* relu_mask, int64_t ldy, int pre_rows, void* stream);
/* avt_conv3d_igemm_x3_f32 (stride 1) with an explicit output extent to x ho x wo (positions past the symmetric-padding formula's
 * far edge are not allowed, fewer are: padding after = whatever the extent needs) and the output-row remap of avt_conv3d_igemm_x3
 * (position (f, ho, wo) -> row (f * out_h + out_row_stride * ho) * out_w + out_row_stride * wo; out_h = out_w = 0: none).  One
 * residue class of a STRIDED convolution's input gradient is such a convolution of dY (train_ops._dgrad_strided: the strided
 * [1,3,3], 1x1x1 and [7,1,1] layers of SlowFast, train.py:139-141 — the last kernels MIOpen ran in the training step). */
int avt_conv3d_igemm_x3_f32_ex(const float* in, const void* wt_hi, const void* wt_lo, const float* wscale, float* out,
                               const int32_t* ktab, int batch, int t, int h, int w, int cin, int cout, int kt, int kh, int kw,
                               int pt, int ph, int pw, int to, int ho, int wo, int ldi, int ldo, int out_row_stride, int out_h,
                               int out_w, int plane_dtype, void* stream);

/* Weight gradient of the same convolution on the split-plane arithmetic (csrc/wgrad_x3.hip; bf16 planes, 2^-16 per
 * product): dw[cout][kt*kh*kw][cin] (the memory of a channels_last_3d Conv3d weight) = sum over output positions of
 * dy[m][co] * x[in(m, tap)][ci]; dy fp32 rows [batch*to*ho*wo, ldy], x fp32 rows [batch*t*h*w, ldx].  dw is zeroed by the
 * call and accumulated with fp32 atomics (summation order not deterministic).  Any stride / padding. */
int avt_conv3d_wgrad_x3_f32(const float* dy, const float* x, float* dw, int batch, int t, int h, int w, int cin, int cout,
                            int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int ldx, int ldy,
                            void* stream);
/* Tile choice of the weight gradient (round 5).  Two tiles: the phase-serial one (128 x {128, 64, 32} outputs, 64-position slabs, two
 * workgroups per CU) and the pipelined one (256 x {128, 64, 32} outputs, 8 waves; global loads two 32-position steps ahead, bf16 split +
 * LDS stage and MFMAs overlapped inside the workgroup; <= 28 taps).  Tensors below 4 GB are read with buffer loads (position tables of
 * byte offsets, hardware zero fill); larger ones by the phase-serial tile with plain loads.
 * avt_wgrad_x3_set_xl: 1 (default) = the pipelined tile where it measured faster (pointwise / temporal-tap layers with >= 512 x 65 or
 * >= 224 x 33 on the longer x shorter axis), 2 = on every layer at the width that fits (tests, probes), 0 = never; 3 / 4 = mode 1 with /
 * without its 64-wide form; 5 / 6 = the phase-serial tile's buffer-load form off / on (tile mode unchanged).  -> the previous mode. */
int avt_wgrad_x3_set_xl(int on);
/* The general form of avt_conv3d_wgrad_x3_f32: input channels in multiples of 4 (the SlowFast stems' 3 channels travel as 4:
 * ldx may be wider than cin), up to 49 taps ([1,7,7]), an explicit output extent (to, ho, wo; 0 = the symmetric-padding
 * formula) and any pt / ph / pw — so one frame-tap slice of a longer filter (the fast stem's [5,7,7] = five [1,7,7] slices,
 * pt = pad - dt) is a call of its own writing into its columns of dW (rows ldw elements apart; 0 = taps * cin); zero_dw: dW
 * is zeroed here (whole-filter calls) or was by the caller (slices).  Replaces the stems' MIOpen bwd_weight (train.py:139-141). */
int avt_conv3d_wgrad_x3_sub_f32(const float* dy, const float* x, float* dw, int batch, int t, int h, int w, int cin, int cout,
                                int kt, int kh, int kw, int st, int sh, int sw, int pt, int ph, int pw, int to, int ho, int wo,
                                int ldx, int ldy, int ldw, int zero_dw, void* stream);

/* The SlowFast stems in the TRAINING step, patch-resident like the inference stem (csrc/stem_conv.hip, csrc/stem_train.hip;
 * the reference trains them through autograd -> MIOpen, train.py:114-141):
 * avt_clip_planes_f32: the clip, a [batch, 3, t, h, w] fp32 view with element strides (sb, sc, st, sh, sw), -> the two
 *   16-bit planes [batch, t, h, w, 4] (4th channel 0; read as pixel pairs [.., w/2, 8] by the stem kernels).
 * avt_stem_conv_x3_f32: avt_stem_conv_x3 without bias / ReLU and with fp32 output [batch, to*tgroup, h/2, pw, cout/tgroup]
 *   (NDHWC rows; the time-grouped fast stem's channels go back to their frames): the forward of the training step.
 * avt_stem_wgrad_x3: the weight gradient of Conv3d(3, cout, [kt,7,7], stride [1,2,2], pad [pt,3,3]) in the pixel-pair form:
 *   dw[cout][kt][7][4 pair taps][8 = pixel-in-pair * 4 + channel] fp32, ZEROED BY THE CALLER, accumulated with fp32 atomics;
 *   pair tap dp, pixel p = column tap 2 dp + p - 1 (dp = p = 0 is the structural zero of the pair form: ignore it);
 *   x_* = bf16 planes of avt_clip_planes_f32, dy fp32 [batch, to, h/2, pw, cout].  Split-plane bf16 arithmetic, 2^-16 per
 *   product.  avt_stem_wgrad_x3_supported: (h/2) % 4 == 0, 8 <= pw <= 128, kt 1 or 5, cout 8 or a multiple of 16. */
int avt_clip_planes_f32(const float* in, int batch, int t, int h, int w, int64_t sb, int64_t sc, int64_t st, int64_t sh,
                        int64_t sw, void* out_hi, void* out_lo, int plane_dtype, void* stream);
int avt_stem_conv_x3_f32(const void* in_hi, const void* in_lo, const void* wt_hi, const void* wt_lo, const float* wscale,
                         float* out, int batch, int t, int h, int pw, int cout, int kt, int st, int pt, int tgroup,
                         int plane_dtype, int frames_per_tile /* as avt_stem_conv_x3 */, void* stream);
int avt_stem_wgrad_x3_supported(int h, int pw, int cout, int kt);
int avt_stem_wgrad_x3(const void* x_hi, const void* x_lo, const float* dy, float* dw, int batch, int t, int h, int pw,
                      int cout, int kt, int pt, void* stream);

/* The planes conv_x3 reads, from a training convolution's fp32 weight (re-made after every optimizer step; csrc/stem_train.hip).
 * avt_weight_planes_f32: w [cout][k] fp32 (a channels-last Conv3d weight: k = taps * cin) -> hi / lo [cout][k]; fp16 planes
 *   (plane_dtype 1) are scaled per row by a power of two into [2^9, 2^10) and wscale [cout] receives 1 / scale; bf16 planes
 *   (plane_dtype 0) are unscaled and wscale must be NULL.
 * avt_weight_planes_t_f32: the input gradient's filter: hi / lo [cin][nsel][cout] bf16 planes with
 *   out[ci][a][co] = w[co][sel[a]][ci] (sel = HOST array of nsel <= 32 source taps: all taps reversed for a stride-1 layer,
 *   one residue class's taps for a strided one, train_ops._dgrad_strided). */
int avt_weight_planes_f32(const float* w, int cout, int k, void* hi, void* lo, float* wscale, int plane_dtype, void* stream);
/* ... with the rows gathered through an index map (round 6): out row r, column k = w[map[r * k_cols + k]] (map < 0: zero) — the
 * pixel-grouped (block-Toeplitz) forms of the few-channel layers' weights, forward and input-gradient filters alike
 * (train_ops._grouped_planes: one launch where torch ran a flip, a cat, an index, a permute and a copy per weight and step). */
int avt_weight_planes_gather_f32(const float* w, const int32_t* map, int rows, int k, void* hi, void* lo, float* wscale,
                                 int plane_dtype, void* stream);
int avt_weight_planes_t_f32(const float* w, int cout, int taps, int cin, const int32_t* sel, int nsel, void* hi, void* lo,
                            void* stream);
/* ... and ALL of a step's planes in one launch (round 6): `jobs` is a DEVICE array of AvtPlaneJob — one per launch of the three entry
 * points above, with the arguments that launch would take — and `blk2job` a DEVICE int32 [nblocks] naming the job of every block
 * (job j owns blocks blk0 .. blk0 + its own grid size: rows for kinds 0 / 2, gx * gy * nsel for kind 1).  The blocks run the same device
 * code as the single launches: the planes are bit-identical.  The caller validates the jobs (it has made each of them once through the
 * single entry points); avt_weight_planes_job_bytes() = sizeof(AvtPlaneJob), for bindings that build the table as raw bytes. */
typedef struct AvtPlaneJob {
  const float* w;        /* the weight (rows / transposed), or the tensor `map` indexes (gathered rows) */
  void* hi;
  void* lo;
  float* wscale;         /* kinds 0 / 2 with fp16 planes: [rows] receives 1 / scale; else NULL */
  const int32_t* map;    /* kind 2: [rows][k] element offsets into w, < 0 = zero */
  int32_t kind;          /* 0: avt_weight_planes_f32, 1: avt_weight_planes_t_f32, 2: avt_weight_planes_gather_f32 */
  int32_t f16;           /* kinds 0 / 2: 1 = fp16 planes (row-scaled), 0 = bf16 planes */
  int32_t rows, k;       /* kinds 0 / 2 */
  int32_t cout, taps, cin, nsel; /* kind 1 */
  int32_t gx, gy;        /* kind 1: ceil(cin / 32), ceil(cout / 32) */
  int32_t blk0;          /* first block of the job in the launch */
  int32_t pad_;
  int32_t sel[32];       /* kind 1: source tap of output tap a */
} AvtPlaneJob;
int avt_weight_planes_job_bytes(void);
int avt_weight_planes_multi(const void* jobs, const int32_t* blk2job, int nblocks, void* stream);

/* MaxPool3d((1,3,3),(1,2,2),(0,1,1)) of the stems in the training step on fp32 NDHWC rows [bt, h, w, c] (csrc/stem_train.hip;
 * the reference: the third-party SlowFast stem under autograd, train.py:114-141).  fwd: y [bt, ho, wo, c] and `tap`
 * (bt*ho*wo*c/2 bytes: 4 bits per element = which of the 9 taps held the maximum; the first one on ties, a NaN wins — torch's
 * rule).  bwd: dx = the gradient routed to those taps, as a gather (no atomics, fixed order).  c % 4 == 0. */
/* ldy / ld_dy (round 4, ABI 5; 0 = c): as for avt_bn_train_*: y / dy as a channel slice of wider rows (the slow stem's pool feeds
 * the first lateral fusion). */
int avt_maxpool_train_fwd(const float* x, float* y, void* tap, int bt, int h, int w, int c, int64_t ldy, void* stream);
int avt_maxpool_train_bwd(const float* dy, const void* tap, float* dx, int bt, int h, int w, int c, int64_t ld_dy, void* stream);

/* SuperSloMo interpolation at the jumps of the stitched video (contrastive_video_textures/interpolate.py:75-147, called from
 * validate.py:588-611): the passes around the two UNets, whose convolutions are avt_conv3d_igemm_x3 with relu = 2
 * (csrc/interp.hip).  Plane pairs as above (plane_dtype AVT_X3_*), NHWC rows; `mean3` is a HOST array of 3 floats.
 * avt_interp_pack_pair_u8: frames [h, w, 3] uint8 RGB (device) -> img [2, h, w, 4] fp32 = x / 255 - mean (4th lane 0) and
 *   flowComp's input planes [h*w, 8] = (I0 rgb, I1 rgb, 0, 0).
 * avt_avgpool2_x3: F.avg_pool2d(x, 2): [batch, h, w, c] rows of stride ldi -> [batch, h/2, w/2, c] rows of stride ldo.
 * avt_upsample2_bilinear_x3: F.interpolate(x, scale_factor=2, mode="bilinear") (align_corners False):
 *   [batch, h, w, c] -> [batch, 2h, 2w, c]; ldo lets the result land in a channel slice of a concat buffer.
 * avt_interp_mid_input: flow planes [h*w, 8] (flowComp's output: F_0_1, F_1_0, 4 pad) -> for t = i / sf, i = 1..sf-1:
 *   ArbTimeFlowIntrp's input planes [(sf-1)*h*w, 24] (20 channels + 4 pad) and ft [(sf-1), h, w, 4] = (F_t_0, F_t_1).
 * avt_interp_final_u8: o planes [(sf-1)*h*w, 8] (ArbTimeFlowIntrp's 5 outputs + pad) -> out [(sf-1), h, w, 3] uint8,
 *   the frames ToPILImage would give (x * 255 truncated). */
int avt_interp_pack_pair_u8(const uint8_t* frame0, const uint8_t* frame1, int height, int width, const float* mean3,
                            float* img, void* x_hi, void* x_lo, int plane_dtype, void* stream);
int avt_avgpool2_x3(const void* in_hi, const void* in_lo, int batch, int h, int w, int c, int ldi, void* out_hi,
                    void* out_lo, int ldo, int plane_dtype, void* stream);
int avt_upsample2_bilinear_x3(const void* in_hi, const void* in_lo, int batch, int h, int w, int c, int ldi,
                              void* out_hi, void* out_lo, int ldo, int plane_dtype, void* stream);
int avt_interp_mid_input(const float* img, const void* flow_hi, const void* flow_lo, int height, int width, int sf,
                         void* x_hi, void* x_lo, float* ft, int plane_dtype, void* stream);
int avt_interp_final_u8(const float* img, const float* ft, const void* o_hi, const void* o_lo, int height, int width,
                        int sf, const float* mean3, uint8_t* out, int plane_dtype, void* stream);

/* D1[i, j] = || x_i - x_j ||_2 of the classic video-texture baseline (baselines/classic_video_textures/
 * computeD1.py:47-96; BASELINE config 1): x [n, d] fp32 device rows (flattened frames), out [n, n] fp32.
 * fp64 accumulation in a fixed order, one sqrt, one rounding. */
int avt_pairwise_l2_f32(const float* x, int n, int64_t d, float* out, void* stream);
/* The next two steps of the same baseline on device matrices (csrc/classic.hip):
 * avt_diag_filter_f32: D2[i, j] = sum_k w[k] * D1[i + k, j + k], out [(n-fs+1)^2] (computeD2.py:21-52: conv2d with a diagonal
 *   [fs, fs] kernel, valid padding);
 * avt_q_learning_f32: the future-cost iteration of q_learning.py:27-68 on d3 = D2^p [n, n] (n <= 200: one workgroup, matrix in
 *   LDS): sweeps until mean((new - old)^2) <= tol or max_iter; out [n, n], *iters (device int, may be NULL) = sweeps run. */
int avt_diag_filter_f32(const float* d1, int n, const float* w, int fs, float* out, void* stream);
int avt_q_learning_supported(int n);
int avt_q_learning_f32(const float* d3, int n, float alpha, float tol, int max_iter, float* out, int* iters,
                       void* stream);

/* Conv3d [1,3,3] 64 -> 64, stride 1, pad 1 (+ BN folded, optional ReLU) of the slow res2 blocks with the input
 * strip resident in LDS (csrc/conv33_c64.hip; same model, models/models.py:335, 399).  in [batch, t, h, w, 64],
 * out [batch, t, h, w, .] with row stride ldo elements, bf16.  wb [9 taps][2][4][64][8] in MFMA fragment order with
 * the output rows permuted: tile nt, row r -> channel 32*(nt/2) + 8*(r/4) + 4*(nt%2) + r%4, k = 32*kh + 8*q + e;
 * bias [64] fp32 in channel order.  avt_conv33_c64_supported(cin, cout, w): (64, 64, 56). */
int avt_conv33_c64_supported(int cin, int cout, int w);
int avt_conv33_c64_bf16(const void* in, const void* wb, const float* bias, void* out, int batch, int t,
                        int h, int w, int ldo, int relu, void* stream);

/* Two pointwise layers of consecutive slow-pathway bottlenecks in one pass over the rows (csrc/pw_chain.hip; same
 * model, models/models.py:335, 399):  y = ReLU(W1 x1 + b1 [+ res])  (block i's c conv + BN + residual + ReLU) and
 * z = ReLU(W2 [bf16(y) | x2] + b2)  (block i+1's a conv + BN + ReLU); y stays in registers between the two GEMMs.
 * x1 [m, ldx] (k1 channels used), res [m, ldr] or NULL, y [m, ldy] (n1 channels), x2 [m, ldx2] or NULL (k2x more
 * input channels of the second layer: the lateral features at the res2 -> res3 boundary), z [m, ldz] (n2 channels),
 * bf16 rows, strides in elements.  w1 [n1/16][ceil(k1/32)][64][8], w2 [n2/16][(n1 + k2x)/32][64][8]: MFMA fragments with the
 * output rows permuted (tile nt, row r -> channel 32*(nt/2) + 8*(r/4) + 4*(nt%2) + r%4, k = 32*ks + 8*q + e,
 * zero beyond k1); b1 [n1], b2 [n2] fp32 in channel order.
 * avt_pw_chain_supported(k1, n1, n2, has_res, k2x): (64, 256, 64, 1, 0), (144, 256, 64, 0, 0), (64, 256, 128, 1, 64),
 * (128, 512, 128, 1, 0). */
int avt_pw_chain_supported(int k1, int n1, int n2, int has_res, int k2x);
int avt_pw_chain_bf16(const void* x1, int ldx, int k1, const void* w1, const float* b1, const void* res, int ldr,
                      void* y, int ldy, int n1, const void* x2, int ldx2, int k2x, const void* w2, const float* b2,
                      void* z, int ldz, int n2, int64_t m, void* stream);

/* VGGish audio front-end (utils/mel_features.py:21-92, 176-205 log_mel_spectrogram; called once per
 * video from utils/vggish_utils.py:27-69), float64 like the reference's NumPy code:
 * frame f = wave[f*hop, f*hop+win) * window -> |DFT_fft_len| -> spec[fft_len/2+1] . melmat -> log(. + log_offset).
 * wave: float32 (wave_is_f64 = 0) or float64 device samples; window [win] and melmat [fft_len/2+1, n_mel]
 * are float64 device tables built by the caller with the reference's formulas (periodic Hann :41-43, HTK mel
 * matrix :117-173).  logmel [n_frames, n_mel], n_frames = 1 + (n_samples - win) / hop (incomplete tail
 * dropped, mel_features.py:21-45); n_samples < win writes nothing. */
int avt_logmel_f64(const void* wave, int wave_is_f64, int64_t n_samples,
                   const double* window, int win, int hop, int fft_len,
                   const double* melmat, int n_mel, double log_offset,
                   double* logmel, void* stream);

/* Example framing of vggish_utils.py:60-68 with the float32 cast of validate.py:160-161:
 * out[e, r, m] = (float) logmel[e*ex_hop + r, m], n_ex = 1 + (n_frames - ex_len) / ex_hop. */
int avt_logmel_examples_f32(const double* logmel, int64_t n_frames, int n_mel,
                            int ex_len, int ex_hop, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AVT_H */
