/* yv1.h -- C ABI of libyv1.so, the MI355X (gfx950) hot path of YOLO-v1 training.
 *
 * The reference (haoran1062/YOLO_V1) is pure Python on PyTorch: it has no FFI of its own.  Each
 * entry point below replaces the ATen/cuDNN work behind a reference call site (cited per group);
 * the Python binding a maintainer adds is the ctypes table in yolo_v1_amd/_lib.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors in the host layer);
 *     outputs and workspaces are caller-allocated; nothing is allocated or freed inside;
 *   - `stream` is the hipStream_t (passed as void*) all work is enqueued on; calls are asynchronous;
 *   - return value: 0 on success, a hipError_t value, or YV1_ERR_* below;
 *   - activations: NHWC bfloat16; `ld*` = pixel stride in elements (>= channels, multiple of 8), so a
 *     channel window of a wider buffer is addressed by offsetting the base pointer;
 *   - "stats partials": fp32 [rows][2][C] = per-tile (sum, sum of squares) rows to be summed.
 */
#ifndef YV1_H
#define YV1_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YV1_ERR_BAD_ARG 1001
#define YV1_ERR_UNSUPPORTED 1002
#define YV1_ERR_WORKSPACE 1003

typedef void* yv1_stream_t; /* hipStream_t */

/* ---- loss: v1Loss.py:22-118 (YOLOLossV1.forward) + its autograd backward; utils/utils.py:10-75 --------- */
size_t yv1_loss_workspace_bytes(int N, int S);
/* pred [N,S,S,B*5+C] fp32 with element strides ps0..ps3 (the reference passes a permuted NCHW view,
 * OriginResNet.py:189); target contiguous.  Writes the total (v1Loss.py:104-105, divided by batch_size),
 * the 4 raw component sums {location, contain, not-contain, classify} (v1Loss.py:108) and, when grad_pred
 * != NULL, d total / d pred (contiguous), including the gradient through the IoU target. */
int yv1_loss_fwd_bwd(const float* pred, long long ps0, long long ps1, long long ps2, long long ps3, const float* target,
                     int N, int S, int B, int C, float l_coord, float l_noobj, float batch_size, float* out_loss,
                     float* out_components, float* grad_pred, void* workspace, size_t workspace_bytes,
                     yv1_stream_t stream);
/* x[i] *= *scalar (device scalar): chains the upstream autograd gradient without a host sync */
int yv1_scale_by_device_scalar(float* x, const float* scalar, long long n, yv1_stream_t stream);

/* ---- decoder + NMS: utils/utils.py:94-147 (decoder), :150-184 (nms), :10-57, :59-75 ------------------ */
/* One workgroup per image.  pred [N,S,S,B*5+C] fp32 contiguous.  Outputs are [N][S*S*B] rows valid up to
 * out_counts[n]: boxes (x1,y1,x2,y2), class index (int64), score, candidate index (int64, position in the
 * (row,col,box)-ordered candidate list) in keep order; out_ncand[n] = candidates before NMS (0 means the
 * reference's single all-zero box was substituted, utils.py:134-137).  thresh is compared in double. */
int yv1_decode_nms_batched(const float* pred, int N, int S, int B, int C, double thresh, float nms_th, float* out_boxes,
                           long long* out_cls, float* out_scores, long long* out_keep_idx, int* out_counts,
                           int* out_ncand, yv1_stream_t stream);
/* ---- FP8 (OCP e4m3) inference convolution: BASELINE config 5 ------------------------------------------ */
/* conv + folded eval-mode BatchNorm + residual + ReLU in one launch (OriginResNet.py:87-107 in eval mode):
 *   t = bf16(acc * alpha[c] + beta[c]);  out = relu?(t + residual);  y_bf16 = bf16(out), y_fp8 = e4m3(clamp(bf16(out)))
 * x8 [N,IH,IW,*] e4m3 (pixel stride ldx bytes), w8 [Cout][k*k][Cin] e4m3 from yv1_prep_weights_fp8, residual /
 * y_bf16 bf16 with pixel strides ldr / ld16, y_fp8 e4m3 with pixel stride ld8; either output may be NULL.
 * Cin % 64 == 0, Cout % 64 == 0 (pad the weight rows).  MFMA: v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales. */
int yv1_conv2d_fwd_nhwc_fp8(const void* x8, const void* w8, const float* alpha, const float* beta, const void* residual,
                            int ldr, void* y_bf16, int ld16, void* y_fp8, int ld8, int N, int IH, int IW, int ldx, int Cin,
                            int Cout, int k, int stride, int pad, int relu, yv1_stream_t stream);
/* training form ("fp8 forward GEMMs, bf16 backward"): y = bf16(acc * alpha[c]), alpha = 1/q from
 * yv1_prep_weights_fp8_multi, zero_beta = Cout zeros; stats: yv1_conv2d_fp8_stats_rows(M, Cout) rows of [2][Cout]
 * partial sums for yv1_bn_finalize, as yv1_conv2d_fwd_nhwc_bf16 writes them */
int yv1_conv2d_fp8_stats_rows(int M, int Cout);
int yv1_conv2d_fwd_stats_nhwc_fp8(const void* x8, const void* w8, const float* alpha, const float* zero_beta, void* y,
                                  int ldy, float* stats, int N, int IH, int IW, int ldx, int Cin, int Cout, int k,
                                  int stride, int pad, yv1_stream_t stream);
int yv1_prep_weights_fp8_max_tensors(void);
int yv1_prep_weights_fp8_multi(const float* const* w, const long long* strides, const int* O, const int* I, const int* k,
                               void* const* w8, float* const* alpha, int n, yv1_stream_t stream);
/* bf16 NHWC -> e4m3 NHWC (saturating at +-448), C % 8 == 0 */
int yv1_quantize_bf16_to_fp8(const void* x, int ldx, void* y8, int ldy, long long npix, int C, yv1_stream_t stream);
/* fp32 OIHW (element strides so,si,sh,sw) -> e4m3 [Opad][k*k][Ipad]; q[o] (Opad floats) = the power of two each
 * output channel was multiplied by before rounding (largest with amax*q <= 448; 1 for all-zero / padded rows) */
int yv1_prep_weights_fp8(const float* w, long long so, long long si, long long sh, long long sw, int O, int I, int k,
                         int Opad, int Ipad, void* w8, float* q, yv1_stream_t stream);
/* alpha[c] = scale[c] / q[c], beta[c] = shift[c] for c < C, zero up to Cpad (scale/shift NULL = 1 / 0) */
int yv1_fp8_fold_bn(const float* scale, const float* shift, const float* q, int C, int Cpad, float* alpha, float* beta,
                    yv1_stream_t stream);

/* ---- target encoder: utils/YOLODataLoader.py:200-230 for a whole batch -------------------------------- */
/* boxes [N][Kmax][4] fp32 normalised (cx,cy,w,h), labels [N][Kmax] int64, counts [N] (boxes used per image).
 * target [N][S][S][B*5+C] fp32 is written completely (zeros where no box lands).  Later boxes replace earlier
 * ones in the same cell; a coordinate of exactly 0 wraps to the last cell (Python's index -1).  *err_flag is
 * set to 1 when a box falls outside the grid or a label outside [0,C) (the reference raises IndexError). */
int yv1_encode_targets(const float* boxes, const long long* labels, const int* counts, int N, int Kmax, int S, int B,
                       int C, float* target, int* err_flag, yv1_stream_t stream);

/* greedy class-agnostic NMS, n <= 896; out_keep: indices into the input in keep order */
int yv1_nms(const float* boxes, const float* scores, int n, float threshold, long long* out_keep, int* out_count,
            yv1_stream_t stream);
int yv1_iou_matrix(const float* b1, int n, const float* b2, int m, float* out, yv1_stream_t stream);
int yv1_convert_cxcywh_to_xyxy(const float* in, int n, int S, float* out, yv1_stream_t stream);

/* ---- convolution: nn.Conv2d behind OriginResNet.py:21-29,:121,:159-163; OriginDenseNet.py:24-29,:52-53,:77,:101 */
/* y = conv(x, w), square kernel k; w bf16 [Cout][k*k][Cin] (Cin % 32 == 0, Cout % 32 == 0).
 * stats (nullable): BatchNorm partials [yv1_conv2d_stats_rows(M,Cout,Cin,k,stride,pad)][2][Cout], M = N*OH*OW. */
int yv1_conv2d_fwd_nhwc_bf16(const void* x, const void* w, void* y, int N, int IH, int IW, int ldx, int Cin, int Cout,
                             int ldy, int k, int stride, int pad, float* stats, yv1_stream_t stream);
int yv1_conv2d_stats_rows(int M, int Cout, int Cin, int k, int stride, int pad);
/* 7x7/2 pad-3 stem over the packed image: xp [N][H+6][W+6][4] bf16; w bf16 [Cout][7][32] */
int yv1_conv2d_stem_fwd_bf16(const void* xp, const void* w, void* y, int N, int H, int W, int Cout, int ldy, float* stats,
                             yv1_stream_t stream);
int yv1_pack_input_nhwc4(const float* x_nchw, void* y, int N, int H, int W, yv1_stream_t stream);
/* inference forms (network in eval() mode, OriginResNet.py:87-107,:174-176): conv + folded BatchNorm (scale/shift per
 * output channel, yv1_bn_eval_coeffs) + bf16 residual + ReLU in one launch:
 *   t = bf16(acc * scale[c] + shift[c]);  y = bf16(relu?(t + residual));  without residual the ReLU precedes the rounding */
int yv1_conv2d_fwd_bn_act_nhwc_bf16(const void* x, const void* w, void* y, int N, int IH, int IW, int ldx, int Cin, int Cout,
                                    int ldy, int k, int stride, int pad, const float* scale, const float* shift,
                                    const void* residual, int ldres, int relu, yv1_stream_t stream);
int yv1_conv2d_stem_fwd_bn_act_bf16(const void* xp, const void* w, void* y, int N, int H, int W, int Cout, int ldy,
                                    const float* scale, const float* shift, int relu, yv1_stream_t stream);
/* dx (+)= conv_transpose(dy, w); wt bf16 [Cin][k*k][Cout]; stride 1 or 2.  1x1 strided: only the sampled
 * pixels of dx are written (accumulate into a dx the main path already wrote). */
int yv1_conv2d_dgrad_nhwc_bf16(const void* dy, const void* wt, void* dx, int N, int IH, int IW, int lddx, int Cin, int Cout,
                               int lddy, int k, int stride, int pad, int accumulate, yv1_stream_t stream);
/* 1x1 stride-1 dgrad with the identity shortcut's gradient folded into the epilogue (OriginResNet.py:104-105
 * backward): dx = dgrad(dy, wt) + (relu_mask bit ? g : 0); g [N,IH,IW,*] bf16 with pixel stride ldg, relu_mask
 * the [pixels][ldmask]-byte 1-bit mask yv1_bn_apply wrote for that block output. */
int yv1_conv2d_dgrad_add_masked_nhwc_bf16(const void* dy, const void* wt, void* dx, int N, int IH, int IW, int lddx,
                                          int Cin, int Cout, int lddy, const void* g, int ldg, const void* relu_mask,
                                          int ldmask, yv1_stream_t stream);
/* dw fp32 [Cout][k*k][Cin] (= channels_last storage of the OIHW gradient) */
size_t yv1_conv2d_wgrad_workspace_bytes(int N, int OH, int OW, int Cin, int Cout, int k);
int yv1_conv2d_wgrad_nhwc_bf16(const void* x, const void* dy, float* dw, int N, int IH, int IW, int ldx, int Cin, int Cout,
                               int lddy, int k, int stride, int pad, void* workspace, size_t workspace_bytes,
                               yv1_stream_t stream);
/* The same for a launch that overlaps with another stream's kernels (the backward runs the weight gradients on a side
 * stream beside the dgrad / BatchNorm chain): narrower split-K -- fewer slabs, fewer workgroups taken from the other
 * stream.  Same workspace size query; the result differs from the entry above in fp32 summation order only. */
int yv1_conv2d_wgrad_shared_nhwc_bf16(const void* x, const void* dy, float* dw, int N, int IH, int IW, int ldx, int Cin,
                                      int Cout, int lddy, int k, int stride, int pad, void* workspace,
                                      size_t workspace_bytes, yv1_stream_t stream);
size_t yv1_conv2d_stem_wgrad_workspace_bytes(int N, int H, int W, int Cout);
int yv1_conv2d_stem_wgrad_bf16(const void* xp, const void* dy, float* dw, int N, int H, int W, int Cout, int lddy,
                               void* workspace, size_t workspace_bytes, yv1_stream_t stream);

/* ---- BatchNorm (train mode), ReLU, residual add: OriginResNet.py:90-105,:174-177,:187; OriginDenseNet.py:22-27 */
int yv1_reduce_rows(const float* in, float* out, int rows, int W, int RB, yv1_stream_t stream);
int yv1_stats_merge(const float* partials, int rows, int Cseg, float* table, int ldo, int c0, yv1_stream_t stream);
int yv1_bn_finalize(const float* partials, int rows, int C, int ld_partials, float count, const float* gamma,
                    const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                    float* invstd, float* scale, float* shift, yv1_stream_t stream);
/* DenseNet: finalize the first C channels of the shared one-row table [2][ld] of channel sums (OriginDenseNet.py:32-36,
 * norm1 over the concatenated features) while merging the partial rows seg_part [seg_rows][2][seg_c] of the features
 * the previous layer produced into table columns [seg_c0, seg_c0+seg_c) -- yv1_stats_merge + yv1_bn_finalize in one launch */
int yv1_bn_finalize_merged(float* table, int C, int ld, float count, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                           float* scale, float* shift, const float* seg_part, int seg_rows, int seg_c0, int seg_c,
                           yv1_stream_t stream);
int yv1_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, float* scale, float* shift, yv1_stream_t stream);
/* z = relu?(scale*y + shift [+ residual | + res_scale*residual + res_shift]); relu_mask (nullable): uint8
 * [npix][C/8], bit k of byte j = (channel 8j+k > 0), so the backward need not re-read z for its sign */
int yv1_bn_apply(const void* y, int ldy, void* z, int ldz, const void* residual, int ldr, const float* scale,
                 const float* shift, const float* res_scale, const float* res_shift, long long npix, int C, int relu,
                 void* relu_mask, yv1_stream_t stream);
/* the same with a second, e4m3 copy of z (pixel stride ldz8 bytes): the operand of the next fp8 convolution */
int yv1_bn_apply_q8(const void* y, int ldy, void* z, int ldz, const void* residual, int ldr, const float* scale,
                 const float* shift, const float* res_scale, const float* res_shift, long long npix, int C, int relu,
                 void* relu_mask, void* z8, int ldz8,
                    yv1_stream_t stream);
int yv1_bn_reduce_rows(long long npix, int C);
int yv1_bn_stats(const void* y, int ldy, long long npix, int C, float* partials, yv1_stream_t stream);
/* mask_mode: 0 none, 1 ReLU mask from z > 0, 2 from scale*y+shift > 0, 3 z is yv1_bn_apply's relu_mask (ldz = C/8 bytes) */
int yv1_bn_bwd_reduce(const void* dz, int lddz, const void* z, int ldz, const void* y, int ldy, const float* mean,
                      const float* invstd, const float* scale, const float* shift, long long npix, int C, int mask_mode,
                      float* partials, yv1_stream_t stream);
int yv1_bn_bwd_finalize(const float* partials, int rows, int C, float count, const float* gamma, const float* invstd,
                        float* dgamma, float* dbeta, float* k1, float* k2, float* k3, yv1_stream_t stream);
int yv1_bn_bwd_apply(const void* dz, int lddz, const void* z, int ldz, const void* y, int ldy, const float* mean,
                     const float* invstd, const float* scale, const float* shift, const float* k1, const float* k2,
                     const float* k3, long long npix, int C, int mask_mode, void* dy, int lddy, void* dres, int lddres,
                     int accumulate, yv1_stream_t stream);
/* Dual form for a projection Bottleneck (OriginResNet.py:100-105: out = relu(bn3(y3) + bn_d(yd))): bn3 and the
 * downsample BatchNorm receive the same ReLU-masked gradient; one reduction pass and one apply pass read dz and the
 * mask once for both.  partials / partials2 have yv1_bn_reduce_rows(npix, C) rows each (finalize them separately). */
int yv1_bn_bwd_reduce_dual(const void* dz, int lddz, const void* z, int ldz, const void* y, int ldy, const float* mean,
                           const float* invstd, const void* y2, int ldy2, const float* mean2, const float* invstd2,
                           long long npix, int C, int mask_mode, float* partials, float* partials2, yv1_stream_t stream);
int yv1_bn_bwd_apply_dual(const void* dz, int lddz, const void* z, int ldz, const void* y, int ldy, const float* mean,
                          const float* invstd, const float* k1, const float* k2, const float* k3, void* dy, int lddy,
                          const void* y2, int ldy2, const float* mean2, const float* invstd2, const float* k1b,
                          const float* k2b, const float* k3b, void* dy2, int lddy2, long long npix, int C, int mask_mode,
                          yv1_stream_t stream);
/* Finalize fused into the consuming launch (round 3).  Replaces the launch PAIRS yv1_bn_finalize + yv1_bn_apply(_q8) and
 * yv1_bn_bwd_finalize + yv1_bn_bwd_apply(_dual) of a training-mode nn.BatchNorm2d (OriginResNet.py:90-105 forward, its
 * autograd backward): the first workgroups of the grid finalize the partial rows (same arithmetic, fixed summation order),
 * publish the per-channel coefficients with agent-scope stores and exit; the other workgroups wait for them, then stream the
 * tensor.  `sync`: TWO zero-initialised unsigned words owned by this launch until it completes (the kernel leaves them
 * zero again, so a captured graph replays without a memset); `fault`: one unsigned word the caller checks after a
 * synchronize -- set to 1 if a consumer gave up waiting (~1 s; the spin's exit condition; never observed).
 * yv1_bn_finalize_apply: partials [rows][2][ld_partials] of the BatchNorm applied to y; r_partials (nullable) the same for
 * the BatchNorm of `residual` (projection shortcut; then r_mean..r_shift are outputs), else r_scale / r_shift are inputs
 * (already final) or NULL (plain residual / none).  rows <= 2048 (pre-reduce longer tables with yv1_reduce_rows). */
int yv1_bn_finalize_apply(const float* partials, int rows, int ld_partials, float count, const float* gamma,
                          const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                          float* invstd, float* scale, float* shift, const float* r_partials, int r_rows, int r_ld_partials,
                          const float* r_gamma, const float* r_beta, float* r_running_mean, float* r_running_var,
                          float* r_mean, float* r_invstd, float* r_scale, float* r_shift, const void* y, int ldy, void* z,
                          int ldz, const void* residual, int ldr, long long npix, int C, int relu, void* relu_mask, void* z8,
                          int ldz8, unsigned* sync, unsigned* fault, yv1_stream_t stream);
/* DenseNet form (OriginDenseNet.py:32-36, norm1 over the concatenation): yv1_bn_finalize_merged + yv1_bn_apply in one
 * launch; seg_part may be NULL (plain one-row table: transition norm / norm5 with nothing pending). */
int yv1_bn_finalize_merged_apply(float* table, int C, int ld, float count, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                                 float* scale, float* shift, const float* seg_part, int seg_rows, int seg_c0, int seg_c,
                                 const void* y, int ldy, void* z, int ldz, long long npix, int relu, unsigned* sync,
                                 unsigned* fault, yv1_stream_t stream);
/* partials [rows][2][C] from yv1_bn_bwd_reduce; k1/k2/k3: scratch [C] each; other arguments as yv1_bn_bwd_apply. */
int yv1_bn_bwd_finalize_apply(const float* partials, int rows, float count, const float* gamma, float* dgamma, float* dbeta,
                              float* k1, float* k2, float* k3, const void* dz, int lddz, const void* z, int ldz,
                              const void* y, int ldy, const float* mean, const float* invstd, const float* scale,
                              const float* shift, long long npix, int C, int mask_mode, void* dy, int lddy, void* dres,
                              int lddres, int accumulate, unsigned* sync, unsigned* fault, yv1_stream_t stream);
/* partials / partials2 from yv1_bn_bwd_reduce_dual; k6: scratch [6][C]; other arguments as yv1_bn_bwd_apply_dual. */
int yv1_bn_bwd_finalize_apply_dual(const float* partials, const float* partials2, int rows, float count, const float* gamma,
                                   const float* gamma2, float* dgamma, float* dbeta, float* dgamma2, float* dbeta2, float* k6,
                                   const void* dz, int lddz, const void* z, int ldz, const void* y, int ldy,
                                   const float* mean, const float* invstd, void* dy, int lddy, const void* y2, int ldy2,
                                   const float* mean2, const float* invstd2, void* dy2, int lddy2, long long npix, int C,
                                   int mask_mode, unsigned* sync, unsigned* fault, yv1_stream_t stream);
/* "bn3's backward as algebra" (round 3).  An identity-shortcut Bottleneck ends
 * out = relu(bn3(conv3(z2)) + x) (OriginResNet.py:97-105); conv3 is pointwise, so BatchNorm-3's backward commutes with it:
 * from the MASKED output gradient gm, T = gm^T z2 (yv1_conv2d_wgrad_nhwc_bf16 fed gm), G = z2^T z2 and column sums,
 *   dgamma/dbeta/k1/k2/k3*is  <- yv1_bn3_coeffs       (no pass over y3)
 *   wcat = [diag(k1) W3 ; -W3^T diag(k3*is) W3], bias  <- yv1_bn3_build
 *   dz2 = [gm | z2] wcat^T + bias                      <- yv1_conv2d_dgrad_cat_bias_nhwc_bf16 (second K source, bias epilogue)
 *   dW3                                                <- yv1_bn3_dw
 * and gm itself, with its column sums, comes out of the conv1 data gradient of the block above
 * (yv1_conv2d_dgrad_add_masked_out_nhwc_bf16: output mask + per-tile sums in the epilogue).  The stand-alone
 * yv1_bn_bwd_reduce / _finalize / _apply passes over the 4p-wide tensors do not run; dy3 is never formed. */
int yv1_conv2d_dgrad_add_masked_out_nhwc_bf16(const void* dy, const void* wt, void* dx, int N, int IH, int IW, int lddx,
                                              int Cin, int Cout, int lddy, const void* g, int ldg, const void* relu_mask,
                                              int ldmask, const void* out_mask, int ldom, float* gsum,
                                              yv1_stream_t stream);
int yv1_conv2d_dgrad_out_nhwc_bf16(const void* dy, const void* wt, void* dx, int N, int IH, int IW, int lddx, int Cin, int Cout,
                                   int lddy, int stride, int accumulate, const void* out_mask, int ldom, float* gsum,
                                   yv1_stream_t stream);   /* 1x1 pad-0 data gradient (stride 1 | 2) + output mask + sums */
int yv1_conv2d_dgrad_gsum_rows(int M, int Cin, int Cout);
int yv1_conv2d_dgrad_cat_bias_nhwc_bf16(const void* g, int ldg, int C1, const void* z, int ldz, int C2, const void* wcat,
                                        const float* one, const float* bias, void* dx, int lddx, int Cdx, int N, int H, int W,
                                        int stride, int accumulate, const void* out_mask, int ldom, float* gsum,
                                        yv1_stream_t stream);   /* gsum rows: yv1_conv2d_dgrad_gsum_rows(N*H*W, Cdx, C1+C2) */
/* x [N,H,W,*] -> y [N,H/2,W/2,*]: the pixels a stride-2 1x1 convolution reads, made dense (bf16, C % 8 == 0) */
int yv1_subsample2_nhwc_bf16(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, yv1_stream_t stream);
int yv1_bn3_coeffs(const float* gsum, int rows, const float* T, const void* w3, int p, int C4, const float* mean,
                   const float* invstd, const float* gamma, float count, float* dgamma, float* dbeta, float* k1, float* k2,
                   float* k3is, yv1_stream_t stream);
int yv1_bn3_build(const void* w3, int p, int C4, const float* k1, const float* k2, const float* k3is, const float* mean,
                  void* wcat, float* bias, yv1_stream_t stream);
int yv1_bn3_dw(const float* T, const float* G, const float* sz_partials, int rows, const void* w3, int p, int C4,
               const float* k1, const float* k2, const float* k3is, const float* mean, float* dW, yv1_stream_t stream);
/* Deferred BatchNorm backward (round 3): relu(bn(x)) feeding a pointwise convolution -- DenseNet's
 * norm1 -> relu1 -> conv1 over the concatenated features (OriginDenseNet.py:22-27,:32-36) and the transitions' norm -> relu ->
 * conv (:50-52).  dx = a*d - a*mean(d) - a*xhat*mean(d*xhat) with d the ReLU-masked gradient at the BatchNorm output and
 * a = gamma*invstd:  the first term is added by the data gradient's own epilogue (which also emits the two sums per pixel
 * tile: part [rows][2][Cin] = sum d, sum d*(x - mean)), the other two are affine in x per channel: yv1_bn_bwd_finalize_deferred
 * returns their coefficients KA / KB (and dgamma / dbeta); they are subtracted by the NEXT data gradient into the same
 * buffer (pend_a / pend_b, same epilogue pass) or, for channels no later launch covers, by yv1_bn_deferred_fix before the
 * gradient of those channels is consumed -- the buffer never holds more than one layer's uncorrected term.  Replaces
 * yv1_conv2d_dgrad_nhwc_bf16 + yv1_bn_bwd_reduce + yv1_bn_bwd_finalize + yv1_bn_bwd_apply(accumulate) there; the masked
 * gradient tensor is never stored.  1x1, stride 1, pad 0; Cout (the convolution's output channels) % 64 == 0. */
int yv1_conv2d_dgrad_bn_deferred_nhwc_bf16(const void* dy, const void* wt, void* dx, int N, int IH, int IW, int lddx, int Cin,
                                           int Cout, int lddy, const void* x, int ldx, const float* scale, const float* shift,
                                           const float* mean, int accumulate, float* part, int wt_rows, const float* pend_a,
                                           const float* pend_b, yv1_stream_t stream);   /* pend_*: nullable, see below */
int yv1_conv2d_dgrad_bn_deferred_rows(int M, int Cin, int Cout, int wt_rows);   /* wt_rows: readable (zero-padded) rows of wt */
int yv1_bn_bwd_finalize_deferred(const float* part, int rows, int C, float count, const float* gamma, const float* mean,
                                 const float* invstd, float* dgamma, float* dbeta, float* KA, float* KB, int accumulate,
                                 yv1_stream_t stream);   /* rows <= 2048 (yv1_reduce_rows first) */
int yv1_bn_deferred_fix(void* g, int ldg, const void* x, int ldx, const float* KA, const float* KB, long long npix, int C,
                        yv1_stream_t stream);            /* g -= KA + KB * x over a C-channel window */
/* The reduction pass of a BatchNorm(+ReLU) backward folded into the epilogue of the data gradient that PRODUCES its input
 * gradient (round 3; the mirror image of the forward's statistics epilogue): for a stride-1 k x k convolution whose input was
 * relu(bn(y)) -- conv2 after bn1 / conv3 after bn2 of a Bottleneck (OriginResNet.py:90-99), conv2 after norm2 of a _DenseLayer
 * (OriginDenseNet.py:26-31) -- dx is stored MASKED (scale*y + shift > 0) and part gets yv1_bn_bwd_reduce's layout
 * [rows][2][Cin] = sum d, sum d*xhat per pixel tile; yv1_bn_bwd_finalize + yv1_bn_bwd_apply(mask_mode 0) finish the BatchNorm
 * backward, yv1_bn_bwd_reduce's pass over (dx, y) does not run.  rows() == 0: this shape has no such kernel, keep the passes. */
int yv1_conv2d_dgrad_bn_sums_nhwc_bf16(const void* dy, const void* wt, void* dx, int N, int IH, int IW, int lddx, int Cin,
                                       int Cout, int lddy, int k, int pad, const void* y, int ldy, const float* scale,
                                       const float* shift, const float* mean, const float* invstd, float* part,
                                       yv1_stream_t stream);
int yv1_conv2d_dgrad_bn_sums_rows(int M, int Cin, int Cout, int k, int pad);
/* BatchNorm(+ReLU) backward behind the stem's 3x3/2 max pool (OriginResNet.py:174-177, OriginDenseNet.py:120-128; autograd
 * of nn.MaxPool2d + nn.ReLU + nn.BatchNorm2d): dpool [N,OH,OW,C] is the gradient of the pool OUTPUT, pool_idx what
 * yv1_maxpool3x3s2_fwd stored; the pool's backward is gathered on the fly, so the 4x larger gradient of the pool input is
 * never materialised.  y: BatchNorm input [N,H,W,C].  mask_mode 0 or 2.  Same results as yv1_maxpool3x3s2_bwd followed by
 * yv1_bn_bwd_reduce / _apply, bit for bit. */
int yv1_bn_bwd_reduce_pooled(const void* dpool, int lddp, const void* pool_idx, const void* y, int ldy, const float* mean,
                             const float* invstd, const float* scale, const float* shift, int N, int H, int W, int C,
                             int mask_mode, float* partials, yv1_stream_t stream);
int yv1_bn_bwd_apply_pooled(const void* dpool, int lddp, const void* pool_idx, const void* y, int ldy, const float* mean,
                            const float* invstd, const float* scale, const float* shift, const float* k1, const float* k2,
                            const float* k3, int N, int H, int W, int C, int mask_mode, void* dy, int lddy,
                            yv1_stream_t stream);

/* ---- pooling and head: OriginResNet.py:125,:188-189; OriginDenseNet.py:54,:80,:127-128 --------------------- */
/* idx (nullable): uint8 [N,OH,OW,C] = window position of the first maximum (torch's tie rule), for the backward */
int yv1_maxpool3x3s2_fwd(const void* x, int ldx, void* y, int ldy, void* idx, int N, int H, int W, int C,
                         yv1_stream_t stream);
/* stems (OriginResNet.py:175-177, OriginDenseNet.py:120-128): pooled = maxpool3x3s2(bf16(relu?(y*scale + shift))) in one
 * pass -- the values yv1_bn_apply + yv1_maxpool3x3s2_fwd produce, bit for bit, without the intermediate tensor */
int yv1_bn_act_maxpool3x3s2_fwd(const void* y, int ldy, const float* scale, const float* shift, int relu, void* pooled,
                                int ldp, void* idx, int N, int H, int W, int C, yv1_stream_t stream);
/* give idx (fast) or x (the forward input; the first maximum is re-derived) */
int yv1_maxpool3x3s2_bwd(const void* x, int ldx, const void* idx, const void* dy, int lddy, void* dx, int lddx, int N, int H,
                         int W, int C, yv1_stream_t stream);
int yv1_avgpool2_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, yv1_stream_t stream);
int yv1_avgpool2_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int H, int W, int C, yv1_stream_t stream);
int yv1_head_sigmoid_fwd(const void* y, int ldy, const float* scale, const float* shift, float* out, long long npix,
                         int C, yv1_stream_t stream);
int yv1_head_sigmoid_bwd(const float* dout, const float* out, const void* y, int ldy, const float* gamma, const float* mean,
                         const float* invstd, void* dy, int lddy, float* dgamma, float* dbeta, long long npix, int C,
                         yv1_stream_t stream);

/* ---- weight re-layout (fp32 OIHW parameter, any strides -> bf16 kernel layouts) ------------------------------ */
int yv1_prep_weights(const float* w, long long so, long long si, long long sh, long long sw, int O, int I, int KH, int KW,
                     int Opad, int Ipad, void* dst_fwd, void* dst_t, yv1_stream_t stream);
int yv1_prep_weights_max_tensors(void);
/* the same for `count` weights in ONE launch; every array is a HOST array (strides4 = 4 strides per weight) */
int yv1_prep_weights_multi(const float* const* w, const long long* strides4, const int* O, const int* I, const int* K,
                           const int* Opad, const int* Ipad, void* const* dst_fwd, void* const* dst_t, int count,
                           yv1_stream_t stream);
int yv1_prep_stem_weights(const float* w, long long so, long long si, long long sh, long long sw, int O, void* dst,
                          yv1_stream_t stream);
int yv1_unpack_stem_grad(const float* g, float* dw, long long so, long long si, long long sh, long long sw, int O,
                         yv1_stream_t stream);

/* ---- optimizer: torch.optim.SGD(momentum=0.99).step(), train.py:84,:172 ------------------------------------ */
int yv1_sgd_max_tensors(void);
/* buf = momentum*buf + grad_scale*g ; w -= (*lr)*buf over `count` dense fp32 tensors.  w/g/m/n are HOST arrays
 * (of device pointers / element counts); *lr is read on the device so the step can live in a hipGraph. */
int yv1_sgd_momentum_step(float* const* w, const float* const* g, float* const* m, const long long* n, int count,
                          const float* lr, float momentum, float grad_scale, yv1_stream_t stream);

/* ---- introspection (tests) ------------------------------------------------------------------------------------ */
/* The kernel template(s) the most recent convolution / dgrad / wgrad entry point launched on the calling host thread,
 * ';'-separated, e.g. "k_conv_dma<128,256,32,2,2,3> direct".  Copies at most cap-1 characters (NUL-terminated) and
 * returns the full length.  buf is a HOST pointer.  The reference has no counterpart: cuDNN picks its algorithm
 * silently behind nn.Conv2d (OriginResNet.py:21-29); here the parity tests assert which tile they covered. */
int yv1_last_config(char* buf, int cap);

#ifdef __cplusplus
}
#endif
#endif /* YV1_H */
