/*
 * cpmrcnn_hip.h -- C ABI of libcpmrcnn_hip.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary for the reference's native extension
 * `pet.lib.ops._C` (pybind module: pet/lib/ops/csrc/vision.cpp:20-48) and for the
 * ATen/cuDNN calls the reference's Python modules make on the hot path.  Every
 * entry point takes plain device pointers, sizes and a HIP stream; there are no
 * torch types.  All functions return 0 on success, a negative CPM_E* code on an
 * argument error (the host wrapper raises RuntimeError, as AT_ASSERTM did) and
 * never synchronise the device.  `stream` is a hipStream_t passed as void*.
 *
 * Tensor layouts: "NCHW" is the reference's; "NHWC" is our resident layout
 * (torch channels_last: same logical shape, C fastest in memory).
 */
#ifndef CPMRCNN_HIP_H
#define CPMRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPM_OK 0
#define CPM_EINVAL (-1)      /* bad argument (shape, enum, null pointer) */
#define CPM_EWORKSPACE (-2)  /* workspace too small */
#define CPM_ELAUNCH (-3)     /* hipLaunch failed (see cpm_last_error) */

#define CPM_LAYOUT_NCHW 0
#define CPM_LAYOUT_NHWC 1

/* library / device info ---------------------------------------------------- */
int cpm_abi_version(void);
const char* cpm_last_error(void);

/* ---- RoIAlign ------------------------------------------------------------
 * Replaces _C.roi_align_forward / _C.roi_align_backward
 * (pet/lib/ops/csrc/ROIAlign/ROIAlign.h:57-146, ROIAlign_cuda.cu:178-487).
 * input [B,C,H,W], rois [K,5]=(batch,x1,y1,x2,y2), output [K,C,PH,PW] in `layout`.
 * interp: 0 bilinear, 1 nearest.  backward ACCUMULATES into grad_input (caller
 * zeroes it; the reference allocates zeros, ROIAlign_cuda.cu:447).            */
int cpm_roi_align_forward(const float* input, const float* rois, int K, int B, int C, int H, int W,
                          float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, int aligned,
                          int interp, int layout, float* output, void* stream);
int cpm_roi_align_backward(const float* grad_output, const float* rois, int K, int B, int C, int H, int W,
                           float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, int aligned,
                           int interp, int layout, float* grad_input, void* stream);

/* Fused multi-level pooler: replaces Pooler.forward + LevelMapper
 * (pet/rcnn/utils/poolers.py:9-40,90-132): one launch for all FPN levels, the
 * level of each RoI computed in-kernel, result written in RoI order (NHWC).
 * feats/grad_feats: host array of `num_levels` device pointers [B,H_l,W_l,C];
 * hs/ws/scales: host arrays.  levels_out (device int32 [K], may be NULL).    */
int cpm_roi_align_fpn_forward(const float* const* feats, const int* hs, const int* ws, const float* scales,
                              int num_levels, const float* rois, int K, int B, int C, int pooled_h, int pooled_w,
                              int sampling_ratio, float k_min, float k_max, float canonical_scale,
                              float canonical_level, float eps, float* output, int32_t* levels_out, void* stream);
int cpm_roi_align_fpn_backward(const float* grad_output, float* const* grad_feats, const int* hs, const int* ws,
                               const float* scales, int num_levels, const float* rois, int K, int B, int C,
                               int pooled_h, int pooled_w, int sampling_ratio, float k_min, float k_max,
                               float canonical_scale, float canonical_level, float eps, void* stream);
/* The same gradient WITHOUT float atomics: every 8x8 tile of every level is owned by one workgroup that gathers from
 * the RoIs binned onto it (a pre-pass lists them) and writes each pixel once.  Levels whose bit is set in
 * accumulate_mask are added to (read-modify-write), the others are cleared and written here (no caller-side zero fill).
 * Bit-reproducible (per-tile RoI lists are sorted).  pooled_h/w <= 16, C % 4 == 0, K <= 8192, maps 16-byte aligned.
 * workspace: cpm_roi_align_fpn_gather_workspace_bytes(hs, ws, num_levels, B, K) bytes of device scratch. */
size_t cpm_roi_align_fpn_gather_workspace_bytes(const int* hs, const int* ws, int num_levels, int B, int K);
int cpm_roi_align_fpn_backward_gather(const float* grad_output, float* const* grad_feats, const int* hs,
                                      const int* ws, const float* scales, int num_levels, const float* rois, int K,
                                      int B, int C, int pooled_h, int pooled_w, int sampling_ratio, float k_min,
                                      float k_max, float canonical_scale, float canonical_level, float eps,
                                      int accumulate_mask, void* workspace, size_t workspace_bytes, void* stream);
/* Several RoI sets -- the pooled gradients of several heads on the same pyramid, each with its own RoIs [K_s,5], pooled
 * size and sampling ratio (HOST arrays of n_sets <= 8 entries; grad_outputs[s] NHWC [K_s][ph_s][pw_s][C]) -- in ONE
 * pass over the tiles: grad_feats receive the sum of what n_sets calls of the function above would add.  A pixel's
 * contributions are added in (set, RoI, bin) order: bit-reproducible.  workspace sized for K = sum K_s. */
int cpm_roi_align_fpn_backward_gather_sets(int n_sets, const float* const* grad_outputs, const float* const* rois,
                                           const int* Ks, const int* pooled_hs, const int* pooled_ws,
                                           const int* sampling_ratios, float* const* grad_feats, const int* hs,
                                           const int* ws, const float* scales, int num_levels, int B, int C, float k_min,
                                           float k_max, float canonical_scale, float canonical_level, float eps,
                                           int accumulate_mask, void* workspace, size_t workspace_bytes, void* stream);

/* ---- NMS -------------------------------------------------------------------
 * Replaces torchvision.ops.nms as bound at pet/lib/ops/nms.py:2,10 (labels ==
 * NULL) and _C.ml_nms (pet/lib/ops/csrc/NMS/ml_nms.h:16-39, ml_nms.cu:11-146).
 * Batched over P independent segments (e.g. image x FPN level): segment p covers
 * rows [h_offsets[p], h_offsets[p+1]) of boxes/scores/labels (h_offsets is a
 * HOST array).  Per segment: stable sort by score descending, greedy suppression
 * of same-label boxes with IoU > thr (areas without +1), at most `topk` kept
 * (0 = unlimited).  keep [total] receives, for segment p at keep+h_offsets[p],
 * the kept row indices RELATIVE to the segment start in descending-score order;
 * keep_count [P] (device) the number kept.  Sort + mask + sweep all run on the
 * device (no D2H mask copy as in ml_nms.cu:117).                               */
size_t cpm_nms_workspace_bytes(const int32_t* h_offsets, int P);
int cpm_nms_batched(const float* boxes, const float* scores, const int64_t* labels, const int32_t* h_offsets,
                    int P, float iou_threshold, int topk, int64_t* keep, int32_t* keep_count, void* workspace,
                    size_t workspace_bytes, void* stream);
/* The same for unlabelled segments whose scores ALREADY descend along each segment (the RPN's: its pre-NMS top-k,
 * inference.py:79-84, leaves them sorted): the stable sort is the identity, so sort and gather are skipped.  Same keep
 * lists as cpm_nms_batched on such input; the caller vouches for the order. */
int cpm_nms_batched_presorted(const float* boxes, const float* scores, const int32_t* h_offsets, int P,
                              float iou_threshold, int topk, int64_t* keep, int32_t* keep_count, void* workspace,
                              size_t workspace_bytes, void* stream);

/* box_iou: replaces _C.box_iou (pet/lib/ops/csrc/Box_ops/box_iou.h:13-30). out [N,K], no +1 */
int cpm_box_iou(const float* boxes, int N, const float* query, int K, float* out, void* stream);

/* PoolPointsInterp: replaces _C.pool_points_interp_{forward,backward}
 * (pet/lib/ops/csrc/PoolPointsInterp/PoolPointsInterp.h:23-62); NCHW, pts [K,3],
 * batch index = k / 196 as in PoolPointsInterp_cuda.cu:74.  backward accumulates. */
int cpm_pool_points_interp_forward(const float* input, const float* pts, int K, int B, int C, int H, int W,
                                   float spatial_scale, float* output, void* stream);
int cpm_pool_points_interp_backward(const float* grad_output, const float* pts, int K, int B, int C, int H, int W,
                                    float spatial_scale, float* grad_input, void* stream);

/* ---- Convolution (implicit GEMM on fp32 MFMA) ------------------------------
 * Replaces the ATen/cuDNN conv2d / conv_transpose2d / linear calls made by
 * nn.Conv2d, nn.Linear and nn.ConvTranspose2d in pet/models/imagenet/resnet.py:71-136,
 * pet/rcnn/modeling/fpn/FPN.py:96-121, rpn/rpn.py:34-41, grid_rcnn/heads/{grid,cls}_heads.py,
 * grid_rcnn/outputs.py:49-104.  All tensors NHWC fp32; weights KRSC
 * ([K][R][S][C/groups], i.e. torch [K,C/g,R,S] in channels_last).
 *
 * cpm_conv2d_forward: y = epilogue(conv(x, w)), epilogue(v)[k] =
 *     relu?( v*scale[k] + shift[k] + residual[m,k] )      (scale/shift/residual may be NULL)
 *   covers conv+bias, conv+frozen AffineChannel2d (pet/lib/ops/affine.py:15-17),
 *   +residual add +ReLU of Bottleneck.forward.
 *   res_mode 0: residual has y's shape; 1: residual is [N,ceil(P/2),ceil(Q/2),K]
 *   and is read at (p/2,q/2) (FPN nearest-2x top-down add, FPN.py:104-106).
 * cpm_conv2d_backward_data: dx = conv^T(dy, w)  (also used as the forward of
 *   nn.ConvTranspose2d); accumulate!=0 adds into dx.
 * cpm_conv2d_backward_weight: dw += x (*) dy   (always accumulates: dw is a
 *   slice of the flat gradient buffer zeroed once per step).
 * split_k <= 0 lets the library choose.  Workspace: cpm_conv2d_workspace_bytes -- it holds the data gradient's
 * re-laid weight image and, in deterministic mode (cpm_set_deterministic), the SLAB PLANES of split reductions: every
 * split stores its partial tile into its own plane with plain stores and one pass folds the planes in split order (and
 * runs the fused epilogue).  Otherwise, and with a NULL / short workspace, the splits add with float atomics. */
typedef struct {
  int N, H, W, C;        /* input  [N,H,W,C]  */
  int K, R, S;           /* weight [K,R,S,C/groups] */
  int stride, pad, dilation, groups;
  int P, Q;              /* output [N,P,Q,K]  */
} cpm_conv_desc;

/* Arithmetic of the conv family (process-wide; set before launching, not thread-safe against running calls):
 *   CPM_MATH_F32    v_mfma_f32_32x32x2_f32: an exact fp32 fmaf chain per output.
 *   CPM_MATH_BF16X3 every fp32 operand split as hi + lo bf16 (x - bf16(x) is exact), a*b = ah*bh + ah*bl + al*bh on
 *                   v_mfma_f32_32x32x16_bf16 with fp32 accumulation: ~2^-17 relative error per product (the
 *                   north_star tolerance for conv tensors is 1e-3), fp32 exponent range, 3/16 of the MFMA time.
 * All three conv kernels honour it (narrow <=32-channel weight-gradient tiles stay on the fp32 MFMA). */
#define CPM_MATH_F32 0
#define CPM_MATH_BF16X3 1
int cpm_set_conv_math(int mode);
int cpm_get_conv_math(void);
/* Deterministic reductions for the conv family (process-wide, default off; environment CPM_DETERMINISTIC=1 at load):
 * every split reduction of the conv family (the k loop of a thin forward / data gradient, the pixel loop of a weight
 * gradient) goes through workspace slab planes folded in split order instead of float atomics -- two runs then produce
 * bit-identical activations, data gradients and weight gradients.  0-9 % slower per training step (measured: R-50
 * equal, R-101 +4..9 %).  Per-channel parameter sums (bias / GroupNorm affine gradients) and the loss scalars still use
 * float atomics. */
int cpm_set_deterministic(int on);
int cpm_get_deterministic(void);

size_t cpm_conv2d_workspace_bytes(const cpm_conv_desc* d);
int cpm_conv2d_forward(const cpm_conv_desc* d, const float* x, const float* w, const float* scale,
                       const float* shift, const float* residual, int res_mode, int relu, float* y,
                       void* workspace, size_t workspace_bytes, void* stream);
/* Pre-split weight images (bf16x3 arithmetic only).  The weight operand of a convolution is the same for every pixel
 * tile and every step until the optimizer moves it, yet the bf16x3 kernels split it into bf16 hi / lo again in every
 * workgroup that loads it (a third to a half of their conversion work).  cpm_split_w4 writes the split once: elements
 * 4i .. 4i+3 of a float array become 8 bytes of bf16 hi and 8 bytes of bf16 lo (lo = bf16(x - hi)) at byte offset 16 i,
 * i.e. an image of the array's own size and offsets (n % 4 == 0, 16-byte aligned buffers).  cpm_conv2d_forward_w4 is
 * cpm_conv2d_forward with `w4` = that image of the KRSC weight (C / groups % 4 == 0) and bit-identical results. */
int cpm_split_w4(const float* src, void* dst, int64_t n, void* stream);
int cpm_conv2d_forward_w4(const cpm_conv_desc* d, const float* x, const void* w4, const float* scale,
                          const float* shift, const float* residual, int res_mode, int relu, float* y,
                          void* workspace, size_t workspace_bytes, void* stream);
int cpm_conv2d_backward_data(const cpm_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate,
                             void* workspace, size_t workspace_bytes, void* stream);
int cpm_conv2d_backward_weight(const cpm_conv_desc* d, const float* x, const float* dy, float* dw,
                               void* workspace, size_t workspace_bytes, void* stream);
/* The same with the bias gradient of a `conv + bias` layer (nn.Conv2d(bias=True): FPN laterals / outputs FPN.py:30-50,
 * RPN predictors rpn/rpn.py:24-31, the grid head's convs grid_heads.py:41-57) folded in: dbias[k] += sum_m dy[m][k],
 * taken from the dy tiles the weight-gradient workgroups of tap 0 / input-channel tile 0 read anyway, instead of a
 * separate pass over dy (cpm_epilogue_backward).  groups with one input channel each are not covered (EINVAL). */
int cpm_conv2d_backward_weight_bias(const cpm_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                    void* workspace, size_t workspace_bytes, void* stream);
/* Data gradient with the epilogue-backward of the layer that PRODUCED x folded into its epilogue:
 *   dx[m][c] = (in_act[m][c] > 0) * in_scale[c] * conv^T(dy, w)[m][c]
 * i.e. when x = relu(prev*in_scale + shift) has this conv as its only consumer, dx is already the gradient at
 * prev's pre-activation and prev's cpm_epilogue_backward pass (a read of dy and y, a write of dpre) disappears.
 * in_act is x itself ([N,H,W,C]); in_scale [C] or NULL. */
int cpm_conv2d_backward_data_gated(const cpm_conv_desc* d, const float* dy, const float* w, float* dx,
                                   const float* in_scale, const float* in_act, void* workspace,
                                   size_t workspace_bytes, void* stream);
/* The general fused form: the frozen per-output-channel factor BEHIND this conv (y = conv(x, w) * k_scale + shift,
 * AffineChannel2d after a conv: pet/models/imagenet/resnet.py:114-136) is folded into the reduction -- dy is the
 * gradient at y and dx = gate( [dx +] conv^T(k_scale * dy, w) ) -- so no elementwise pass forms k_scale * dy; and the
 * gate of the layer that produced x may be combined with accumulation:
 *   accumulate = 0:  dx = (in_act > 0) * in_scale * conv^T(..)          (in_scale / in_act may be NULL)
 *   accumulate = 1:  dx = (in_act > 0) * (dx + conv^T(..))              (in_scale must be NULL)
 * The second form is how a ReLU'd tensor with SEVERAL consuming convs (a bottleneck output: next block's conv1, the
 * downsample conv, an FPN lateral) gets its gate applied by its consumers: masking is linear and idempotent, so every
 * consumer masks the running sum.  k_scale [K] or NULL. */
int cpm_conv2d_backward_data_fused(const cpm_conv_desc* d, const float* dy, const float* w, const float* k_scale,
                                   float* dx, int accumulate, const float* in_scale, const float* in_act,
                                   void* workspace, size_t workspace_bytes, void* stream);
/* Weight (+ bias) gradient with the same frozen factor: dw[k] += k_scale[k] * sum_m dy[m][k] x[..], dbias (may be NULL)
 * += sum_m dy[m][k] (the shift sits behind the factor).  k_scale NULL: cpm_conv2d_backward_weight(_bias). */
int cpm_conv2d_backward_weight_scaled(const cpm_conv_desc* d, const float* x, const float* dy, const float* k_scale,
                                      float* dw, float* dbias, void* workspace, size_t workspace_bytes, void* stream);

/* Data gradient with the weight ALREADY in the data-gradient image ([group][c][tap][k], what the calls above build
 * in their workspace on every call): `wt` comes from cpm_weights_to_dgrad_batched, which transforms all conv weights
 * of a flat parameter buffer in ONE launch per optimizer step (101 small launches per step otherwise).  in_scale /
 * in_act as in cpm_conv2d_backward_data_gated (both NULL: plain); with accumulate = 1 an in_act gate masks the running
 * sum as in cpm_conv2d_backward_data_fused (in_scale must then be NULL).  No workspace. */
typedef struct {
  int64_t src_off, dst_off;   /* element offsets of the KRSC weight in src_base and of its image in dst_base */
  int32_t groups, Kg, RS, Cg; /* K/groups, R*S, C/groups of the convolution the weight belongs to */
  int64_t tile_start;         /* number of 32x32 tiles (ceil(Cg/32)*ceil(Kg/32)*RS*groups) of all earlier entries */
  const float* k_scale;       /* [groups*Kg] or NULL: the image is that of diag(k_scale) * W (see _fused above) */
} cpm_wt_desc;
int cpm_weights_to_dgrad_batched(const cpm_wt_desc* d_descs /* DEVICE table, sorted by tile_start */, int n,
                                 int64_t total_tiles, const float* src_base, float* dst_base, void* stream);
int cpm_conv2d_backward_data_prepared(const cpm_conv_desc* d, const float* dy, const float* wt, float* dx,
                                      int accumulate, const float* in_scale, const float* in_act, void* workspace,
                                      size_t workspace_bytes, void* stream);
/* The images written pre-split (see cpm_split_w4: along k, four at a time) for the bf16x3 arithmetic -- entries with
 * K / groups % 4 != 0 keep their float image -- and the data gradient that reads such an image.  The data-gradient
 * entry points that build their image per call (cpm_conv2d_backward_data / _gated / _fused) do this by themselves. */
int cpm_weights_to_dgrad_batched_w4(const cpm_wt_desc* d_descs, int n, int64_t total_tiles, const float* src_base,
                                    float* dst_base, void* stream);
int cpm_conv2d_backward_data_prepared_w4(const cpm_conv_desc* d, const float* dy, const void* wt4, float* dx,
                                         int accumulate, const float* in_scale, const float* in_act, void* workspace,
                                         size_t workspace_bytes, void* stream);

/* nn.ConvTranspose2d forward (grid_rcnn/outputs.py:24-37,66-71) = the data gradient of the conv
 * described by `d` with a fused bias(+ReLU) epilogue: x [N,P,Q,K] -> y [N,H,W,C], w as for `d`
 * (torch's ConvTranspose2d weight [Cin=K][Cout/groups=C/g][R][S] permuted to KRSC).  Its own
 * backward is cpm_conv2d_forward (data) and cpm_conv2d_backward_weight with the roles of x and
 * dy swapped. */
int cpm_conv_transpose2d_forward(const cpm_conv_desc* d, const float* x, const float* w, const float* bias,
                                 int relu, float* y, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the fused epilogue: dpre = dy * (y > 0 if relu) * scale[k] (in place allowed),
 * dres = dy * mask (gradient of the residual branch; may be NULL), dshift[k] += sum_m dy*mask
 * (bias gradient; may be NULL).  M rows of K channels. */
int cpm_epilogue_backward(const float* dy, const float* y, const float* scale, int relu, int64_t M, int K,
                          float* dpre, float* dres, float* dshift, void* stream);

/* im2col for thin-channel stems (ResNet conv1 7x7/s2 on 3 channels, backbone/ResNet.py:123-126):
 * out [N*P*Q][Kpad], column (r*S+s)*C+c, zero for columns >= R*S*C.  The stem then runs as a
 * 1x1 cpm_conv2d_forward over Kpad channels with its affine+ReLU epilogue.  layout = x's layout. */
int cpm_im2col(const float* x, int layout, int N, int C, int H, int W, int R, int S, int stride, int pad,
               int P, int Q, int Kpad, float* out, void* stream);
/* nn.MaxPool2d(3, 2, 1) of the stem (backbone/ResNet.py:136), NHWC, forward only (frozen stage) */
/* Data gradient of the RPN head's two 1x1 predictors (rpn/rpn.py:24-31: cls_logits, A channels, and bbox_pred, 4A
 * channels, on the same map t) for ALL levels in one launch: dx_l = [t_l > 0] * (dy_cls_l W_cls + dy_box_l W_box) with
 * gate != 0 (t = relu(..): the predictors apply their producer's ReLU gate), the plain sum otherwise.  Maps NHWC, level
 * l has pixels[l] = N * H_l * W_l pixels; w_cls [A][C], w_box [4A][C]; C % 4 == 0, C / 4 divides 256.  Replaces 2 x
 * n_levels implicit-GEMM launches with reductions of 3 and 12 (two passes over the gradient of P2's 137 MB map). */
int cpm_rpn_pred_backward_data(const float* const* dy_cls, const float* const* dy_box, const float* const* t,
                               float* const* dx, const int64_t* pixels, int n_levels, const float* w_cls,
                               const float* w_box, int A, int C, int gate, void* stream);
/* Sparse backward of the RPN head (rpn/rpn.py:34-41) under its loss (rpn/loss.py:88-126): that loss sums over the
 * SAMPLED anchors only (balanced_positive_negative_sampler.py:27-67: 256 per image), so the gradient entering the head
 * is zero at every other anchor and the head's backward pass reduces to <= images x 256 rows of small dense matrices
 * (csrc/rpn_sparse.hip has the algebra).  Three pieces; the matrix products in between are cpm_conv2d_backward_weight*
 * / cpm_conv2d_backward_data on [1, C, rows, 1] "images":
 *   cpm_mask_compact: pos | neg (bool [total]) -> ascending positions, idx int32 [cap] (-1 behind the last), count [1];
 *     workspace: ceil(total / 4096) int32 of device scratch (per-block counts);
 *   cpm_rpn_sparse_rows: per listed anchor (flat index n * per_image + level offset + (h * W + w) * A + a, the order of
 *     concat_box_prediction_layers, rcnn/utils/misc.py:17-26): DT [cap][C] (gradient at the 3x3 conv's output, ReLU
 *     gate applied), Gc [cap][A] / Gb [cap][4A] (the predictors' output gradients), T [cap][C] (their input), X
 *     [cap][9C] (the 3x3 conv's input patch, tap-major = the KRSC weight's column order), pix4 int32 [cap][4] = (level,
 *     image, h, w) or level -1; dlog / dbox / t / feat: per-level NHWC maps [N][H_l][W_l][A | 4A | C | C];
 *     dlog_image_stride / dbox_image_stride (HOST, per level, in floats; NULL = dense): the distance between two images
 *     of dlog_l / dbox_l -- the loss hands back slices of ONE [N][all anchors][1 | 4] tensor, whose images lie a whole
 *     anchor row apart, and only <= cap elements of them are read;
 *   cpm_rpn_sparse_scatter: dfeat_l[n][h + dr][w + ds][:] += dX[row][(tap, :)] (float atomics; a NULL level is skipped).
 * Replaces, for those layers, the dense autograd path the reference takes through nn.Conv2d. */
int cpm_mask_compact(const uint8_t* pos, const uint8_t* neg, int64_t total, int cap, int32_t* idx, int32_t* count,
                     int32_t* workspace, void* stream);
int cpm_rpn_sparse_rows(const int32_t* idx, int cap, int n_img, int n_levels, const int* hs, const int* ws, int A, int C,
                        const float* const* dlog, const float* const* dbox, const float* const* t,
                        const float* const* feat, const float* w_cls, const float* w_box, float* DT, float* Gc, float* Gb,
                        float* T, float* X, int32_t* pix4, const int64_t* dlog_image_stride,
                        const int64_t* dbox_image_stride, void* stream);
int cpm_rpn_sparse_scatter(const int32_t* pix4, int cap, int n_levels, const int* hs, const int* ws, int C,
                           const float* dX, float* const* dfeat, void* stream);
/* The ResNet / ResNeXt stem as one kernel (bf16x3 arithmetic only): y = relu?(conv7x7 / stride 2 / pad 3 (x) * scale +
 * shift), x NHWC [N][H][W][3], w KRSC [64][7][7][3], y NHWC [N][P][Q][64] -- no column image (pet/models/imagenet/
 * resnet.py:175-181: conv1 + frozen bn1 + relu; the max-pool follows as cpm_maxpool3x3s2_forward). */
int cpm_stem7x7_forward(const float* x, const float* w, const float* scale, const float* shift, int relu, int N, int H,
                        int W, float* y, void* stream);
int cpm_maxpool3x3s2_forward(const float* x, int N, int H, int W, int C, int P, int Q, float* y, void* stream);

/* ---- Deformable conv v1 / narrow-group 3x3 (ResNeXt-64x4d + DCN body) -------
 * Replaces _C.deform_conv_forward / _backward_input / _backward_parameters
 * (pet/lib/ops/csrc/vision.cpp:33-38, deform_conv_cuda.cu:324-736; sampling rules
 * deform_conv_cuda_kernel.cu:95-460).  The contraction itself runs on cpm_conv2d_* as a grouped
 * 1x1 conv over the columns, whose weight is the layer's own KRSC weight:
 *   x [N,H,W,C] NHWC; offset [N,P,Q,2*R*S*deformable_groups] NHWC (channel 2*(i*S+j) = row offset,
 *   +1 = column offset) or NULL (= plain im2col, used for ResNeXt's ordinary grouped 3x3);
 *   cols / dcols [N*P*Q][groups][R*S][C/groups].
 * cpm_deform_col2im accumulates into dx (caller zero-fills); cpm_deform_coord_grad overwrites doffset. */
int cpm_deform_im2col(const float* x, const float* offset, int N, int H, int W, int C, int R, int S, int stride,
                      int pad, int dilation, int groups, int deformable_groups, int P, int Q, float* cols,
                      void* stream);
int cpm_deform_col2im(const float* dcols, const float* offset, int N, int H, int W, int C, int R, int S, int stride,
                      int pad, int dilation, int groups, int deformable_groups, int P, int Q, float* dx,
                      void* stream);
int cpm_deform_coord_grad(const float* dcols, const float* x, const float* offset, int N, int H, int W, int C,
                          int R, int S, int stride, int pad, int dilation, int groups, int deformable_groups, int P,
                          int Q, float* doffset, void* stream);

/* The same three reference entries (deform_conv_cuda.cu:324-460 forward, :463-600 input + offset gradient, :603-736
 * parameter gradient) with the sampling INSIDE the contraction -- no column matrix in memory, exact-f32 MFMA
 * arithmetic (csrc/deform_fused.hip).  Takes 3x3 layers with as many output as input channels, 4 / 8 / 16 / 32
 * channels per group, C a multiple of 64 and 64-channel slabs inside one deformable group
 * (cpm_deform_conv_fused_supported = 1; CPM_DEFORM_FUSED=0 answers 0 for everything); the entries fail with
 * CPM_EINVAL on anything else.  w [K][3][3][C/groups] (KRSC); offset NULL = plain grouped 3x3.
 *   forward:          y = relu?( conv * scale[k] + shift[k] ) (scale / shift may be NULL)
 *   backward_data:    dx += (caller zero-fills); dpre = gradient at the conv
 *   backward_params:  dw += (NULL = not wanted), doffset overwritten (NULL = not wanted; needs offset and w) */
int cpm_deform_conv_fused_supported(int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                                    int dilation, int groups, int deformable_groups, int P, int Q);
int cpm_deform_conv_forward(const float* x, const float* offset, const float* w, const float* scale,
                            const float* shift, int relu, int N, int H, int W, int C, int K, int R, int S, int stride,
                            int pad, int dilation, int groups, int deformable_groups, int P, int Q, float* y,
                            void* stream);
int cpm_deform_conv_backward_data(const float* dpre, const float* offset, const float* w, int N, int H, int W, int C,
                                  int K, int R, int S, int stride, int pad, int dilation, int groups,
                                  int deformable_groups, int P, int Q, float* dx, void* stream);
int cpm_deform_conv_backward_params(const float* dpre, const float* x, const float* offset, const float* w, int N,
                                    int H, int W, int C, int K, int R, int S, int stride, int pad, int dilation,
                                    int groups, int deformable_groups, int P, int Q, float* dw, float* doffset,
                                    void* stream);

/* ---- Detection glue of the training step ----------------------
 * Fused replacements for per-image chains of small tensor ops in the reference's Python; all images of the batch
 * in one launch.  Boxes are float4 (x1,y1,x2,y2), 16-byte aligned; gt_off [num_images+1] holds each image's
 * slice of `gts`; roi_img [R] names each RoI's image (NULL = single image).
 *
 * cpm_match_rois: boxlist_iou ("+1" widths, pet/utils/data/structures/boxlist_ops.py:123-158) + Matcher.__call__
 * (pet/rcnn/utils/matcher.py:48-111).  matched[i] = index of the best gt WITHIN its image (first maximum), -1
 * below `low`, -2 between; max_iou[i] = that IoU (may be NULL).  allow_low_quality restores the arg-max of every
 * RoI that ties some gt's best IoU (needs row_max_ws, [num_gts] ints). */
int cpm_match_rois(const float* rois, const int* roi_img, const float* gts, const int* gt_off, int R, int num_gts,
                   float high, float low, int allow_low_quality, int* row_max_ws, int64_t* matched, float* max_iou,
                   void* stream);
/* cpm_grid_bce_loss: GridLossComputation.prepare_target + loss_grid (grid_cascade_rcnn/loss.py:178-262): the
 * 0/1 point targets are rasterised on the fly (never materialised).  logits [R,points,half,half] with element
 * `strides` (host array of 4); sub_xy (host, 2*points) = x,y origin of each point's window in the whole map.
 * *loss_sum += weight * mean(BCEWithLogits);  grad (same strides) = d(weight*mean)/d logits. */
int cpm_grid_bce_loss(const float* logits, const int64_t* strides, const float* rois, const float* gt_boxes, int R,
                      int points, int map_size, const int* sub_xy, float mapping_ratio, int radius, float weight,
                      float* loss_sum, float* grad, void* stream);
/* cpm_grid_decode: GridPostProcessor.get_boxes (grid_cascade_rcnn/inference.py:189-279; no clipping, as there) and,
 * when `keep` is given, _filter_boxes (:281-290) on the UNdecoded RoIs.  */
int cpm_grid_decode(const float* logits, const int64_t* strides, const float* rois, int R, int points, int map_size,
                    const int* sub_xy, float mapping_ratio, const int* roi_img, const float* gts, const int* gt_off,
                    float* out_boxes, unsigned char* keep, void* stream);

/* cpm_rpn_decode: one FPN level of RPNPostProcessor.forward_for_single_feature_map (rpn/inference.py:67-99) after
 * the top-k: gather regression rows + anchors, BoxCoder.decode (utils/box_coder.py:51-94), clip_to_image.
 * reg [N][A][4] (NHWC head output viewed per anchor), topk_idx [N][k] int64 into A, anchors [A][4] (shared by the
 * images), weights4 / im_w / im_h HOST arrays (4 / N / N); out_boxes [N][k][4]. */
int cpm_rpn_decode(const float* reg, const int64_t* topk_idx, const float* anchors, int N, int A, int k,
                   const float* weights4, float clip, const float* im_w, const float* im_h, float* out_boxes,
                   void* stream);

/* ---- GroupNorm (+ReLU), NHWC ----------------------------------------------
 * Replaces nn.GroupNorm + nn.ReLU in grid_heads.py:47-55 and outputs.py:23,68.
 * x [N,HW,C]; mean/rstd [N,G] saved for backward; dgamma/dbeta accumulate.      */
int cpm_groupnorm_forward(const float* x, const float* gamma, const float* beta, int N, int HW, int C, int G,
                          float eps, int relu, float* y, float* mean, float* rstd, void* stream);
int cpm_groupnorm_backward(const float* dy, const float* x, const float* y, const float* gamma, const float* mean,
                           const float* rstd, int N, int HW, int C, int G, int relu, float* dx, float* dgamma,
                           float* dbeta, void* stream);
/* The same with a side job for the layer chains (cpm_layer_chain_*): the forward also leaves fill_out [N,HW,C] -- the
 * NEXT convolution's output tensor -- holding fill_bias [C] in every pixel (zeros with fill_bias == NULL); the backward
 * also clears zero_out [N,HW,C] -- the convolution's INPUT gradient.  With cpm_conv_next_output_prepared the
 * reduction-split launch that follows adds into the tensor without a seed / clear launch of its own. */
int cpm_groupnorm_forward_fill(const float* x, const float* gamma, const float* beta, int N, int HW, int C, int G,
                               float eps, int relu, float* y, float* mean, float* rstd, float* fill_out,
                               const float* fill_bias, void* stream);
int cpm_groupnorm_backward_zero(const float* dy, const float* x, const float* y, const float* gamma, const float* mean,
                                const float* rstd, int N, int HW, int C, int G, int relu, float* dx, float* dgamma,
                                float* dbeta, float* zero_out, void* stream);
/* Tells the NEXT cpm_conv2d_forward* / cpm_conv2d_backward_data* call of this thread that its output tensor already
 * holds what a reduction-split launch starts from -- the bias in every pixel for a forward whose epilogue is a bias (or
 * nothing: zeros), zeros for a non-accumulating data gradient.  Consumed by that call, whatever path it takes. */
void cpm_conv_next_output_prepared(int on);

/* FPN top-down backward (FPN.py:104-106): dtop[n,p,q,c] += sum of dy over the 2x2 block it was
 * upsampled to.  dy [N,P,Q,C], dtop [N,ceil(P/2),ceil(Q/2),C]. */
int cpm_upsample2x_add_backward(const float* dy, int N, int P, int Q, int C, float* dtop, int accumulate,
                                void* stream);

/* ---- soft-NMS on the device, batched over segments (classes) ----------------------------
 * Replaces pet/lib/ops/nms.py:16-28 -> csrc/NMS/soft_nms.cpp:5-160 (a CPU kernel called per class after a device->host
 * copy, boxlist_ops.py:86-90) and, with `labels`, nms.py:31-45 -> csrc/NMS/ml_soft_nms.cpp:5-186 (only boxes carrying
 * the selected box's label decay; the selection stops after `topk` picks: 0 keeps nothing, < 0 never stops -- the
 * reference's `topk == i` rule; topk is ignored without labels).  Segment p = rows [h_offsets[p], h_offsets[p+1])
 * (HOST offsets, <= 64 segments, <= 2048 boxes each).  method: 0 hard, 1 linear, 2 gaussian (SOFT_NMS_METHODS).
 * Results of segment p are written at its own offset, in the reference's output order: out_counts[p] boxes with their
 * DECAYED scores, their labels (out_labels may be NULL) and their index inside the segment.  Linear / hard:
 * bit-identical to the reference; gaussian: device expf. */
int cpm_soft_nms_batched(const float* boxes, const float* scores, const int64_t* labels, const int32_t* h_offsets,
                         int P, float iou_threshold, int method, float sigma, float min_score, int topk,
                         float* out_boxes, float* out_scores, int64_t* out_labels, int64_t* out_idx,
                         int32_t* out_counts, void* stream);

/* ---- bounding-box voting ---------------------------------------------------------------------
 * Replaces _C.box_voting (pet/lib/ops/boxes.py:6-22 -> csrc/Box_ops/box_voting.cu:24-210): every top box becomes the
 * score-weighted mean of the candidate boxes with IoU >= threshold; scoring_method 0 ID, 1 TEMP_AVG, 2 AVG,
 * 3 IOU_AVG, 4 GENERALIZED_AVG, 5 QUASI_SUM re-estimates its score.  boxes [N,4] / scores [N]: top detections;
 * query_* [K,4] / [K]: all detections of the class.  With labels / query_labels (int64; _C.box_ml_voting, boxes.py:25-45 ->
 * box_ml_voting.cu) only candidates carrying the top box's label vote.  One launch, no [N,K,7] intermediate. */
int cpm_box_voting(const float* boxes, const float* scores, const int64_t* labels /* NULL: single label */, int N,
                   const float* query_boxes, const float* query_scores, const int64_t* query_labels, int K,
                   int scoring_method, float beta, float threshold, float* out_boxes, float* out_scores, void* stream);

/* RPN loss after matching and sampling (pet/rcnn/modeling/rpn/loss.py:60-141): BoxCoder.encode of the matched gt vs the
 * anchor, smooth-L1 (beta) over the sampled positives, BCE-with-logits over the sampled anchors -- values AND gradients
 * in one pass over all `total` = images x per_image anchors.  logits [total], reg / anchors [total,4], matched [total]
 * (gt index inside the image, < 0: none), gts [G,4] with gt_off [images+1] (device), pos / neg [total] bool masks.
 * sums2[0] = sum of the BCE terms, sums2[1] = sum of the smooth-L1 terms;
 * dlogits [total], dreg [total,4] = derivatives of those sums (zero outside the sample).
 * quota (device, n_quota int32; NULL: sums and derivatives stay undivided): the sampler's per-image (positive, negative)
 * counts -- sums and derivatives leave divided by their total, the normalisation of loss.py:121-126. */
int cpm_rpn_loss(const float* logits, const float* reg, const float* anchors, const int64_t* matched, const float* gts,
                 const int* gt_off, const uint8_t* pos, const uint8_t* neg, int64_t total, int per_image,
                 const float* weights4, float beta, float* sums2, float* dlogits, float* dreg, const int32_t* quota,
                 int n_quota, void* stream);

/* ---- row-wise top-k for the RPN proposal selection --------------------------------------
 * Replaces `objectness.topk(pre_nms_top_n, dim=1, sorted=True)` of pet/rcnn/modeling/rpn/inference.py:79-84 (torch's
 * multi-kernel radix top-k, one call per FPN level).  scores [rows][n] fp32; for every row the k largest values in
 * descending order with their column indices (int64); equal values are ordered by ascending index (torch leaves
 * that order unspecified).  1 <= k <= min(n, 2048).  NaNs rank above every number, as in torch. */
int cpm_topk_rows(const float* scores, int rows, int n, int k, float* out_scores, int64_t* out_idx, void* stream);
/* The same for several matrices of `rows` rows each in ONE launch (the RPN's five FPN levels, inference.py:67-114 loops
 * over them): HOST arrays of `levels` <= 8 device pointers / sizes; level l: scores[l] [rows][n[l]] -> out_scores[l],
 * out_idx[l] [rows][k[l]].
 * workspace (device, 8-byte aligned, cpm_topk_rows_multi_workspace_bytes; may be NULL): with it, rows of 32 768 elements
 * and more are selected by up to 8 workgroups each (slices of the row, then a merge launch) instead of one -- the same
 * result, the order of equal values included. */
size_t cpm_topk_rows_multi_workspace_bytes(int levels, int rows);
int cpm_topk_rows_multi(const float* const* scores, const int* n, const int* k, int levels, int rows,
                        float* const* out_scores, int64_t* const* out_idx, void* workspace, size_t workspace_bytes,
                        void* stream);

/* ---- fixed-size positive / negative sampling for the whole batch ---------------------------
 * Replaces BalancedPositiveNegativeSampler.__call__ (pet/rcnn/utils/balanced_positive_negative_sampler.py:27-67, called
 * from pet/rcnn/modeling/rpn/loss.py:102 and pet/rcnn/modeling/grid_cascade_rcnn/loss.py:77), which runs per image
 * nonzero() + torch.randperm on the host's schedule.  labels [total] (label_dtype 0 float32 / 1 int64 / 2 int32;
 * >= 1 positive, 0 negative, < 0 ignored), image i owns [h_offsets[i], h_offsets[i+1]) (HOST array, images <= 64).
 * Per image: n_pos = min(#positives, max_pos), n_neg = min(#negatives, batch_size_per_image - n_pos), each a uniformly
 * random subset of its class (keys from a counter hash of (seed, image, index); the same seed gives the same sample).
 * pos / neg [total] receive 0/1 for EVERY element; out_quota [images][2] = (n_pos, n_neg).  Sample sizes up to 1024 per
 * class go through a short candidate list; larger ones through a one-workgroup radix select (slower, same distribution).
 * cand_target: expected length of the per-class short list the selection works on (0 = 2 * quota + 64); any value
 * yields exact sample sizes -- small values only exercise the index-order fill path (tests).
 * workspace: cpm_sample_pos_neg_workspace_bytes() device bytes, contents irrelevant. */
size_t cpm_sample_pos_neg_workspace_bytes(void);
int cpm_sample_pos_neg(const void* labels, int label_dtype, const int64_t* h_offsets, int n_img,
                       int batch_size_per_image, int max_pos, uint64_t seed, int cand_target, uint8_t* pos, uint8_t* neg,
                       int32_t* out_quota, void* workspace, void* stream);

/* ---- image preparation on the device ---------------------------------
 * Replaces, per image, the host transform chain of pet/rcnn/datasets/transform.py:6-50:
 *   Resize (pet/utils/data/transforms/transforms.py:29-64 -> PIL.Image.resize(BILINEAR), i.e. Pillow's
 *   ImagingResample: a horizontal then a vertical uint8 pass with Q22 fixed-point taps),
 *   RandomHorizontalFlip (:67-77), ToTensor (:99-101), Normalize incl. to_bgr255 (:104-115)
 * and the zero padding of to_image_list (pet/utils/data/structures/image_list.py:56-66).
 * src: decoded RGB image [H][W][3] uint8 on the device.  hbounds [ow][2] (first source column, tap count) and
 * hcoef [ow][hksize] int32 Q22 taps (both NULL when ow == W: Pillow skips the pass); vbounds / vcoef likewise
 * over rows (NULL when oh == H).  The caller computes the tables in double precision as Pillow's
 * precompute_coeffs + normalize_coeffs_8bpc do.  flip: mirror the RESIZED image.  lut [3][256] fp32: value of
 * OUTPUT channel o for byte v, i.e. ((v/255)*255 - mean[o]) / std[o] evaluated in fp32; swap_rb: output
 * channel o reads input channel 2-o (image[[2,1,0]]).  tmp: >= H*ow*3 bytes of scratch (horizontal result).
 * dst: one slot of the batch tensor, [dstH][dstW][3] (layout 1, NHWC) or [3][dstH][dstW] (layout 0); every
 * element of the slot is written (zeros outside oh x ow).  Results are bit-identical to the host chain. */
int cpm_image_prep(const uint8_t* src, int H, int W, const int32_t* hbounds, const int32_t* hcoef, int hksize,
                   const int32_t* vbounds, const int32_t* vcoef, int vksize, int oh, int ow, int flip,
                   const float* lut, int swap_rb, uint8_t* tmp, float* dst, int dstH, int dstW, int layout,
                   void* stream);

/* Test-time resize of pet/rcnn/core/test.py:340-358 (get_blob): uint8 [H][W][3] -> fp32 [3][oh][ow] by plain bilinear
 * interpolation with half-pixel centres on float values (cv2.resize(im.astype(float32), None, None, fx, fy,
 * INTER_LINEAR)); inv_fx / inv_fy = 1/fx, 1/fy; flip mirrors the SOURCE columns first (im[:, ::-1, :]); swap_rb writes
 * channel c to plane 2-c (the reference reads BGR with cv2, the loader here decodes RGB).  No normalisation: the
 * model's Norm layer does that at test time (model_builder.py:25-30). */
int cpm_image_resize_linear(const uint8_t* src, int H, int W, int oh, int ow, float inv_fx, float inv_fy, int flip,
                            int swap_rb, float* dst, void* stream);

/* ---- RPN proposal stage, all FPN levels per launch --------------------------------------------
 * cpm_sigmoid_multi: objectness.sigmoid() (pet/rcnn/modeling/rpn/inference.py:72) of `levels` (<= 8) logit arrays of
 *   n[l] elements into out + out_off[l]; evaluates 1 / (1 + expf(-x)) like the framework kernel (same bits, same ties).
 * cpm_rpn_decode_multi: cpm_rpn_decode for every level at once; level l's N x k[l] boxes land at out_boxes + out_off[l]
 *   rows (the level-major segment layout cpm_nms_batched reads).
 * cpm_rpn_labels: anchor labels of RPNLossComputation.prepare_targets (rpn/loss.py:60-79) from the match: 1 matched,
 *   0 below the low threshold, -1 between thresholds (if discard_between) or not visible (visible may be NULL). */
int cpm_sigmoid_multi(const float* const* in, const int* n, const int* out_off, int levels, float* out, void* stream);
int cpm_rpn_decode_multi(const float* const* reg, const int64_t* const* topk_idx, const float* const* anchors,
                         const int* A, const int* k, const int* out_off, int levels, int N, const float* weights4,
                         float clip, const float* im_w, const float* im_h, float* out_boxes, void* stream);
int cpm_rpn_labels(const int64_t* matched, const uint8_t* visible, int64_t total, int discard_between, float* labels,
                   void* stream);

/* ISM loss: l2_loss (pet/lib/ops/l2_loss.py:4-11) of x [R, 2] against target [R, 2] -- or, with target NULL, against
 * (1 - iou[r], iou[r]) (GridLossComputation.prepare_iou_target, grid_cascade_rcnn/loss.py:164-176).  The reference's
 * x[pos_inds] gathers, for every positive target entry (r, c), rows r AND c:
 *   loss = (sum_r cnt[r] E[r] + n_col0 E[0] + n_col1 E[1]) / P,  E[i] = 0.5 sum_j (x[i,j] - t[i,j])^2.
 * *loss and grad [R, 2] (d loss / d x) from one launch; R >= 2. */
int cpm_l2_loss_pairs(const float* x, const float* iou, const float* target, int R, float* loss, float* grad,
                      void* stream);

/* RoI counts to the host without a copy command (the training step's three count reads, SURVEY 8f-1: the reference
 * reads its counts through nonzero() / .item() on every BoxList operation).  cpm_host_device_pointer: the device's
 * address of a pinned host allocation.  cpm_publish_counts: one launch stores counts[0..n) and then `seq` at index n of
 * the mapped buffer, system-scope release between; the host polls host_mapped[n] == seq. */
int cpm_host_device_pointer(void* host_pinned, void** out);
int cpm_publish_counts(const int32_t* counts, int n, int32_t* host_mapped, int32_t seq, void* stream);

/* Classification loss of the cls and RSM heads: F.cross_entropy(logits [R, C], labels [R]) with the default mean
 * reduction and ignore_index (CLSLossComputation.__call__, pet/rcnn/modeling/grid_cascade_rcnn/loss.py:103-112;
 * cascade_rcnn/loss.py:63) = log_softmax + nll_loss.  *loss and grad [R, C] (d loss / d logits) from one launch; rows
 * labelled ignore_index add nothing and get a zero gradient; labels outside [0, C) other than ignore_index are the
 * caller's error (as in the framework: not checked on the device).  fp32, rows contiguous.  row_loss: R floats of
 * scratch; ticket: one int that is ZERO before the first call and left zero by every call (calls that share it must
 * be ordered on one stream).  The sum over the rows is taken in a fixed order. */
int cpm_softmax_ce(const float* logits, const int64_t* labels, int R, int C, int64_t ignore_index, float* loss,
                   float* grad, float* row_loss, int* ticket, void* stream);

/* ---- device-resident RoI lists of the training step -----------------------------------------
 * Packed lists with a fixed capacity and a per-image count ON THE DEVICE replace the reference's per-image BoxList
 * surgery (nonzero / boolean index / randperm / cat, each a launch and a device->host round trip).  Every call is one
 * launch of one workgroup; counts arrays hold [images] per-image rows followed by the total; rois5 outputs are
 * [capacity][5] = (image, x1, y1, x2, y2), the RoIAlign input format (pet/rcnn/utils/poolers.py:74-85).
 *
 * cpm_proposals_finalize: RPNPostProcessor.forward_for_single_feature_map's post-NMS top-n, select_over_all_levels
 *   (training: ONE top-k over the batch) and add_gt_proposals (pet/rcnn/modeling/rpn/inference.py:101-196).  Inputs
 *   are cpm_nms_batched's: seg_boxes / seg_scores [S] rows of all (level, image) segments (level-major; h_seg_off HOST
 *   [levels*images+1]), keep (segment-relative kept rows, best first) and keep_count.  Of the candidates in reference
 *   order (image, level, rank) the batch_top_k best scores are kept -- ties at the boundary by lowest index, as
 *   torch.topk does -- in that order, each image followed by its gts (objectness 1).  Rows >= total: img -1.
 * cpm_roi_sample: CLSLossComputation.subsample (grid_cascade_rcnn/loss.py:29-97): IoU(+1) match against the image's
 *   gts, Matcher(high, low) labels (gt label / 0 / -1), BalancedPositiveNegativeSampler(batch, max_pos) -- the draw of
 *   cpm_sample_pos_neg for the same seed -- and the sample compacted in input order.  max_grid > 0 also lists the
 *   sample's positives, at most max_grid per image (keep_only_positive_boxes, pet/rcnn/utils/misc.py:54-94; a random
 *   subset drawn from seed_grid), p_src = row in the sample, p_iou = best IoU, p_gt = the box of the best gt if that
 *   IoU >= grid_high, else of the image's first gt (GridLossComputation.subsample at stage 0, loss.py:144-162:
 *   t.bbox[matched.clamp(min=0)]).  Images may hold at most cpm_roi_sample_max_rows() rows;
 *   otherwise *status = 1 and nothing is written.  Padding rows: s_img -1, s_labels -100 (cross_entropy ignore_index).
 * cpm_stage_advance: GridPostProcessor's training filter + the next stage's subsample
 *   (grid_cascade_rcnn/inference.py:281-310, loss.py:144-176): rows with keep != 0 and matched >= 0 in order, then the
 *   image's gts; o_gt = the matched gt box (a gt matches itself, IoU 1), o_src = ride-along row (gt g: gt_src_base+g).
 * cpm_rescore_gather: get_full_sample_boxes (grid_cascade_rcnn.py:231-245): per image the cls sample's rows with
 *   label <= 0, then the last stage's rows; a last-stage row's objectness is followed through its ride-along index
 *   g_src: < n_first -> the cls sample row p_src[g_src], otherwise an appended gt (objectness 1). */
int cpm_proposals_finalize(const float* seg_boxes, const float* seg_scores, const int64_t* keep,
                           const int32_t* keep_count, const int32_t* h_seg_off, int n_images, int n_levels,
                           int post_nms_top_n, int batch_top_k, const float* gts, const int32_t* gt_off, int capacity,
                           float* out_boxes, float* out_obj, float* out_rois5 /* may be NULL */, int32_t* out_img,
                           int32_t* out_counts, void* stream);
int cpm_roi_sample_max_rows(void);
int cpm_roi_sample(const float* boxes, const float* obj, const int32_t* counts, int n_images, const float* gts,
                   const int64_t* gt_labels, const int32_t* gt_off, float high, float low, int batch_size_per_image,
                   int max_pos, uint64_t seed, int max_grid, uint64_t seed_grid, float grid_high, int cap_sample,
                   float* s_boxes, float* s_obj, int64_t* s_labels, int32_t* s_img, float* s_rois5, int32_t* s_counts,
                   int cap_grid, float* p_boxes, float* p_gt, float* p_iou, int64_t* p_src, int32_t* p_img,
                   float* p_rois5, int32_t* p_counts, int32_t* status, void* stream);
int cpm_stage_advance(const float* refined, const uint8_t* keep, const int64_t* matched, const float* iou,
                      const int32_t* img, const int64_t* src /* NULL: row index */, int R, int n_images,
                      int64_t gt_src_base,
                      const float* gts, const int32_t* gt_off, int capacity, float* o_rois, float* o_gt, float* o_iou,
                      int64_t* o_src, int32_t* o_img, float* o_rois5, int32_t* o_counts, void* stream);
int cpm_rescore_gather(const float* s_boxes, const float* s_obj, const int64_t* s_labels, const int32_t* s_counts,
                       const float* g_boxes, const int64_t* g_src, const int32_t* g_counts, const int64_t* p_src,
                       int n_first, int n_images, int capacity, float* o_boxes, float* o_obj, int32_t* o_counts,
                       void* stream);

/* Stream ordering for work the host side forks onto a second stream (the weight-gradient kernels of the backward
 * pass run beside the data-gradient chain): stream `to` waits for everything queued on `from` so far.  The events come
 * from a ring per device (the current device's: both streams must belong to it); safe to call from the autograd
 * engine's per-device worker threads. */
int cpm_stream_fork(void* from, void* to);
/* A stream whose kernels stay off `reserve_cus` compute units (hipExtStreamCreateWithCUMask); *out is a hipStream_t the
 * caller owns.  No reference counterpart: DistributedDataParallel (tools/rcnn/train_net.py:134-136) leaves the overlap of
 * NCCL kernels with the backward pass to the device scheduler; here the weight-gradient stream can leave CUs to RCCL
 * (pet/utils/parallel.py, CPM_WGRAD_RESERVE_CUS). */
int cpm_stream_create_cu_reserve(int reserve_cus, void** out);
int cpm_stream_destroy(void* stream);

/* ---- a chain of RoI-head layers (conv + bias -> [GroupNorm] -> [ReLU]) from one call --------------------------------
 * The CMM grid head (8 x conv3x3 -> GroupNorm -> ReLU, pet/rcnn/modeling/grid_rcnn/heads/grid_heads.py:41-57,146-152),
 * the cls / RSM head (fc6 -> ReLU -> fc7 -> ReLU -> cls_score, heads/cls_heads.py:13-48 + outputs.py:87-104) and the
 * ISM branch (outputs.py:38-45,76-83) on a few dozen RoIs are launch bound; these entry points run the per-layer C-ABI
 * calls above (cpm_conv2d_forward, cpm_groupnorm_forward / _backward, cpm_conv2d_backward_weight[_bias],
 * cpm_conv2d_backward_data[_gated / _prepared]) in a loop, same order and arguments as a caller going layer by layer.
 * A Linear is a 1x1 conv on a 1x1 image, fc6 / iou_fc1 a full-window conv.  The layer table holds what does not
 * change from call to call (geometry per sample: conv.N is ignored; parameter and gradient-sink pointers), so a caller
 * builds it ONCE; the number of samples N (RoIs) comes with every call.  relu: with has_gn the GroupNorm kernel applies
 * it; without, the conv epilogue does and the NEXT layer's data gradient undoes it (gate on its input), so the last
 * layer of a chain must not end in a bare ReLU.  All per-call tensors live in two caller-owned buffers whose sizes
 * cpm_layer_chain_sizes reports for a given N (pieces back to back, 256-byte aligned):
 *   fwd_base: conv_out [N,P,Q,K] (+ gn_out, mean / rstd [N, gn_groups] with has_gn) of every layer (written by
 *             forward, read by backward); the LAST layer's output is the separate tensor y;
 *   bwd_base: d_conv [N,P,Q,K] (has_gn only) and d_in [N,H,W,C] (gradient at the layer's input) of every layer; layer
 *             0's d_in is the separate tensor dx (NULL skips that data gradient);
 *   dw / dbias / dgamma / dbeta: gradient sinks, ACCUMULATED into (dbias may be NULL); wt: the weight's data-gradient
 *             image or NULL (then w is used).
 * backward: `side_stream` (NULL = none) receives the weight-gradient launches, forked per layer; the caller joins it.
 * workspace (and side_workspace) >= the workspace_bytes reported for N. */
typedef struct {
  cpm_conv_desc conv;
  const float* w;
  const float* wt;
  const float* bias;
  const float* gamma;
  const float* beta;
  float* dw;
  float* dbias;
  float* dgamma;
  float* dbeta;
  int has_gn;
  int relu;
  int gn_groups;
  float eps;
  int dgrad_flat;   /* full-window layer: data gradient as the [N,K] x [K, R*S*C] GEMM it is (wt = that matrix's image) */
  int wt_w4;        /* wt is a pre-split image (cpm_weights_to_dgrad_batched_w4): read under the bf16x3 arithmetic only */
  const void* w4;   /* the pre-split image of w (cpm_split_w4 / cpm_sgd_step_w4) or NULL: forward under bf16x3 reads it */
} cpm_chain_layer;
int cpm_layer_chain_sizes(const cpm_chain_layer* layers, int n_layers, int N, size_t* fwd_floats, size_t* bwd_floats,
                          size_t* workspace_bytes);
int cpm_layer_chain_forward(const cpm_chain_layer* layers, int n_layers, int N, const float* x, float* fwd_base,
                            float* y, void* workspace, size_t workspace_bytes, void* stream);
int cpm_layer_chain_backward(const cpm_chain_layer* layers, int n_layers, int N, const float* x, const float* dy,
                             float* fwd_base, float* y, float* bwd_base, float* dx, void* workspace,
                             size_t workspace_bytes, void* side_workspace, size_t side_workspace_bytes, void* stream,
                             void* side_stream);

/* ---- measurement hooks (bench.py) ----------------------------------------------
 * cpm_prof_enable(1) brackets every conv kernel launch with HIP events on its own stream and
 * remembers the launch's ALGORITHMIC flops (2*N*P*Q*K*R*S*C/groups); cpm_prof_enable(0) stops and
 * clears.  cpm_prof_summary sums the event durations per kernel kind: 0 = igemm forward-gather,
 * 1 = igemm data-gradient-gather, 2 = weight gradient.  Not thread safe; off on the hot path. */
int cpm_prof_enable(int on);
int cpm_prof_summary(int kind, double* total_ms, double* total_flops, int64_t* launches);
/* per-launch CSV (kind, conv dims, algorithmic GFLOP, HIP-event ms) of everything recorded since cpm_prof_enable(1) */
int cpm_prof_dump(const char* path);

/* ---- fused SGD with momentum over a flat parameter buffer -------------------
 * Replaces torch.optim.SGD.step as built by pet/utils/optimizer.py:40-65 (3 param
 * groups: weights / biases (lr x2, no wd) / GN).  The flat buffer is cut into segments
 * that start on 64-element boundaries: block_seg[i/64] (device int32) is the segment of
 * element i or -1 in an alignment gap; seg_end / seg_group (device, per segment) give its
 * end and parameter group; h_group_lr / h_group_wd are HOST arrays of `ngroups` (<= 8)
 * values passed by value to the kernel (the schedule changes them every iteration).
 * d = g*grad_scale + wd*p; buf = momentum*buf + d (buf = d on the first step); p -= lr*buf
 * (torch.optim.SGD, dampening 0).                                                      */
int cpm_sgd_step(float* params, const float* grads, float* momentum_buf, const int32_t* block_seg,
                 const int64_t* seg_end, const int32_t* seg_group, const float* h_group_lr,
                 const float* h_group_wd, int ngroups, int64_t total, float momentum, float grad_scale,
                 int first_step, void* stream);
/* The same pass also writing the pre-split image (cpm_split_w4's format) of the UPDATED parameters into `w4_out`
 * (`total` floats' worth of bytes, the parameters' own offsets): the weight operand of the next step's forward convs
 * (cpm_conv2d_forward_w4) without a pass of its own. */
int cpm_sgd_step_w4(float* params, const float* grads, float* momentum_buf, const int32_t* block_seg,
                    const int64_t* seg_end, const int32_t* seg_group, const float* h_group_lr,
                    const float* h_group_wd, int ngroups, int64_t total, float momentum, float grad_scale,
                    int first_step, void* w4_out, void* stream);
/* The update of elements [begin, end) only (64-element boundaries; the pointers are those of the whole buffers, w4_out
 * may be NULL): a trainer updates a chunk of the flat buffer as soon as its gradients are complete (and all-reduced), on
 * a side stream beside the rest of the backward pass (pet/utils/parallel.py).  zero_grads != 0 clears every gradient
 * element behind its use: the next step's zero_grad (a 614 MB memset for R-50, 0.3 ms) without a pass of its own. */
int cpm_sgd_step_range(float* params, float* grads, float* momentum_buf, const int32_t* block_seg,
                       const int64_t* seg_end, const int32_t* seg_group, const float* h_group_lr,
                       const float* h_group_wd, int ngroups, int64_t begin, int64_t end, float momentum, float grad_scale,
                       int first_step, void* w4_out, int zero_grads, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CPMRCNN_HIP_H */
