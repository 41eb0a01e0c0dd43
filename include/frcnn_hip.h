/*
 * frcnn_hip.h — C ABI of libfrcnn_hip.so, the gfx950 (MI355X) HIP library behind the
 * Faster R-CNN hot path of mathild7/faster_rcnn_pytorch_multimodal.
 *
 * The reference has no FFI of its own (it is 100 % Python on torch / torchvision); each entry
 * point below names the reference call it replaces (path:line under the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch allocates), unless marked host;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), makes no allocation and
 *     no host synchronisation, so a whole frame can be captured in one hipGraph;
 *   - activations are NHWC fp32, filters are KRSC fp32 (K = output channels), both with C % 4 == 0;
 *   - scratch memory comes from the caller: `*_ws_bytes()` says how much `(ws, ws_bytes)` must hold;
 *   - return value 0 = ok, negative = error (frcnn_last_error() gives the message of the calling
 *     thread's last failure). Counts produced on the device (NMS survivors ...) stay on the device.
 */
#ifndef FRCNN_HIP_H_
#define FRCNN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRCNN_OK 0
#define FRCNN_ERR_ARG (-1)     /* bad shape / null pointer / unsupported configuration */
#define FRCNN_ERR_WS (-2)      /* workspace too small */
#define FRCNN_ERR_LAUNCH (-3)  /* hipLaunch / runtime error */

/* Library version (major*10000 + minor*100 + patch) and last error text of this thread. */
int frcnn_version(void);
/* How the library initialises / copies device memory inside its launch sequences: 0 (default) = its own fill / copy
 * KERNELS, so that a stream capture of any entry point holds kernel nodes only; 1 = hipMemsetAsync / hipMemcpyAsync
 * (memset / memcpy graph nodes).  See DESIGN.md section 4.8 for why the default is 0.  Process-wide, read at call time. */
int frcnn_set_memops_mode(int mode);
int frcnn_get_memops_mode(void);
/* Hash of the CURRENT VALUES of every process-wide switch that selects kernels (frcnn_set_memops_mode, frcnn_conv2d_set_tile /
 * _set_algo / _set_staging, frcnn_roi_align_set_variant, frcnn_filter_set_variant, frcnn_nms_set_suppress_at_equal): a caller
 * that holds captured hipGraphs of entry points of this library (the reference's per-frame loop lib/model/test.py:183-228
 * replayed by model/frame_graph.FramePool) keys its captures by it - another value means the captures describe other kernels. */
unsigned frcnn_settings_signature(void);
const char* frcnn_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Dense backbone: conv + folded BN + residual + ReLU  (lib/nets/resnet.py:98-127 Bottleneck.forward,
 * :24-32 conv3x3/conv1x1, :152-156 stem; lib/nets/fpn.py:33-39 lateral/smoothing convs; RPN convs of
 * the missing lib/nets/network.py).  Implicit GEMM on v_mfma_f32_32x32x2_f32:
 *     y[n,ho,wo,k] = act( (sum_{r,s,c} x[n,ho*stride-pad+r,wo*stride-pad+s,c] * w[k,r,s,c]) * scale[k]
 *                          + shift[k] + residual[n,ho,wo,k] )
 * scale/shift/residual may be NULL (1, 0, 0). relu != 0 applies max(.,0).
 * split_k == 0 lets the library choose; >= 1 forces that many K splits (partials go through ws).
 * ------------------------------------------------------------------------------------------- */
size_t frcnn_conv2d_fwd_ws_bytes(int n, int h, int w, int c, int k, int r, int s, int stride, int pad,
                                 int split_k);
int frcnn_conv2d_fwd(const float* x, const float* wgt, const float* scale, const float* shift,
                     const float* residual, float* y, int n, int h, int w, int c, int k, int r, int s,
                     int stride, int pad, int relu, int split_k, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Convolution backward (autograd of the same nn.Conv2d call sites; lib/model/train_val.py:458 train_step ->
 * loss.backward()).  Gradients w.r.t. the folded-BN scale/shift are not produced: BatchNorm is frozen on
 * this path (lib/nets/imagenet.py:110-116).
 *
 * frcnn_conv2d_transpose_filter: w (K,R,S,C) -> w_t (C,R,S,K) with the taps flipped, the filter of the
 *   data-gradient convolution (prepare once per weight version).
 * frcnn_conv2d_bwd_data: dx (n,h,w,c) = conv_transpose(dy (n,ho,wo,k), w) [+ add (n,h,w,c), may be NULL].
 *   n,h,w,c,k,r,s,stride,pad describe the FORWARD convolution.  Needs k % 4 == 0, r == s, pad <= r-1.
 * frcnn_conv2d_bwd_weight: dw (K,R,S,C) = sum_pixels dy x patch(x); db (K) = sum_pixels dy (db may be NULL).
 * ------------------------------------------------------------------------------------------- */
int frcnn_conv2d_transpose_filter(const float* w_krsc, float* w_crsk_flipped, int k, int r, int s, int c,
                                  void* stream);
size_t frcnn_conv2d_bwd_data_ws_bytes(int n, int h, int w, int c, int k, int r, int s, int stride, int pad);
int frcnn_conv2d_bwd_data(const float* dy, const float* w_crsk_flipped, const float* add, float* dx, int n, int h,
                          int w, int c, int k, int r, int s, int stride, int pad, void* ws, size_t ws_bytes,
                          void* stream);
/* frcnn_conv2d_bwd_data with the Winograd transform of the data-gradient filter supplied by the caller: w_winograd =
 * frcnn_conv2d_winograd_filter(w_crsk_flipped viewed as a (c,3,3,k) filter), (16,c,k), or NULL.  The weights change once per
 * optimizer step (every cfg.TRAIN.BATCH_SIZE frames, lib/model/train_val.py:379-382), so a caller transforms once per step
 * instead of once per frame.  Only for 3x3 / stride 1 / pad 1 layers and add == NULL. */
int frcnn_conv2d_bwd_data_pre(const float* dy, const float* w_crsk_flipped, const float* w_winograd, const float* add,
                              float* dx, int n, int h, int w, int c, int k, int r, int s, int stride, int pad, void* ws,
                              size_t ws_bytes, void* stream);
/* ... and with the activation backward of the layer BELOW folded into the store: dx = act_y > 0 ? dx * act_scale[c] : 0, i.e.
 * what frcnn_act_bwd(dx, act_y, act_scale, relu = 1) would make of the result in a second pass (act_y = that layer's ReLU
 * output, same shape as dx; act_scale = its folded BatchNorm scale or NULL).  lib/nets/resnet.py:98-127 backward: the data
 * gradient of conv3 directly yields the gradient of conv2's pre-activation output, that of conv2 the one of conv1's.  Not for
 * the strided 1x1 form. */
int frcnn_conv2d_bwd_data_act(const float* dy, const float* w_crsk_flipped, const float* w_winograd, const float* add,
                              const float* act_y, const float* act_scale, float* dx, int n, int h, int w, int c, int k, int r,
                              int s, int stride, int pad, void* ws, size_t ws_bytes, void* stream);
size_t frcnn_conv2d_bwd_weight_ws_bytes(int n, int h, int w, int c, int k, int r, int s, int stride, int pad);
/* counters: frcnn_conv2d_bwd_weight_counters(c, k, r, s) ints of device memory that are ZERO on entry and are left zero - one
 * per output tile: the workgroups that share a tile's pixel-split sums count themselves there and the last one adds the
 * slabs (in split order: deterministic) and writes / accumulates the result, so no second kernel runs.  Two launches that may
 * be in flight at the same time must not share counters (give each launch of a captured sequence its own range; launches that
 * are ordered on one stream may reuse a range).  NULL: only plans that need no counters are used (the register-staged kernel
 * with its reduction kernels, or an unsplit LDS-DMA launch). */
int frcnn_conv2d_bwd_weight_counters(int c, int k, int r, int s);
int frcnn_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* db, int n, int h, int w, int c, int k,
                            int r, int s, int stride, int pad, void* ws, size_t ws_bytes, int* counters, void* stream);
/* Which filter-gradient kernels may be planned: 0 = all (default), 1 = conv_wgrad_f32 (register-staged, separate reduction /
 * accumulation kernels) only, 2 = conv_wgrad_dma_f32 (LDS-DMA ring, fused reduction + accumulation) wherever it applies
 * (operands < 2 GB).  Forgets the tuned filter-gradient plans.  A/B switch for tests and benchmarks. */
int frcnn_conv2d_wgrad_set_variant(int variant);
/* Force the filter-gradient plan of every following call (tests): kernel 1 / 2 = conv_wgrad_f32 with the 64 / 128 tile,
 * 3 / 4 = conv_wgrad_dma_f32 with the 64 / 128 tile; `splits` pixel ranges (clamped to the number of 32-pixel steps).
 * kernel 0 returns to the tuned / modelled plans.  A call the forced kernel cannot serve fails with FRCNN_ERR_ARG. */
int frcnn_conv2d_wgrad_set_plan(int kernel, int splits);

/* The same filter gradient ACCUMULATED into a parameter's own gradient buffer: grad_w (K, c_real, R, S) += dw (the layout of
 * nn.Conv2d.weight; nn.Linear.weight with R = S = 1), c_real <= c real input channels (c is the padded count of x),
 * grad_b (K) += db (may be NULL).  One pass sums the pixel-split slabs, drops the channel padding, changes the layout and
 * adds - instead of a reduction launch, a permute copy and an add per parameter; with `counters` (see frcnn_conv2d_bwd_weight)
 * that pass is the epilogue of the filter-gradient kernel itself.  Uses the plan frcnn_conv2d_bwd_weight tuned for the shape;
 * workspace of frcnn_conv2d_bwd_weight_ws_bytes. */
int frcnn_conv2d_bwd_weight_acc(const float* x, const float* dy, float* grad_w, int c_real, float* grad_b, int n, int h, int w,
                                int c, int k, int r, int s, int stride, int pad, void* ws, size_t ws_bytes, int* counters,
                                void* stream);
/* The same for `groups` (<= 24) convolutions of IDENTICAL shape in one launch pair: grad_w[g] += dW(x[g], dy[g]).  x, dy, grad_w
 * are HOST arrays of `groups` device pointers.  The repeated Bottlenecks of a ResNet stage (22 of the 23 blocks of layer3,
 * lib/nets/resnet.py:131-240) have identical filter-gradient problems whose outputs are too small to fill the chip one at a
 * time (hence pixel-split slabs and a reduction pass per layer); together they do, so every tile runs the whole pixel loop
 * and nothing is reduced afterwards.  ws: frcnn_conv2d_bwd_weight_acc_grouped_ws_bytes.  Deterministic. */
size_t frcnn_conv2d_bwd_weight_acc_grouped_ws_bytes(int groups, int c, int k, int r, int s);
int frcnn_conv2d_bwd_weight_acc_grouped(const float* const* x, const float* const* dy, float* const* grad_w, int groups,
                                        int c_real, int n, int h, int w, int c, int k, int r, int s, int stride, int pad,
                                        void* ws, size_t ws_bytes, void* stream);

/* Tuning / test hook: force the workgroup tile to (64*tm) x (64*tn) output pixels x channels for all
 * following frcnn_conv2d_fwd calls of this process; (tm,tn) in {(4,2),(2,4)} (8 waves, one workgroup per
 * CU), {(2,2),(2,1),(1,2),(1,1)} (4 waves); (0,0) restores the automatic choice. */
int frcnn_conv2d_set_tile(int tm, int tn);
/* Plan autotuner (like a "benchmark mode"): when enabled, the first frcnn_conv2d_fwd / _bwd_data call of a shape
 * OUTSIDE stream capture times every (tile, split-K) candidate on the caller's tensors with HIP events — this
 * synchronises with the host — and caches the fastest; later calls (also captured ones) reuse it.  While enabled,
 * frcnn_conv2d_fwd_ws_bytes returns room for the largest candidate of a not-yet-tuned shape.  Default: off (the
 * analytic model picks).  frcnn_conv2d_bwd_weight follows the same switch with its own cache (tile 128x128 or 64x64 x
 * pixel splits).  frcnn_conv2d_clear_plans forgets both caches.
 * enable == 2: after one pass that times every candidate alone, the six fastest are timed UNDER LOAD - four launches of the
 * candidate in flight at once on four streams (the caller's and three of the library's own; all copies compute the same values
 * into the same tensors) - and ranked by that, i.e. by throughput on a busy chip, which is what a caller that keeps several
 * frames in flight needs (lib/model/test.py:183-228 replayed four frames at a time), instead of by the latency of one launch on
 * an idle chip. */
/* Form of the cached (tuned or imported) plan of a stride-1-output forward call of this shape: -1 none, 0 implicit GEMM,
 * 1 Winograd F(2x2,3x3).  Calls with and without a residual operand are planned separately.  Callers that keep a
 * pre-transformed Winograd filter (frcnn_conv2d_fwd_pre) use this to build it only for layers whose plan reads it. */
int frcnn_conv2d_plan_algo(int n, int h, int w, int c, int k, int r, int s, int stride, int pad, int has_residual);

/* Per-dispatch timing of the kernels frcnn_conv2d_fwd launches (the main implicit-GEMM kernel and, for a split-K
 * plan, the second pass): between _begin and _end every launch gets its own start / stop HIP events on the launch stream
 * (hipExtLaunchKernelGGL), i.e. the begin -> end time of that dispatch.  _end synchronises and fills, per dispatch in
 * launch order, the duration in microseconds, the number of the frcnn_conv2d_fwd call it belongs to (0-based since
 * _begin) and its kind (0 main kernel, 1 second pass).  Returns the number of dispatches recorded.  Not for use during
 * stream capture. */
int frcnn_conv2d_profile_begin(void);
int frcnn_conv2d_profile_end(float* us, int* call, int* kind, int capacity);
int frcnn_conv2d_set_autotune(int enable);
int frcnn_conv2d_clear_plans(void);
/* The plan cache as a table of 13 ints per entry (shape key n,h,w,c,k,r,s,stride,pad,out_stride [+256: call with a residual];
 * tile index [+16: Winograd F(2x2,3x3) around the grouped GEMM, +32: with the input transform fused into the 64x64 GEMM], splits,
 * K-steps per split), so that a tuned table can be saved and replayed (e.g. under a profiler, whose instrumentation would
 * otherwise perturb the tuning; or shipped with a deployment: bench.py loads profiles/r05_plans.json).  Tile indices:
 * 0 256x128, 1 128x256 (8 waves, LDS-DMA three stages), 2 128x128, 3 128x64, 4 64x128, 5 64x64 (register-staged), 6 128x128
 * LDS-DMA two stages, 7 64x64, 8 128x64, 9 64x128, 10 128x128, 11 256x128, 12 128x256 (LDS-DMA through buffer loads, three
 * stages), 13 64x64 with PERSISTENT workgroups (one K-step stream across a workgroup's tiles).  export returns the number of cached entries (fills at most capacity_entries); import validates and inserts. */
int frcnn_conv2d_export_plans(int* out, int capacity_entries);
int frcnn_conv2d_import_plans(const int* in, int entries);

/* Tuning / test hook: 1 (default) stages the 8-wave tiles with LDS-DMA (global_load_lds) when C % 32 == 0 (and the
 * autotuner may pick the two-stage LDS-DMA 128x128 tile, plan tile index 6), 0 uses the register-staged kernels
 * everywhere, 2 additionally runs a FORCED 128x128 tile (frcnn_conv2d_set_tile(2, 2)) on the two-stage LDS-DMA kernel,
 * 3 runs FORCED tiles (and plan tile indices 0 .. 5) on the buffer-load LDS-DMA kernel (conv_igemm_buf_f32: plan tile
 * indices 7 .. 12 select it in every mode but 0; needs C % 32 == 0 and operands below 2 GB, else the register-staged kernel).
 * Results are bit-identical for split_k = 1. */
int frcnn_conv2d_set_staging(int use_lds_dma);

/* Algorithm of the 3x3 / stride 1 / pad 1 convolutions without a residual (lib/nets/resnet.py:119-121 conv2 of a
 * Bottleneck, the RPN 3x3 of the Network):  0 (default) = the autotuner times the implicit GEMM and Winograd F(2x2, 3x3)
 * (same fp32 arithmetic, 2.25x fewer multiplications, four launches) and keeps the faster; without autotuning the
 * implicit GEMM runs;  1 = implicit GEMM only;  2 = Winograd wherever it applies (tests).  A Winograd plan is exported
 * with 16 added to its tile index. */
int frcnn_conv2d_set_algo(int mode);

/* The filter side of a Winograd plan is constant while the weights are: U[16][k][c] = G g G^T of a (k,3,3,c) filter,
 * computed once per parameter version by the caller and handed to frcnn_conv2d_fwd_pre, which is frcnn_conv2d_fwd
 * with that one extra operand (NULL = transform inside the call, as frcnn_conv2d_fwd does).  The operand is only read
 * when the layer's plan is a Winograd plan. */
size_t frcnn_conv2d_winograd_filter_bytes(int k, int c);
int frcnn_conv2d_winograd_filter(const float* w_krsc, float* u, int k, int c, void* stream);
int frcnn_conv2d_fwd_pre(const float* x, const float* w_krsc, const float* w_winograd, const float* scale,
                         const float* shift, const float* residual, float* y, int n, int h, int w, int c, int k, int r,
                         int s, int stride, int pad, int relu, int split_k, void* ws, size_t ws_bytes, void* stream);

/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1)  (lib/nets/resnet.py:156), NHWC. */
int frcnn_maxpool3x3s2_fwd(const float* x, float* y, int n, int h, int w, int c, void* stream);
/* Its backward (trainable stem, cfg.RESNET.FIXED_BLOCKS == -1): dx (n,h,w,c) receives dy of every window whose first
 * maximum (row-major scan, like the index torch stores) is that pixel; gather form, deterministic. */
int frcnn_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int n, int h, int w, int c, void* stream);

/* NHWC image/BEV blob (lib/roi_data_layer/minibatch.py:670 layout, (1,H,W,C)) -> NHWC with the channel
 * count padded with zeros to c_pad (multiple of 4) so the stem conv reads 16-byte pixels. */
int frcnn_pad_channels(const float* x, float* y, int64_t pixels, int c, int c_pad, void* stream);

/* Image input producer: prep_im_for_blob + im_list_to_blob (lib/utils/blob.py:16-54) for one frame on
 * the device.  img (h,w,3) uint8 in cv2.imread order -> blob (out_h,out_w,c_out) fp32 with
 * blob[..,c] = (resize(img)[.., arrange[c]] - means[c]) / stddevs[c], c_out = 3 or 4 (4th channel zero).
 * cv2.resize(fx=fy=scale, INTER_LINEAR) semantics restated (cv2 is not vendored by the reference).
 * means / stddevs (3 doubles) and arrange (3 ints) are HOST pointers. */
int frcnn_prep_image_out_size(int h, int w, float scale, int* out_h, int* out_w);
int frcnn_prep_image(const uint8_t* img_hwc3, int h, int w, float scale, const double* means_host,
                     const double* stddevs_host, const int* arrange_host, int c_out, float* blob, void* stream);

/* ---------------------------------------------------------------------------------------------
 * RPN / proposal stage
 * ------------------------------------------------------------------------------------------- */
/* generate_anchors_pre (lib/layer_utils/snippets.py:13-40): anchors[(y*W+x)*A + a] = base[a] + shift.
 * `base` is the (A,4) table of generate_anchors (lib/layer_utils/generate_anchors.py:41-54), computed on
 * the host in float64 (DEVICE pointer to A*4 doubles); the shift is added in float64 and the sum is
 * rounded once to fp32, exactly like the reference's astype(float32). */
int frcnn_generate_anchors(const double* base, int num_base, int height, int width, int feat_stride,
                           float* anchors, void* stream);

/* GridAnchor3dGenerator._generate (lib/layer_utils/generate_3d_anchors.py:15-118) and the axis-aligned BEV
 * box of each 3-D anchor (bbaa_graphics_gems, lib/utils/bbox.py:256-293, clip=False) for the LiDAR RPN.
 * base (DEVICE, num_types x 9 floats), one row per (size, rotation) type, rotation fastest:
 *   [lo_x, lo_y, hi_x, hi_y, z, l, w, h, ry]  (host-computed like generate_3d_anchors.py:37-38,100 and
 *   bbox.py:258-279).  anchors_3d (H*W*T, 7) [x,y,z,l,w,h,ry], anchors_2d (H*W*T, 4), order (H, W, T). */
int frcnn_generate_anchors_3d(const float* base, int num_types, int height, int width, int feat_stride,
                              float* anchors_3d, float* anchors_2d, void* stream);

/* Fused front half of proposal_layer (lib/layer_utils/proposal_layer.py:32-36) on the raw RPN head
 * output `rpn` (H*W, ld) NHWC with channels [0,A) = bg logits, [A,2A) = fg logits, [2A,6A) = deltas:
 *   fg prob of the 2-way softmax, bbox_transform_inv (lib/model/bbox_transform.py:75-105) and
 *   clip_boxes (:235-257, info = [x_min,x_max,y_min,y_max,...]).
 * With probs_in != NULL the logits are ignored and (H*W*A) ready-made fg probabilities are copied
 * (the exact proposal_layer signature, which receives rpn_cls_prob). info is a HOST pointer to 4+ floats. */
int frcnn_rpn_decode_clip(const float* rpn, int ld, const float* probs_in, const float* deltas_in,
                          const float* anchors, const float* info_host, int hw, int num_anchors,
                          float* scores, float* proposals, void* stream);

/* Stand-alone box codec (the proposal and tail kernels fuse the same arithmetic):
 * bbox_transform_inv (lib/model/bbox_transform.py:75-105): boxes rows of box_ld floats whose first 4 are
 * [x1,y1,x2,y2]; deltas/out (n, 4*num_classes); scale > 0 divides the boxes first (:77-78), scale <= 0 = None.
 * clip_boxes (:235-257): num_boxes 4-float boxes; info HOST pointer [x_min,x_max,y_min,y_max]. */
int frcnn_bbox_transform_inv(const float* boxes, int box_ld, const float* deltas, int n, int num_classes,
                             float scale, float* out, void* stream);
int frcnn_clip_boxes(const float* boxes, int num_boxes, const float* info_host, float* out, void* stream);
/* lidar_3d_bbox_transform_inv (lib/model/bbox_transform.py:174-233): rois rows of roi_ld floats whose first 4
 * are [x1,y1,x2,y2]; anchors_3d (n,7); deltas/out (n, 7*num_classes); scale <= 0 = None. */
int frcnn_lidar_bbox_transform_inv(const float* rois, int roi_ld, const float* anchors_3d, const float* deltas,
                                   int n, int num_classes, float scale, float* out, void* stream);

/* uncertainty_transform_inv (lidar == 0) / lidar_3d_uncertainty_transform_inv (lidar != 0), lib/model/bbox_transform.py:
 * 107-130 / 132-169: uncertainty (n, 7K) of the deltas [x,y,z,l,w,h,ry] -> squared box-space terms, out (n, 4K) [x,y,l,w]
 * or (n, 7K).  rois (n rows, first 4 = BEV [x1,y1,x2,y2], divided by scale when scale > 0), anchors_3d (n,7) supplies the
 * height of the z term (may be NULL when lidar == 0).  input_is_variance != 0: the square root of every input element is
 * taken first (Monte-Carlo variances -> standard deviations). */
int frcnn_uncertainty_transform_inv(const float* rois, int roi_ld, const float* anchors_3d, const float* uncertainty, int n,
                                    int num_classes, float scale, int lidar, int input_is_variance, float* out, void* stream);

/* scores.sort(descending=True)[:top_n] (proposal_layer.py:39-42) with the canonical total order
 * (score desc, index asc).  order_out[top_n] int64 source indices, scores_out[top_n];
 * count_out[0] = min(n, top_n).  top_n <= 16384.  n <= 16384: one workgroup in LDS; larger n: multi-workgroup
 * radix select over all CUs (workspace from frcnn_sort_topk_desc_ws_bytes), same result. */
size_t frcnn_sort_topk_desc_ws_bytes(int n, int top_n);
int frcnn_sort_topk_desc(const float* scores, int n, int top_n, int64_t* order_out, float* scores_out,
                         int* count_out, void* ws, size_t ws_bytes, void* stream);

/* rows_out[i,:] = rows[order[i],:]  for i < count (device count), `width` floats per row. */
int frcnn_gather_rows(const float* rows, const int64_t* order, const int* count, int max_count, int width,
                      float* rows_out, void* stream);

/* torchvision.ops.nms at IoU == threshold exactly (proposal_layer.py:46, filter_predictions.py:67-69): on != 0 (default)
 * suppresses the box like torchvision 0.4.0's CPU kernel (`iou >= threshold`), 0 keeps it like its CUDA kernel
 * (`iou > threshold`).  Process-wide, read when frcnn_nms / frcnn_filter_per_class* launch. */
int frcnn_nms_set_suppress_at_equal(int on);
int frcnn_nms_get_suppress_at_equal(void);

/* torchvision.ops.nms (call sites proposal_layer.py:46, filter_predictions.py:67-69) on boxes already
 * sorted by descending score: box j is dropped when IoU(i,j) > thresh for an earlier kept i
 * (areas without +1).  n_dev (device int, may be NULL -> n_max) boxes are live.
 * keep_idx[max_keep] int64 positions of survivors in score order (entries past keep_count are written as 0),
 * keep_mask[n_max] bytes, keep_count[0] = min(#survivors, max_keep). */
size_t frcnn_nms_ws_bytes(int n_max);
int frcnn_nms(const float* boxes, const int* n_dev, int n_max, float thresh, int max_keep,
              int64_t* keep_idx, uint8_t* keep_mask, int* keep_count, void* ws, size_t ws_bytes,
              void* stream);

/* Tail of proposal_layer (proposal_layer.py:50-55): rois[i] = [0, proposals[order... keep[i]]],
 * roi_scores[i]; rows >= keep_count are zero-filled.  sorted_boxes are the score-ordered boxes. */
int frcnn_make_rois(const float* sorted_boxes, const float* sorted_scores, const int64_t* keep_idx,
                    const int* keep_count, int max_keep, float* rois, float* roi_scores, void* stream);

/* ---------------------------------------------------------------------------------------------
 * RoIAlign (torchvision.ops.roi_align 0.4.0 semantics = aligned=False; call sites
 * lib/utils/torchpoolers.py:165-170,194-197 and the _crop_pool_layer of the missing network.py).
 * feat NHWC (n,H,W,C); rois (R,5) [batch,x1,y1,x2,y2]; out (R,P,P,C) NHWC.
 * sampling_ratio <= 0 -> adaptive ceil(roi/P).  roi_count (device int, may be NULL) masks rows
 * >= count to zero.  level_of_roi/level (may be NULL/-1): only rois with level_of_roi[r]==level
 * are written (MultiScaleRoIAlign, torchpoolers.py:187-199).
 * ------------------------------------------------------------------------------------------- */
/* feat holds n images (n,H,W,C).  rois_per_image == 0: the image of a RoI is its batch column and roi_count (may be NULL)
 * is ONE live count; rois_per_image > 0: rows [i*rois_per_image, (i+1)*rois_per_image) belong to image i whatever their
 * batch column says, and roi_count points to n live counts (frames batched into one call).
 * Tuning / test hook: 0 automatic (map-resident kernel when one 16-channel slice of the map fits the LDS: pooled 7,
 * c % 16 == 0, h <= 64, h*w*64 B + 10 KB <= 160 KB; else the planned pair when a workspace is given; else generic),
 * 1 generic kernel, 2 generic with per-XCD channel slices, 3 / 4 planned kernel with 8 / 4 loads in flight per lane,
 * 5 map-resident kernel or an error. */
int frcnn_roi_align_set_variant(int variant);
/* Scratch for the planned kernel pair: the compact work-item list + the per-bin axis weight tables.  Returns 0
 * when the shape has no planned path (pooled != 7).  Passing ws == NULL to frcnn_roi_align_fwd is always valid (the
 * map-resident and the generic kernels need none). */
size_t frcnn_roi_align_fwd_ws_bytes(int h, int w, int c, int num_rois, int pooled);
int frcnn_roi_align_fwd(const float* feat, int n, int h, int w, int c, const float* rois, const int* roi_count,
                        int num_rois, int rois_per_image, int pooled, float spatial_scale, int sampling_ratio,
                        const int* level_of_roi, int level, float* out, void* ws, size_t ws_bytes, void* stream);
/* frcnn_roi_align_fwd with a per-channel epilogue on the pooled value: out = act(pooled * scale[c] + shift[c]) (scale / shift
 * device float[c] or NULL, relu 0/1).  Pooling and a bias-free 1x1 convolution commute (both linear, different axes), so
 * layer4[0].conv1 and layer4[0].downsample[0] - which the reference applies to pool5, lib/nets/resnet.py:98-127 via
 * _head_to_tail - can run on the H x W feature map BEFORE the pooling (6x fewer pixels than 300 x 7 x 7) with their folded
 * BatchNorm + ReLU applied here, after the pooling, where the reference applies them.  Not available with the map-resident
 * kernel (variant 5). */
int frcnn_roi_align_fwd_affine(const float* feat, int n, int h, int w, int c, const float* rois, const int* roi_count,
                               int num_rois, int rois_per_image, int pooled, float spatial_scale, int sampling_ratio,
                               const int* level_of_roi, int level, float* out, const float* scale, const float* shift,
                               int relu, void* ws, size_t ws_bytes, void* stream);
/* One map, two outputs over the channel ranges [0, split_c) and [split_c, c), ONE plan launch: out1 (num_rois,7,7,split_c) =
 * act1(pooled * scale + shift), out2 (num_rois,7,7,c - split_c) likewise (scale / shift: device float[c] or NULL, indexed by
 * the MAP's channel; relu1 / relu2 0/1).  The inference path runs layer4[0].conv1 and layer4[0].downsample[0]
 * (lib/nets/resnet.py:98-127) as one 1x1 convolution with concatenated filters and pools its 512 + 2048 channels here.
 * split_c % 256 == 0; pooled == 7; ws: frcnn_roi_align_fwd_ws_bytes(h, w, 4, num_rois, 7). */
int frcnn_roi_align_fwd_split(const float* feat, int h, int w, int c, const float* rois, const int* roi_count, int num_rois,
                              int pooled, float spatial_scale, int sampling_ratio, int split_c, float* out1, float* out2,
                              const float* scale, const float* shift, int relu1, int relu2, void* ws, size_t ws_bytes,
                              void* stream);

/* LevelMapper (lib/utils/torchpoolers.py:20-51) of MultiScaleRoIAlign: levels[i] = clamp(floor(canonical_level +
 * log2(sqrt(area_i) / canonical_scale) + eps), k_min, k_max) - k_min for rois (n,5); area without +1. */
int frcnn_fpn_level_map(const float* rois, int num_rois, int k_min, int k_max, float canonical_scale,
                        float canonical_level, float eps, int* levels, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Detection tail (_head_to_tail mean + _region_classification + test-time decode of network.py;
 * evidence lib/model/test.py:75-79, lib/model/config.py:219-223):
 *   fc7 = x.mean(3).mean(2); cls_score = fc7 W_c^T + b_c; cls_prob = softmax; deltas = fc7 W_b^T + b_b;
 *   pred_boxes = bbox_transform_inv(rois[:,1:5], deltas*stds+means, scale).
 * x (R,P,P,C) NHWC; w_cls (K,C); w_box (K*4,C); stds/means HOST pointers to 4 floats.
 * Outputs: fc7 (R,C), cls_score (R,K), cls_prob (R,K), bbox_pred (R,4K) raw, pred_boxes (R,4K).
 * ------------------------------------------------------------------------------------------- */
int frcnn_head_fc_softmax_decode(const float* x, int num_rois, int pooled, int c, const float* w_cls,
                                 const float* b_cls, const float* w_box, const float* b_box,
                                 int num_classes, const float* rois, const float* stds_host,
                                 const float* means_host, float scale, float* fc7, float* cls_score,
                                 float* cls_prob, float* bbox_pred, float* pred_boxes, void* stream);

/* LiDAR tail: as above with 7-DoF boxes — bbox_pred (R,7K), pred_boxes = lidar_3d_bbox_transform_inv(
 * rois[:,1:5], roi_anchors_3d, deltas*stds+means, scale) (lib/model/bbox_transform.py:174-233);
 * roi_anchors_3d (R,7) are the 3-D anchors of the RoIs (proposal_layer.py:44,52); stds/means 7 floats. */
int frcnn_head_fc_softmax_decode_lidar(const float* x, int num_rois, int pooled, int c, const float* w_cls,
                                       const float* b_cls, const float* w_box, const float* b_box,
                                       int num_classes, const float* rois, const float* roi_anchors_3d,
                                       const float* stds_host, const float* means_host, float scale, float* fc7,
                                       float* cls_score, float* cls_prob, float* bbox_pred, float* pred_boxes,
                                       void* stream);

/* filter_and_draw_prep + nms_hstack_torch + the max_dets cut (lib/utils/filter_predictions.py:75-130,
 * 45-72; lib/model/test.py:210-221) for the image detector, all on the device:
 * clamp to [0, frame/scale-1] in place, per class j>=1 keep score > thresh, NMS(nms_thresh) in
 * descending score order, keep dets with score >= the max_dets-th best.
 * dets (K, max_out, 5) [x1,y1,x2,y2,score], det_count (K) ints: EVERY row is written by the call (rows past a class's
 * count and the whole background class 0 are zero), the caller need not clear them.  roi_count device int or NULL.
 * det_roi (K, max_out) ints or NULL: RoI row of every detection (-1 past det_count) - what nms_hstack_var_torch
 * (filter_predictions.py:23-43) needs to gather the per-RoI uncertainties of the kept detections. */
/* Test hook: 0 = automatic (num_rois <= 1024: LDS-resident kernel), 1 = always the general workspace kernel. */
int frcnn_filter_set_variant(int variant);
size_t frcnn_filter_per_class_ws_bytes(int num_rois, int num_classes);
int frcnn_filter_per_class(float* pred_boxes, const float* cls_prob, const int* roi_count, int num_rois,
                           int num_classes, float frame_w, float frame_h, float scale, float thresh,
                           float nms_thresh, int max_dets, int max_out, float* dets, int* det_count,
                           int* det_roi, void* ws, size_t ws_bytes, void* stream);

/* LiDAR input producer (lib/roi_data_layer/minibatch.py:232-235,434-512): points (num_points, point_stride >= 4)
 * rows [x,y,z,intensity,(elongation)...] in file order -> bev (gy, gx, num_slices + num_meta) fp32, the blob of one
 * frame (already transposed to y-major like the reference's final np.transpose).
 * pc_range_host[6] = [xmin,ymin,zmin,xmax,ymax,zmax] AFTER the reference's shift (zmin = 0, zmax = Z1 - Z0),
 * z_shift = cfg.LIDAR.Z_RANGE[0] (subtracted from every z), voxel_size_host[3]; grid = round((max-min)/size).
 * spconv VoxelGeneratorV2 semantics restated (third party, absent: parity unpinned): voxels numbered by first
 * appearance, at most max_voxels, each keeps its first max_points points.  Height slice = max z - slice*voxel height;
 * meta channels (density, tanh(mean intensity), tanh(mean elongation) or 0 when elongation_col < 0): the voxel
 * created last in a column wins.  num_voxels (device int, may be NULL) = occupied cells before the max_voxels cap. */
int frcnn_bev_voxelize_grid(const float* pc_range_host, const float* voxel_size_host, int* grid_host);
size_t frcnn_bev_voxelize_ws_bytes(int num_points, const float* pc_range_host, const float* voxel_size_host,
                                   int max_voxels);
int frcnn_bev_voxelize(const float* points, int num_points, int point_stride, const float* pc_range_host,
                       const float* voxel_size_host, float z_shift, int max_points, int max_voxels, int num_slices,
                       int num_meta, int elongation_col, float* bev, int* num_voxels, void* ws, size_t ws_bytes,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training path (BASELINE config 4: FPN forward + backward of one train_step, lib/model/train_val.py:458)
 * ------------------------------------------------------------------------------------------- */
/* Backward of act(conv*scale + shift + residual): dz = relu ? (y > 0 ? dy : 0) : dy; d_res = dz (may be NULL);
 * d_conv = dz * scale[k] (scale may be NULL).  rows x k fp32, k % 4 == 0.  (F.relu / frozen BatchNorm / the
 * residual add of lib/nets/resnet.py:98-127.) */
int frcnn_act_bwd(const float* dy, const float* y, const float* scale, int relu, int64_t rows, int k,
                  float* d_conv, float* d_res, void* stream);

/* BatchNorm2d with BATCH statistics (module in train() mode; the LiDAR backbone trains layer2/layer3 this way,
 * lib/nets/lidarnet.py:110,152-175; torch.nn.functional.batch_norm(training=True)).  y/out/residual/dout/dy/dres are
 * rows x c fp32 (NHWC activations), c % 4 == 0.
 *   fwd: mean/biased var over the rows -> save_mean, save_invstd = 1/sqrt(var+eps);
 *        out = act((y - mean)*alpha + beta [+ residual]), alpha = gamma*invstd (gamma/beta may be NULL = 1/0);
 *        running_mean/var (may be NULL) <- (1-momentum)*running + momentum*(mean | UNBIASED var).
 *   bwd: g = relu ? (out > 0 ? dout : 0) : dout; dbeta = sum g; dgamma = sum g*xhat;
 *        dy = gamma*invstd*(g - dbeta/rows - xhat*dgamma/rows); dres = g (may be NULL).
 *        accumulate != 0: dgamma / dbeta are the parameters' own gradient buffers and are added into.
 * Column sums are carried in fp64 and added in a fixed order.
 * counters: frcnn_bn_train_counters(c) ints, zero on entry and left zero, not shared by launches that can be in flight
 * together (the rule of frcnn_conv2d_bwd_weight's counters): the last workgroup of a 64-channel column block turns the column
 * sums into the statistics / coefficients itself - two launches per call instead of three.  NULL: a separate pass does. */
size_t frcnn_bn_train_ws_bytes(int c);
int frcnn_bn_train_counters(int c);
int frcnn_bn_train_fwd(const float* y, int64_t rows, int c, const float* gamma, const float* beta, float eps,
                       float momentum, float* running_mean, float* running_var, const float* residual, int relu,
                       float* out, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, int* counters,
                       void* stream);
int frcnn_bn_train_bwd(const float* dout, const float* out, const float* y, int64_t rows, int c, const float* gamma,
                       const float* save_mean, const float* save_invstd, int relu, float* dy, float* dres,
                       float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, int* counters,
                       void* stream);

/* fc7 = x.mean(3).mean(2) (_head_to_tail of the non-FPN detector; x (rows,P,P,c) NHWC -> out (rows,c)) and its
 * backward dx = dout / P^2 broadcast over the P x P positions. */
int frcnn_spatial_mean_fwd(const float* x, float* out, int rows, int pooled, int c, void* stream);
int frcnn_spatial_mean_bwd(const float* dout, float* dx, int rows, int pooled, int c, void* stream);

/* fpn._upsample_add (lib/nets/fpn.py:42-45): out (n,out_h,out_w,c) = F.interpolate(x (n,h,w,c), size=(out_h,out_w),
 * mode='bilinear', align_corners=False) + lateral.  Backward: dx = interpolate^T(dout) (deterministic gather);
 * the gradient of `lateral` is dout itself. */
int frcnn_upsample_bilinear_add_fwd(const float* x, const float* lateral, float* out, int n, int h, int w,
                                    int out_h, int out_w, int c, void* stream);
int frcnn_upsample_bilinear_bwd(const float* dout, float* dx, int n, int h, int w, int out_h, int out_w, int c,
                                void* stream);

/* Backward of frcnn_roi_align_fwd: dfeat (1,H,W,C) += scatter(dout (R,P,P,C)); dfeat must be zero-filled by the
 * caller (float atomics: the summation order, hence the last bits, vary from run to run; the same holds for
 * frcnn_roi_align_bwd_planned and frcnn_scatter_add_patches - every other kernel of the library is deterministic). */
int frcnn_roi_align_bwd(const float* dout, int h, int w, int c, const float* rois, const int* roi_count,
                        int num_rois, int pooled, float spatial_scale, int sampling_ratio, const int* level_of_roi,
                        int level, float* dfeat, void* stream);
/* The same gradient through the forward's plan (frcnn_roi_align_fwd_ws_bytes of the shape as workspace; pooled 7, c % 4 == 0):
 * the row bins of a window row are folded in registers first, so ONE atomic is issued per (RoI, window pixel of a column bin,
 * channel) instead of four per sample - 2.5x (2x2 samples per bin) to 10x+ (larger adaptive grids) fewer atomics.  Same sums
 * in another order (float atomics either way). */
int frcnn_roi_align_bwd_planned(const float* dout, int h, int w, int c, const float* rois, const int* roi_count,
                                int num_rois, int pooled, float spatial_scale, int sampling_ratio, const int* level_of_roi,
                                int level, float* dfeat, void* ws, size_t ws_bytes, void* stream);

/* The RPN losses read only the anchors the anchor target layer labelled (<= cfg.TRAIN.RPN_BATCHSIZE,
 * lib/layer_utils/anchor_target_layer.py:91-107), so the gradient of the RPN head is non-zero on at most that many pixels.
 * frcnn_labelled_pixels: labels (hw * A) in (H,W,A) order -> idx[cap] = the pixels with a label != -1, ascending, -1 beyond
 * count[0]; count[0] = min(total, cap), count[1] = total (a caller checks count[1] <= cap where the labels are not its own).
 * frcnn_gather_patches: out (cap, r, s, c) = the r x s window of x (h, w, c) around each listed pixel (zero outside the map and
 * for rows >= count[0]) - the input of a VALID r x s convolution that reproduces the padded convolution at those pixels
 * (lib/nets/network.py rpn_net on net_conv).  frcnn_scatter_add_patches: its adjoint, dx += scatter(d) with float atomics. */
size_t frcnn_labelled_pixels_ws_bytes(int hw);
int frcnn_labelled_pixels(const float* labels, int hw, int num_anchors, int cap, int64_t* idx, int* count, void* ws,
                          size_t ws_bytes, void* stream);
int frcnn_gather_patches(const float* x, int h, int w, int c, const int64_t* idx, const int* count, int cap, int r, int s,
                         int pad, float* out, void* stream);
int frcnn_scatter_add_patches(const float* d, int h, int w, int c, const int64_t* idx, const int* count, int cap, int r, int s,
                              int pad, float* dx, void* stream);

/* RPN losses on the fused head output rpn (hw, ld) = [A bg | A fg | 4A deltas | pad]:
 *   losses[0] = F.cross_entropy over anchors with labels != -1 (mean), losses[1] = smooth_l1_loss('RPN', ...,
 *   dim=[1,2,3]) (lib/utils/loss_utils.py:39-101), losses[2] = number of labelled anchors.
 * labels (hw*A) in (H,W,A) order with values -1/0/1; targets/inside/outside (hw*A, 4).
 * drpn (hw, ld), may be NULL: gradient of grad_ce*losses[0] + grad_box*losses[1]. */
size_t frcnn_rpn_loss_ws_bytes(void);
int frcnn_rpn_loss(const float* rpn, int ld, int num_anchors, int hw, const float* labels, const float* targets,
                   const float* inside, const float* outside, float grad_ce, float grad_box, float* losses,
                   float* drpn, void* ws, size_t ws_bytes, void* stream);

/* Detection losses on the sampled RoIs: losses[0] = F.cross_entropy(cls_score (R,K), labels (R) as floats),
 * losses[1] = smooth_l1_loss('DET', bbox_pred (R,E*K), targets, inside, outside) (image boxes);
 * dcls / dbox (may be NULL): gradients of grad_ce*losses[0] + grad_box*losses[1]. */
int frcnn_det_loss(const float* cls_score, const float* labels, int num_rois, int num_classes,
                   const float* bbox_pred, const float* targets, const float* inside, const float* outside,
                   int bbox_elem, float grad_ce, float grad_box, float* losses, float* dcls, float* dbox,
                   void* stream);

/* bbox_overlaps (lib/utils/bbox.py:5-33): IoU with the +1 area convention; boxes (n rows of box_ld floats, first
 * 4 = [x1,y1,x2,y2]) x query (k rows of query_ld floats) -> overlaps (n,k). */
int frcnn_bbox_overlaps(const float* boxes, int box_ld, int n, const float* query, int query_ld, int k,
                        float* overlaps, void* stream);

/* bbox_transform (lib/model/bbox_transform.py:52-70): regression targets of gt_rois against ex_rois, row by row
 * (n rows of ex_ld / gt_ld floats, first 4 = [x1,y1,x2,y2]) -> targets (n,4) [dx,dy,dw,dh]; dx,dy over the box
 * DIAGONAL, widths with the +1 convention. */
int frcnn_bbox_transform(const float* ex_rois, int ex_ld, const float* gt_rois, int gt_ld, int n, float* targets,
                         void* stream);

/* lidar_3d_bbox_transform (lib/model/bbox_transform.py:16-49): ex_rois (n rows, first 4 = BEV [x1,y1,x2,y2]),
 * ex_anchors_3d (n,7), gt_rois (n rows of gt_ld >= 7 floats [xc,yc,zc,l,w,h,ry]) -> targets (n,7). */
int frcnn_lidar_bbox_transform(const float* ex_rois, int roi_ld, const float* ex_anchors_3d, const float* gt_rois,
                               int gt_ld, int n, float* targets, void* stream);

/* anchor_target_layer_torch (lib/layer_utils/anchor_target_layer.py:22-165; IGNORE_DC off, CLOBBER_POSITIVES off,
 * uniform example weights): anchors (n,4) in (H,W,A) order, gt_boxes (num_gt,5) [x1,y1,x2,y2,cls], info HOST
 * [x_min,x_max,y_min,y_max].  Outputs in anchor order: labels (n) in {-1,0,1}, targets/inside/outside (n,4);
 * counts (2 ints, may be NULL) = fg / bg candidates before sub-sampling.  Sub-sampling to
 * fg_fraction*rpn_batchsize foreground and rpn_batchsize total draws hash keys from `seed` + *seed_dev (seed_dev: device
 * uint32, may be NULL - a launch replayed from a hipGraph keeps its `seed` argument, so per-step seeds come through
 * device memory the caller rewrites before each replay; same for the proposal target layers below). */
size_t frcnn_anchor_target_layer_ws_bytes(int num_anchors_total, int num_gt, int rpn_batchsize);
int frcnn_anchor_target_layer(const float* anchors, int n, const float* gt_boxes, int num_gt,
                              const int* num_gt_dev /* device int or NULL: live rows of a gt buffer padded to num_gt rows
                              (one captured training step serves every number of gt boxes).  REQUIRED: 1 <= *num_gt_dev <=
                              num_gt - the kernels clamp the value into that range, so a 0 would make row 0 of the buffer a
                              ground-truth box; a frame without gt boxes must not be launched (the reference stops on it too,
                              lib/layer_utils/proposal_target_layer.py:232-235) */,
                              const float* info_host, int rpn_batchsize, float fg_fraction, float negative_overlap,
                              float positive_overlap, uint32_t seed, const uint32_t* seed_dev, float* labels,
                              float* targets, float* inside, float* outside, int* counts, void* ws, size_t ws_bytes,
                              void* stream);

/* proposal_target_layer (lib/layer_utils/proposal_target_layer.py:22-262, image detector, USE_GT / IGNORE_DC off):
 * rois (num_rois,5), roi_scores (num_rois) or NULL, roi_count device int or NULL, gt_boxes (num_gt,5).
 * Outputs with rois_per_frame rows (foreground rows first): labels, out_rois (.,5), out_scores, targets / inside /
 * outside (., 4*num_classes; targets normalised with means/stds, HOST 4 floats each), gt_assignment (.) ints,
 * counts[4] = {fg rows, bg rows, fg candidates, bg candidates}.  num_rois <= 4096.
 * skip_mask (num_rois bytes, may be NULL): rows with a non-zero byte are no candidates (TRAIN.IGNORE_DC: proposals whose
 * overlap with a don't-care box reaches DC_THRESH, proposal_target_layer.py:180-191). */
int frcnn_proposal_target_layer(const float* rois, const float* roi_scores, const int* roi_count, int num_rois,
                                const float* gt_boxes, int num_gt, const int* num_gt_dev /* as above */, int num_classes,
                                int rois_per_frame,
                                float fg_fraction, float fg_thresh, float bg_thresh_hi, float bg_thresh_lo,
                                const float* means_host, const float* stds_host, uint32_t seed, const uint32_t* seed_dev,
                                float* labels, float* out_rois, float* out_scores, float* targets, float* inside,
                                float* outside, int* gt_assignment, int* counts, const unsigned char* skip_mask,
                                void* stream);

/* LiDAR form (proposal_target_layer.py:142-154, NET_TYPE 'lidar'): overlaps and labels on the BEV rectangles gt_boxes
 * (num_gt,5); targets = lidar_3d_bbox_transform(roi, the RoI's 3-D anchor, true_gt_boxes (num_gt,8)
 * [xc,yc,zc,l,w,h,ry,cls]) (lib/model/bbox_transform.py:16-49) normalised with 7 means/stds; anchors3d (num_rois,7)
 * follows the RoIs, out_anchors3d (rois_per_frame,7) the sampled rows; targets/inside/outside are (.,7*num_classes). */
int frcnn_proposal_target_layer_lidar(const float* rois, const float* roi_scores, const int* roi_count, int num_rois,
                                      const float* anchors3d, const float* gt_boxes, const float* true_gt_boxes,
                                      int num_gt, const int* num_gt_dev, int num_classes, int rois_per_frame,
                                      float fg_fraction, float fg_thresh, float bg_thresh_hi, float bg_thresh_lo,
                                      const float* means_host,
                                      const float* stds_host, uint32_t seed, const uint32_t* seed_dev, float* labels,
                                      float* out_rois, float* out_scores, float* out_anchors3d, float* targets, float* inside,
                                      float* outside, int* gt_assignment, int* counts, const unsigned char* skip_mask,
                                      void* stream);

/* frcnn_det_loss for the 7-element LiDAR boxes (lib/utils/loss_utils.py:61-77): the yaw difference goes through
 * sin() before the Huber term when ry_sin (cfg.LIDAR.EN_RY_SIN), every element is scaled by reg_loss_weight_host[7]
 * (cfg.LIDAR.REG_LOSS_WEIGHT, NULL = ones). */
int frcnn_det_loss_lidar(const float* cls_score, const float* labels, int num_rois, int num_classes,
                         const float* bbox_pred, const float* targets, const float* inside, const float* outside,
                         const float* reg_loss_weight_host, int ry_sin, float grad_ce, float grad_box, float* losses,
                         float* dcls, float* dbox, void* stream);

/* Uncertainty pieces whose arithmetic IS in the reference snapshot (lib/utils/loss_utils.py):
 *  - frcnn_det_loss with the aleatoric attenuation of loss_utils.py:82-85: bbox_var = predicted log-variance s,
 *    per-element loss (0.5*huber*exp(-s) + 0.5*s)*inside; dvar receives d(loss)/ds.  bbox_elem 4 or 7
 *    (7: sin(ry) + reg_loss_weight_host as in frcnn_det_loss_lidar).
 *  - frcnn_mc_bbox_var: samples (T, elems) -> unbiased variance over the T stochastic passes, clamped at 0
 *    (compute_bbox_var, loss_utils.py:114-120).
 *  - frcnn_mc_cls_stats: cls_score samples (T, num_rois, K) -> mean softmax (num_rois,K), its entropy in bits and the
 *    mutual information H(mean p) - mean H(p) (loss_utils.py:122-141). */
int frcnn_det_loss_aleatoric(const float* cls_score, const float* labels, int num_rois, int num_classes,
                             const float* bbox_pred, const float* bbox_var, const float* targets, const float* inside,
                             const float* outside, int bbox_elem, const float* reg_loss_weight_host, int ry_sin,
                             float grad_ce, float grad_box, float* losses, float* dcls, float* dbox, float* dvar,
                             void* stream);
int frcnn_mc_bbox_var(const float* samples, int num_samples, int64_t elems, float* var, void* stream);
int frcnn_mc_cls_stats(const float* cls_score_samples, int num_samples, int num_rois, int num_classes,
                       float* mean_prob, float* entropy, float* mutual_info, float* prob_var /* (num_rois,K) or NULL:
                       compute_bbox_var of the softmax samples */, void* stream);
/* mean over the T leading samples of a (T, elems) stack (mean logits / mean deltas of the Monte-Carlo passes). */
int frcnn_mc_mean(const float* samples, int num_samples, int64_t elems, float* mean, void* stream);

/* Uncertainty heads (rebuilt from the module names and rates of lib/nets/imagenet.py:52-91 / lidarnet.py:56-102; wiring
 * choices are named constants in nets/network.py).  Random draws are counter-based, value = f(seed, stream_id, index)
 * (csrc/rng.h), so the CPU oracle replays them:
 *  - frcnn_dropout_fwd: nn.Dropout(p) in train() mode on `repeat` stochastic copies of x (elems): y (repeat, elems);
 *    repeat = cfg.UC.E_NUM_SAMPLE starts the Monte-Carlo passes (lib/model/test.py:74-77) from one activation.
 *  - frcnn_dropout_bwd: dx (elems) = sum over the copies of mask * dy / (1 - p).
 *  - frcnn_logit_distort: logit_distort of loss_utils.py:143-147, samples (S, elems) = score + sqrt(var) * N(0,1).
 *  - frcnn_bayesian_cross_entropy: loss_utils.py:149-169 on the same draws: loss[0] = mean over RoIs of
 *    -log(mean_s softmax(score + sqrt(var) eps_s)[label]); per_roi (num_rois) scratch/diagnostic; dscore / dvar
 *    (num_rois, K, may be NULL) receive grad * d loss / d{score, var}.
 *  `seed_dev` (device uint32, may be NULL): the draws use seed + *seed_dev - a launch replayed from a hipGraph keeps its
 *  scalar `seed`, so the per-frame seed of a captured frame / training step comes through device memory (version 107). */
int frcnn_dropout_fwd(const float* x, int64_t elems, int repeat, float p, uint32_t seed, const uint32_t* seed_dev,
                      uint32_t stream_id, float* y, void* stream);
int frcnn_dropout_bwd(const float* dy, int64_t elems, int repeat, float p, uint32_t seed, const uint32_t* seed_dev,
                      uint32_t stream_id, float* dx, void* stream);
int frcnn_logit_distort(const float* score, const float* var, int64_t elems, int num_samples, uint32_t seed,
                        const uint32_t* seed_dev, uint32_t stream_id, int var_is_log /* var holds log-variances */,
                        float* samples, float* var_out /* (elems) exp(var) or var; may be NULL */, void* stream);
int frcnn_bayesian_cross_entropy(const float* cls_score, const float* cls_var, const float* labels, int num_rois,
                                 int num_classes, int num_samples, uint32_t seed, const uint32_t* seed_dev,
                                 uint32_t stream_id, int var_is_log, float grad, float* loss, float* per_roi,
                                 float* dscore, float* dvar, void* stream);
/* y = exp(x) elementwise (log-variance heads -> variances at test time). */
int frcnn_exp(const float* x, int64_t elems, float* y, void* stream);

/* LiDAR form (filter_predictions.py:55-62,67, db_type 'lidar'): no clamp, NMS on the yaw-less BEV rectangle
 * xc -+ l/2, yc -+ w/2 of the 7-DoF boxes, dets (K, max_out, 8) [xc,yc,zc,l,w,h,ry,score].
 * Workspace: frcnn_filter_per_class_ws_bytes. */
int frcnn_filter_per_class_lidar(const float* pred_boxes, const float* cls_prob, const int* roi_count,
                                 int num_rois, int num_classes, float thresh, float nms_thresh, int max_dets,
                                 int max_out, float* dets, int* det_count, int* det_roi, void* ws, size_t ws_bytes,
                                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FRCNN_HIP_H_ */
