// Detection tail: spatial mean (fc7), the two Linear heads, class softmax and test-time box decode,
// fused in one launch (one workgroup per RoI).
//
// Reference: _head_to_tail tail `.mean(3).mean(2)` and _region_classification of the missing
// lib/nets/network.py (module names lib/nets/imagenet.py:83-86), test-time de-normalisation with
// cfg.TRAIN.IMAGE.BBOX_NORMALIZE_{STDS,MEANS} (lib/model/config.py:222-223) and
// bbox_transform_inv(rois[:,1:5], deltas, scales=info[6]) (lib/model/bbox_transform.py:75-105;
// evidence lib/model/test.py:75-79, lib/utils/filter_predictions.py:85-91).
// Built with -ffp-contract=off (box_math.h).
#include "common.h"
#include "box_math.h"

using namespace frcnn;

namespace {

constexpr int HEAD_THREADS = 256;
constexpr int MAX_OUT = 64;  // K + 4K <= 64  ->  up to 12 classes

struct HeadNorm {
  float stds[7], means[7];
};

// E = 4: image boxes [x1,y1,x2,y2]; E = 7: LiDAR boxes [xc,yc,zc,l,w,h,ry] decoded from the RoI and the
// RoI's 3-D anchor (lib/model/bbox_transform.py:174-233).
template <int E>
__global__ __launch_bounds__(HEAD_THREADS) void head_fc_softmax_decode_kernel(
    const float* __restrict__ x, int P, int C, const float* __restrict__ w_cls, const float* __restrict__ b_cls,
    const float* __restrict__ w_box, const float* __restrict__ b_box, int K, const float* __restrict__ rois,
    const float* __restrict__ roi_anchors, HeadNorm norm, float scale, float* __restrict__ fc7,
    float* __restrict__ cls_score, float* __restrict__ cls_prob, float* __restrict__ bbox_pred,
    float* __restrict__ pred_boxes) {
  extern __shared__ __attribute__((aligned(16))) float head_smem[];  // [C] fc7 + [MAX_OUT] head outputs
  float* s_fc7 = head_smem;
  float* s_out = head_smem + C;
  const int r = blockIdx.x, t = threadIdx.x;
  const int C4 = C / 4;
  const float4* xr = reinterpret_cast<const float4*>(x) + (size_t)r * P * P * C4;
  const float inv = (float)P;
  // fc7 = x.mean(3).mean(2): mean over W inside each row, then mean over the P row means
  for (int c4 = t; c4 < C4; c4 += HEAD_THREADS) {
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h = 0; h < P; ++h) {
      float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int w = 0; w < P; ++w) {
        const float4 v = xr[(size_t)(h * P + w) * C4 + c4];
        row.x += v.x; row.y += v.y; row.z += v.z; row.w += v.w;
      }
      tot.x += row.x / inv; tot.y += row.y / inv; tot.z += row.z / inv; tot.w += row.w / inv;
    }
    tot.x /= inv; tot.y /= inv; tot.z /= inv; tot.w /= inv;
    reinterpret_cast<float4*>(s_fc7)[c4] = tot;
    if (fc7) reinterpret_cast<float4*>(fc7)[(size_t)r * C4 + c4] = tot;
  }
  __syncthreads();
  // K class logits + 4K box deltas: one wave per output, 64 lanes stride the channel dimension
  const int lane = t & 63, wave = t >> 6;
  const int n_out = K * (1 + E);
  for (int o = wave; o < n_out; o += HEAD_THREADS / 64) {
    const float* wrow = o < K ? w_cls + (size_t)o * C : w_box + (size_t)(o - K) * C;
    float acc = 0.f;
    for (int c4 = lane; c4 < C4; c4 += 64) {
      const float4 a = reinterpret_cast<const float4*>(s_fc7)[c4];
      const float4 b = reinterpret_cast<const float4*>(wrow)[c4];
      acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) s_out[o] = acc + (o < K ? b_cls[o] : b_box[o - K]);
  }
  __syncthreads();
  if (t == 0) {
    // softmax over classes (torch: subtract max, exp, normalise)
    float m = s_out[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, s_out[k]);
    float e[MAX_OUT / 5 + 1], sum = 0.f;  // K <= MAX_OUT / (1 + E) <= MAX_OUT / 5
    for (int k = 0; k < K; ++k) { e[k] = exp_f32(s_out[k] - m); sum += e[k]; }
    for (int k = 0; k < K; ++k) {
      cls_score[(size_t)r * K + k] = s_out[k];
      cls_prob[(size_t)r * K + k] = e[k] / sum;
    }
  }
  if (t < K) {
    // boxes = rois[:,1:5] / scale ; deltas*stds + means ; decode
    const float* roi = rois + (size_t)r * 5;
    const float x1 = roi[1] / scale, y1 = roi[2] / scale, x2 = roi[3] / scale, y2 = roi[4] / scale;
    float d[E];
    for (int q = 0; q < E; ++q) {
      const float raw = s_out[K + t * E + q];
      bbox_pred[(size_t)r * K * E + t * E + q] = raw;
      d[q] = raw * norm.stds[q] + norm.means[q];
    }
    float o[E];
    if (E == 4) {
      decode_box(x1, y1, x2, y2, d[0], d[1], d[2], d[3], o);
    } else {
      float d7[7], o7[7];
      for (int q = 0; q < 7; ++q) d7[q] = d[q < E ? q : 0];
      decode_box_lidar(x1, y1, x2, y2, roi_anchors + (size_t)r * 7, d7, o7);
      for (int q = 0; q < E; ++q) o[q] = o7[q];
    }
    for (int q = 0; q < E; ++q) pred_boxes[(size_t)r * K * E + t * E + q] = o[q];
  }
}

}  // namespace

template <int E>
static int launch_head(const float* x, int num_rois, int pooled, int c, const float* w_cls, const float* b_cls,
                       const float* w_box, const float* b_box, int num_classes, const float* rois,
                       const float* roi_anchors, const float* stds_host, const float* means_host, float scale,
                       float* fc7, float* cls_score, float* cls_prob, float* bbox_pred, float* pred_boxes,
                       void* stream_) {
  FRCNN_REQUIRE(x && w_cls && b_cls && w_box && b_box && rois && stds_host && means_host && cls_score && cls_prob &&
                    bbox_pred && pred_boxes && (E == 4 || roi_anchors),
                "head_fc_softmax_decode: null argument");
  FRCNN_REQUIRE(num_rois > 0 && pooled > 0 && c > 0 && c % 4 == 0 && num_classes >= 2 &&
                    num_classes * (1 + E) <= MAX_OUT && scale > 0.f,
                "head_fc_softmax_decode: bad shape (c%%4==0, 2 <= classes <= %d)", MAX_OUT / (1 + E));
  HeadNorm norm;
  for (int q = 0; q < 7; ++q) { norm.stds[q] = q < E ? stds_host[q] : 1.f; norm.means[q] = q < E ? means_host[q] : 0.f; }
  const size_t lds = ((size_t)c + MAX_OUT) * sizeof(float);
  FRCNN_REQUIRE(lds <= 64 * 1024, "head_fc_softmax_decode: c=%d too large", c);
  hipLaunchKernelGGL(head_fc_softmax_decode_kernel<E>, dim3(num_rois), dim3(HEAD_THREADS), lds,
                     static_cast<hipStream_t>(stream_), x, pooled, c, w_cls, b_cls, w_box, b_box, num_classes, rois,
                     roi_anchors, norm, scale, fc7, cls_score, cls_prob, bbox_pred, pred_boxes);
  return check_launch("head_fc_softmax_decode_kernel");
}

extern "C" int frcnn_head_fc_softmax_decode(const float* x, int num_rois, int pooled, int c, const float* w_cls,
                                            const float* b_cls, const float* w_box, const float* b_box,
                                            int num_classes, const float* rois, const float* stds_host,
                                            const float* means_host, float scale, float* fc7, float* cls_score,
                                            float* cls_prob, float* bbox_pred, float* pred_boxes, void* stream_) {
  return launch_head<4>(x, num_rois, pooled, c, w_cls, b_cls, w_box, b_box, num_classes, rois, nullptr, stds_host,
                        means_host, scale, fc7, cls_score, cls_prob, bbox_pred, pred_boxes, stream_);
}

extern "C" int frcnn_head_fc_softmax_decode_lidar(const float* x, int num_rois, int pooled, int c,
                                                  const float* w_cls, const float* b_cls, const float* w_box,
                                                  const float* b_box, int num_classes, const float* rois,
                                                  const float* roi_anchors_3d, const float* stds_host,
                                                  const float* means_host, float scale, float* fc7,
                                                  float* cls_score, float* cls_prob, float* bbox_pred,
                                                  float* pred_boxes, void* stream_) {
  return launch_head<7>(x, num_rois, pooled, c, w_cls, b_cls, w_box, b_box, num_classes, rois, roi_anchors_3d,
                        stds_host, means_host, scale, fc7, cls_score, cls_prob, bbox_pred, pred_boxes, stream_);
}
