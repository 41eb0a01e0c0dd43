// RoIAlign forward, torchvision 0.4.0 semantics (aligned=False), NHWC in / NHWC out.
// Reference call sites: lib/utils/torchpoolers.py:165-170,194-197 (roi_align) and the 'align' pooling of
// the missing network.py (_crop_pool_layer; POOLING_MODE lib/model/config.py:364).
//
// HBM-bound: the output (R*P*P*C floats) dominates the traffic.  One work item = 4 consecutive
// channels of one output bin; consecutive threads walk the channel dimension, so every neighbour-pixel
// read and every output write is a contiguous 16 B/lane stream and all sampling weights are
// wave-uniform (scalar registers).  Built with -ffp-contract=off so the sample coordinates and the
// accumulation order follow the CPU kernel of the library the reference calls.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void roi_align_fwd_nhwc(const float* __restrict__ feat, int H, int W, int C4,
                                                         const float* __restrict__ rois,
                                                         const int* __restrict__ roi_count, int num_rois, int P,
                                                         float spatial_scale, int sampling_ratio,
                                                         const int* __restrict__ level_of_roi, int level,
                                                         float* __restrict__ out) {
  const int live = roi_count ? min(*roi_count, num_rois) : num_rois;
  const size_t total = (size_t)num_rois * P * P * C4;
  for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total;
       item += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(item % C4);
    size_t bin = item / C4;
    const int pw = (int)(bin % P);
    bin /= P;
    const int ph = (int)(bin % P);
    const int r = (int)(bin / P);
    if (level_of_roi && level_of_roi[r] != level) continue;  // another level writes this roi
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < live) {
      const float* roi = rois + (size_t)r * 5;
      const int b = (int)roi[0];
      const float roi_start_w = roi[1] * spatial_scale;
      const float roi_start_h = roi[2] * spatial_scale;
      const float roi_end_w = roi[3] * spatial_scale;
      const float roi_end_h = roi[4] * spatial_scale;
      const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.0f);
      const float roi_height = fmaxf(roi_end_h - roi_start_h, 1.0f);
      const float bin_size_h = roi_height / (float)P;
      const float bin_size_w = roi_width / (float)P;
      const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)P);
      const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)P);
      const float count = (float)(grid_h * grid_w);
      const float4* fb = reinterpret_cast<const float4*>(feat) + (size_t)b * H * W * C4 + c4;
      for (int iy = 0; iy < grid_h; ++iy) {
        float y = roi_start_h + ph * bin_size_h + ((float)iy + .5f) * bin_size_h / (float)grid_h;
        for (int ix = 0; ix < grid_w; ++ix) {
          float x = roi_start_w + pw * bin_size_w + ((float)ix + .5f) * bin_size_w / (float)grid_w;
          float yy = y;
          if (yy < -1.0f || yy > (float)H || x < -1.0f || x > (float)W) continue;  // empty sample
          if (yy <= 0.f) yy = 0.f;
          if (x <= 0.f) x = 0.f;
          int y_low = (int)yy, x_low = (int)x, y_high, x_high;
          if (y_low >= H - 1) { y_high = y_low = H - 1; yy = (float)y_low; } else y_high = y_low + 1;
          if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
          const float ly = yy - (float)y_low, lx = x - (float)x_low;
          const float hy = 1.f - ly, hx = 1.f - lx;
          const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
          const float4 v1 = fb[((size_t)y_low * W + x_low) * C4];
          const float4 v2 = fb[((size_t)y_low * W + x_high) * C4];
          const float4 v3 = fb[((size_t)y_high * W + x_low) * C4];
          const float4 v4 = fb[((size_t)y_high * W + x_high) * C4];
          acc.x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
          acc.y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
          acc.z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
          acc.w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
        }
      }
      acc.x /= count; acc.y /= count; acc.z /= count; acc.w /= count;
    }
    reinterpret_cast<float4*>(out)[item] = acc;
  }
}

}  // namespace

extern "C" int frcnn_roi_align_fwd(const float* feat, int h, int w, int c, const float* rois, const int* roi_count,
                                   int num_rois, int pooled, float spatial_scale, int sampling_ratio,
                                   const int* level_of_roi, int level, float* out, void* stream_) {
  FRCNN_REQUIRE(feat && rois && out && h > 0 && w > 0 && c > 0 && c % 4 == 0 && num_rois > 0 && pooled > 0,
                "roi_align_fwd: bad arguments (c%%4==0)");
  const size_t total = (size_t)num_rois * pooled * pooled * (c / 4);
  const size_t blocks = std::min<size_t>((total + 255) / 256, (size_t)1 << 20);
  hipLaunchKernelGGL(roi_align_fwd_nhwc, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), feat, h,
                     w, c / 4, rois, roi_count, num_rois, pooled, spatial_scale, sampling_ratio, level_of_roi, level, out);
  return frcnn::check_launch("roi_align_fwd_nhwc");
}
