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

// XCD_SPLIT: workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one), so workgroup b works on
// channel slice b % 8 only: each XCD's private 4 MB L2 then holds 1/8 of the feature map (1.2 MB of the 9.8 MB
// res101 map) instead of thrashing on all of it — the PMC passes showed 127 MB fetched from beyond L2 per launch
// against 9.8 MB algorithmic.  Placement only changes speed: every (bin, channel) item is computed exactly once
// either way.  The thread->item map keeps one thread per (bin, 4 channels), i.e. the same parallelism.
template <bool XCD_SPLIT>
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc(const float* __restrict__ feat, int H, int W, int C4,
                                                         const float* __restrict__ rois,
                                                         const int* __restrict__ roi_count, int num_rois, int P,
                                                         float spatial_scale, int sampling_ratio,
                                                         const int* __restrict__ level_of_roi, int level,
                                                         float* __restrict__ out) {
  const int live = roi_count ? min(*roi_count, num_rois) : num_rois;
  const size_t total = (size_t)num_rois * P * P * C4;
  const int cs = XCD_SPLIT ? C4 >> 3 : C4;                 // float4 groups per channel slice
  const size_t per_slice = total / (XCD_SPLIT ? 8 : 1);
  const int slice = XCD_SPLIT ? (int)(blockIdx.x & 7) : 0;
  const size_t first = XCD_SPLIT ? (size_t)(blockIdx.x >> 3) * blockDim.x + threadIdx.x
                                 : (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = XCD_SPLIT ? (size_t)(gridDim.x >> 3) * blockDim.x : (size_t)gridDim.x * blockDim.x;
  for (size_t it = first; it < per_slice; it += step) {
    const int c4 = slice * cs + (int)(it % cs);
    size_t bin = it / cs;
    const size_t item = bin * C4 + c4;
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


// ------------------------------------------------------------------------------------------------
// Separable form (C % 32 == 0).  Bilinear sampling factorises per axis — sample weight = wy(y) * wx(x), a sample is
// dropped when EITHER coordinate is outside [-1, size] — so a bin's average is
//     out[ph][pw] = 1/count * sum_y Wy[ph][y] * ( sum_x Wx[pw][x] * F[y][x] )
// with Wy / Wx the per-axis weights of a bin's samples accumulated per feature row / column.  One workgroup =
// (RoI, 32-channel slice): it builds Wx, Wy in LDS, walks every feature row of the RoI window ONCE accumulating the
// seven per-column-bin row sums T[y][pw] (LDS), then combines rows into the 49 bins.  Each window pixel is read
// about once instead of up to 4*gh*gw times per bin: the generic kernel is bound by the 64 B/clk/CU vector-L1
// path (3.6 GB of L1 traffic for 300 RoIs), this one moves ~10x less.  The summation order differs from the
// library kernel the oracle follows (agreement ~1e-6 relative, well inside the 1e-4 bar).
// Workgroup b handles slice b % nslices, so an XCD (b % 8) only touches 1/8 of the channels (L2 locality).
// RoIs whose window is taller than the LDS budget fall back to the direct per-bin loop inside the same kernel.
// ------------------------------------------------------------------------------------------------
constexpr int SEP_CC = 32;        // channels per workgroup
constexpr int SEP_CL = SEP_CC / 4;

__global__ __launch_bounds__(256) void roi_align_fwd_sep(const float* __restrict__ feat, int H, int W, int C,
                                                        const float* __restrict__ rois,
                                                        const int* __restrict__ roi_count, int num_rois, int P,
                                                        float spatial_scale, int sampling_ratio,
                                                        const int* __restrict__ level_of_roi, int level, int hcap,
                                                        float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sep_smem[];
  const int nslices = C / SEP_CC;
  const int r = blockIdx.x / nslices, slice = blockIdx.x - r * nslices;
  if (level_of_roi && level_of_roi[r] != level) return;
  const int t = threadIdx.x;
  const int live = roi_count ? min(*roi_count, num_rois) : num_rois;
  const int C4 = C >> 2;
  float4* ob = reinterpret_cast<float4*>(out) + (size_t)r * P * P * C4 + slice * SEP_CL;
  if (r >= live) {
    for (int i = t; i < P * P * SEP_CL; i += 256) ob[(size_t)(i / SEP_CL) * C4 + (i % SEP_CL)] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float* roi = rois + (size_t)r * 5;
  const int b = (int)roi[0];
  const float roi_start_w = roi[1] * spatial_scale, roi_start_h = roi[2] * spatial_scale;
  const float roi_end_w = roi[3] * spatial_scale, roi_end_h = roi[4] * spatial_scale;
  const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.0f), roi_height = fmaxf(roi_end_h - roi_start_h, 1.0f);
  const float bin_size_h = roi_height / (float)P, bin_size_w = roi_width / (float)P;
  const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)P);
  const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)P);
  const float count = (float)(grid_h * grid_w);
  const float4* fb = reinterpret_cast<const float4*>(feat) + (size_t)b * H * W * C4 + slice * SEP_CL;

  // LDS: Wx[P][W], Wy[P][H], per-bin ranges, then T[rows][P][SEP_CL] float4
  float* Wx = sep_smem;
  float* Wy = Wx + P * W;
  int* rng = reinterpret_cast<int*>(Wy + P * H);   // xlo[P], xhi[P], ylo[P], yhi[P]
  float4* T = reinterpret_cast<float4*>(sep_smem + ((P * W + P * H + 4 * P + 3) & ~3));

  // one axis sample -> (low index, high index, weight of low, weight of high, valid)
  auto axis_sample = [](float start, float bin_size, int pbin, int i, int grid, int size, int& lo, int& hi, float& wlo,
                        float& whi) -> bool {
    float v = start + pbin * bin_size + ((float)i + .5f) * bin_size / (float)grid;
    if (v < -1.0f || v > (float)size) return false;
    if (v <= 0.f) v = 0.f;
    lo = (int)v;
    if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; } else hi = lo + 1;
    whi = v - (float)lo;
    wlo = 1.f - whi;
    return true;
  };
  for (int i = t; i < P * W + P * H; i += 256) sep_smem[i] = 0.f;
  __syncthreads();
  // per-axis weights: thread (axis, bin) walks its samples in order -> deterministic sums
  if (t < 2 * P) {
    const bool is_x = t < P;
    const int pb = is_x ? t : t - P;
    const int grid = is_x ? grid_w : grid_h, size = is_x ? W : H;
    float* wrow = is_x ? Wx + pb * W : Wy + pb * H;
    int lo_all = size, hi_all = -1;
    for (int i = 0; i < grid; ++i) {
      int lo, hi;
      float wlo, whi;
      if (!axis_sample(is_x ? roi_start_w : roi_start_h, is_x ? bin_size_w : bin_size_h, pb, i, grid, size, lo, hi, wlo, whi))
        continue;
      wrow[lo] += wlo;
      wrow[hi] += whi;
      lo_all = min(lo_all, lo);
      hi_all = max(hi_all, hi);
    }
    rng[(is_x ? 0 : 2 * P) + pb] = lo_all;
    rng[(is_x ? P : 3 * P) + pb] = hi_all;
  }
  __syncthreads();
  int y0 = H, y1 = -1;
  for (int ph = 0; ph < P; ++ph) { y0 = min(y0, rng[2 * P + ph]); y1 = max(y1, rng[3 * P + ph]); }
  const int rows = y1 - y0 + 1;   // <= 0: no valid sample at all
  if (rows > hcap) {
    // window taller than the LDS budget: direct per-bin evaluation (same maths as the generic kernel)
    for (int i = t; i < P * P * SEP_CL; i += 256) {
      const int c4 = i % SEP_CL, bin = i / SEP_CL;
      const int ph = bin / P, pw = bin - ph * P;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int iy = 0; iy < grid_h; ++iy) {
        int yl, yh;
        float wyl, wyh;
        if (!axis_sample(roi_start_h, bin_size_h, ph, iy, grid_h, H, yl, yh, wyl, wyh)) continue;
        for (int ix = 0; ix < grid_w; ++ix) {
          int xl, xh;
          float wxl, wxh;
          if (!axis_sample(roi_start_w, bin_size_w, pw, ix, grid_w, W, xl, xh, wxl, wxh)) continue;
          const float4 v1 = fb[((size_t)yl * W + xl) * C4 + c4], v2 = fb[((size_t)yl * W + xh) * C4 + c4];
          const float4 v3 = fb[((size_t)yh * W + xl) * C4 + c4], v4 = fb[((size_t)yh * W + xh) * C4 + c4];
          const float w1 = wyl * wxl, w2 = wyl * wxh, w3 = wyh * wxl, w4 = wyh * wxh;
          acc.x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
          acc.y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
          acc.z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
          acc.w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
        }
      }
      ob[(size_t)bin * C4 + c4] = make_float4(acc.x / count, acc.y / count, acc.z / count, acc.w / count);
    }
    return;
  }
  // phase 1: T[y - y0][pw][c4] = sum_x Wx[pw][x] * F[y][x]; item = (row, pw, c4)
  for (int i = t; i < rows * P * SEP_CL; i += 256) {
    const int c4 = i % SEP_CL, pw = (i / SEP_CL) % P, yr = i / (SEP_CL * P);
    const float4* frow = fb + (size_t)(y0 + yr) * W * C4 + c4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int xl = rng[pw], xh = rng[P + pw];
    const float* wr = Wx + pw * W;
    for (int x = xl; x <= xh; ++x) {
      const float wgt = wr[x];
      const float4 v = frow[(size_t)x * C4];
      acc.x += wgt * v.x; acc.y += wgt * v.y; acc.z += wgt * v.z; acc.w += wgt * v.w;
    }
    T[(yr * P + pw) * SEP_CL + c4] = acc;
  }
  __syncthreads();
  // phase 2: out[ph][pw] = 1/count * sum_y Wy[ph][y] * T[y][pw]
  for (int i = t; i < P * P * SEP_CL; i += 256) {
    const int c4 = i % SEP_CL, bin = i / SEP_CL;
    const int ph = bin / P, pw = bin - ph * P;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int yl = rng[2 * P + ph], yh = rng[3 * P + ph];
    const float* wr = Wy + ph * H;
    for (int y = yl; y <= yh; ++y) {
      const float wgt = wr[y];
      const float4 v = T[((y - y0) * P + pw) * SEP_CL + c4];
      acc.x += wgt * v.x; acc.y += wgt * v.y; acc.z += wgt * v.z; acc.w += wgt * v.w;
    }
    ob[(size_t)bin * C4 + c4] = make_float4(acc.x / count, acc.y / count, acc.z / count, acc.w / count);
  }
}

}  // namespace

// tuning hook: 0 = automatic, 1 = generic, 2 = generic with XCD channel slices, 3 = separable
static int g_roi_variant = 0;
extern "C" int frcnn_roi_align_set_variant(int v) {
  g_roi_variant = v;
  return FRCNN_OK;
}

extern "C" int frcnn_roi_align_fwd(const float* feat, int h, int w, int c, const float* rois, const int* roi_count,
                                   int num_rois, int pooled, float spatial_scale, int sampling_ratio,
                                   const int* level_of_roi, int level, float* out, void* stream_) {
  FRCNN_REQUIRE(feat && rois && out && h > 0 && w > 0 && c > 0 && c % 4 == 0 && num_rois > 0 && pooled > 0,
                "roi_align_fwd: bad arguments (c%%4==0)");
  if ((g_roi_variant == 0 || g_roi_variant == 3) && c % SEP_CC == 0 && pooled <= 16 &&
      (size_t)pooled * (w + h) * 4 < 48 * 1024) {
    // separable kernel: LDS = weight tables + T rows; windows taller than hcap take the in-kernel direct path
    const size_t tables = ((size_t)pooled * (w + h) + 4 * pooled + 3) & ~(size_t)3;
    const int hcap = std::min(std::min(h, 48), (int)((96 * 1024 - tables * 4) / ((size_t)pooled * SEP_CC * 4)));
    const size_t lds = tables * 4 + (size_t)hcap * pooled * SEP_CC * 4;
    static size_t configured = 0;
    if (lds > configured) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&roi_align_fwd_sep),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return frcnn::fail(FRCNN_ERR_LAUNCH, "roi_align_fwd: set LDS size: %s", hipGetErrorString(e));
      configured = lds;
    }
    hipLaunchKernelGGL(roi_align_fwd_sep, dim3((unsigned)(num_rois * (c / SEP_CC))), dim3(256), lds,
                       static_cast<hipStream_t>(stream_), feat, h, w, c, rois, roi_count, num_rois, pooled, spatial_scale,
                       sampling_ratio, level_of_roi, level, hcap, out);
    return frcnn::check_launch("roi_align_fwd_sep");
  }
  const size_t total = (size_t)num_rois * pooled * pooled * (c / 4);
  if (g_roi_variant != 1 && c % 32 == 0) {   // 8 channel slices of c/8 channels, one per XCD; grid = multiple of 8
    const size_t per_slice_blocks = std::min<size_t>((total / 8 + 255) / 256, (size_t)1 << 17);
    hipLaunchKernelGGL(roi_align_fwd_nhwc<true>, dim3((unsigned)(per_slice_blocks * 8)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), feat, h, w, c / 4, rois, roi_count, num_rois, pooled,
                       spatial_scale, sampling_ratio, level_of_roi, level, out);
    return frcnn::check_launch("roi_align_fwd_nhwc<xcd>");
  }
  const size_t blocks = std::min<size_t>((total + 255) / 256, (size_t)1 << 20);
  hipLaunchKernelGGL(roi_align_fwd_nhwc<false>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     feat, h, w, c / 4, rois, roi_count, num_rois, pooled, spatial_scale, sampling_ratio, level_of_roi,
                     level, out);
  return frcnn::check_launch("roi_align_fwd_nhwc");
}
