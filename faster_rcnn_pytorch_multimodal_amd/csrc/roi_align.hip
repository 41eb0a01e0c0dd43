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
// (RoI, CC-channel slice): it builds Wx, Wy in LDS, then walks the RoI window in chunks of RCH feature rows: the seven
// per-column-bin row sums T[y][pw] of the chunk go to LDS, and every thread adds the chunk's rows into the bins it
// owns (accumulators stay in registers across chunks).  Each window pixel is read about once instead of up to
// 4*gh*gw times per bin (the generic kernel is bound by the 64 B/clk/CU vector-L1 path), and the LDS footprint is
// RCH*P*CC*4 bytes whatever the window height: the op is a chain of short dependent phases, so what matters is how
// many workgroups a CU can keep in flight (8 at ~17 KB each), not the work of one.
// The summation order differs from the library kernel the oracle follows (agreement ~1e-6 relative, well inside the
// 1e-4 bar).  Workgroup b handles slice b % nslices, so an XCD (b % 8) only touches 1/8 of the channels (L2 locality).
// ------------------------------------------------------------------------------------------------
template <int CC, int RCH>
__global__ __launch_bounds__(256) void roi_align_fwd_sep(const float* __restrict__ feat, int H, int W, int C,
                                                        const float* __restrict__ rois,
                                                        const int* __restrict__ roi_count, int num_rois, int P,
                                                        float spatial_scale, int sampling_ratio,
                                                        const int* __restrict__ level_of_roi, int level,
                                                        float* __restrict__ out) {
  constexpr int CL = CC / 4;          // float4 per pixel of the slice
  constexpr int MAXJ = 4;             // bins*CL / 256 rounded up, for P = 7: 392/256 -> 2 (CC 32), 784/256 -> 4 (CC 64)
  extern __shared__ __attribute__((aligned(16))) float sep_smem[];
  const int nslices = C / CC;
  const int r = blockIdx.x / nslices, slice = blockIdx.x - r * nslices;
  if (level_of_roi && level_of_roi[r] != level) return;
  const int t = threadIdx.x;
  const int live = roi_count ? min(*roi_count, num_rois) : num_rois;
  const int C4 = C >> 2;
  const int items = P * P * CL;
  float4* ob = reinterpret_cast<float4*>(out) + (size_t)r * P * P * C4 + slice * CL;
  if (r >= live) {
    for (int i = t; i < items; i += 256) ob[(size_t)(i / CL) * C4 + (i % CL)] = make_float4(0.f, 0.f, 0.f, 0.f);
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
  const float4* fb = reinterpret_cast<const float4*>(feat) + (size_t)b * H * W * C4 + slice * CL;

  // LDS: Wx[P][W], Wy[P][H], per-bin ranges, then T[RCH][P][CL] float4
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
  // the bins this thread owns: item = t + 256*j -> (bin, c4)
  float4 acc[MAXJ];
  int a_pw[MAXJ], a_c4[MAXJ], a_yl[MAXJ], a_yh[MAXJ];
  const float* a_wr[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int i = min(t + 256 * j, items - 1);
    const int bin = i / CL, ph = bin / P;
    a_c4[j] = i % CL;
    a_pw[j] = bin - ph * P;
    a_yl[j] = rng[2 * P + ph];
    a_yh[j] = rng[3 * P + ph];
    a_wr[j] = Wy + ph * H;
  }
  for (int ys = y0; ys <= y1; ys += RCH) {      // y1 < y0: no valid sample at all -> zeros
    const int rows = min(RCH, y1 - ys + 1);
    // phase 1: T[y - ys][pw][c4] = sum_x Wx[pw][x] * F[y][x]; item = (row, pw, c4)
    for (int i = t; i < rows * P * CL; i += 256) {
      const int c4 = i % CL, pw = (i / CL) % P, yr = i / (CL * P);
      const float4* frow = fb + (size_t)(ys + yr) * W * C4 + c4;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      const int xl = rng[pw], xh = rng[P + pw];
      const float* wr = Wx + pw * W;
      for (int x = xl; x <= xh; ++x) {
        const float wgt = wr[x];
        const float4 v = frow[(size_t)x * C4];
        a.x += wgt * v.x; a.y += wgt * v.y; a.z += wgt * v.z; a.w += wgt * v.w;
      }
      T[(yr * P + pw) * CL + c4] = a;
    }
    __syncthreads();
    // phase 2: acc[ph][pw] += sum_{y in chunk} Wy[ph][y] * T[y][pw]
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      if (t + 256 * j >= items) continue;
      const int lo = max(a_yl[j], ys), hi = min(a_yh[j], ys + rows - 1);
      for (int y = lo; y <= hi; ++y) {
        const float wgt = a_wr[j][y];
        const float4 v = T[((y - ys) * P + a_pw[j]) * CL + a_c4[j]];
        acc[j].x += wgt * v.x; acc[j].y += wgt * v.y; acc[j].z += wgt * v.z; acc[j].w += wgt * v.w;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int i = t + 256 * j;
    if (i >= items) continue;
    ob[(size_t)(i / CL) * C4 + a_c4[j]] =
        make_float4(acc[j].x / count, acc[j].y / count, acc[j].z / count, acc[j].w / count);
  }
}

template <int CC, int RCH>
int launch_sep(const float* feat, int h, int w, int c, const float* rois, const int* roi_count, int num_rois, int pooled,
               float spatial_scale, int sampling_ratio, const int* level_of_roi, int level, float* out,
               hipStream_t stream) {
  const size_t tables = ((size_t)pooled * (w + h) + 4 * pooled + 3) & ~(size_t)3;
  const size_t lds = tables * 4 + (size_t)RCH * pooled * CC * 4;
  static size_t configured = 0;
  if (lds > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&roi_align_fwd_sep<CC, RCH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return frcnn::fail(FRCNN_ERR_LAUNCH, "roi_align_fwd: set LDS size: %s", hipGetErrorString(e));
    configured = lds;
  }
  hipLaunchKernelGGL((roi_align_fwd_sep<CC, RCH>), dim3((unsigned)(num_rois * (c / CC))), dim3(256), lds, stream, feat, h,
                     w, c, rois, roi_count, num_rois, pooled, spatial_scale, sampling_ratio, level_of_roi, level, out);
  return frcnn::check_launch("roi_align_fwd_sep");
}

}  // namespace

// tuning hook: 0 = automatic, 1 = generic, 2 = generic with XCD channel slices, 3..6 = separable <32,12> <64,8> <64,16> <32,24>
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
  // separable kernel: needs the 7x7-style item count to fit the per-thread bin ownership (pooled*pooled*CC/4 <= 1024)
  // and the weight tables to fit in LDS next to the row chunk
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const bool sep_ok = pooled <= 8 && (size_t)pooled * (w + h) * 4 < 48 * 1024;
  if (sep_ok && c % 64 == 0 && g_roi_variant == 4)
    return launch_sep<64, 8>(feat, h, w, c, rois, roi_count, num_rois, pooled, spatial_scale, sampling_ratio, level_of_roi,
                             level, out, stream);
  if (sep_ok && c % 64 == 0 && (g_roi_variant == 0 || g_roi_variant == 5))   // measured best (round 1)
    return launch_sep<64, 16>(feat, h, w, c, rois, roi_count, num_rois, pooled, spatial_scale, sampling_ratio,
                              level_of_roi, level, out, stream);
  if (sep_ok && c % 32 == 0 && (g_roi_variant == 0 || g_roi_variant == 3))
    return launch_sep<32, 12>(feat, h, w, c, rois, roi_count, num_rois, pooled, spatial_scale, sampling_ratio,
                              level_of_roi, level, out, stream);
  if (sep_ok && c % 32 == 0 && g_roi_variant == 6)
    return launch_sep<32, 24>(feat, h, w, c, rois, roi_count, num_rois, pooled, spatial_scale, sampling_ratio,
                              level_of_roi, level, out, stream);
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
