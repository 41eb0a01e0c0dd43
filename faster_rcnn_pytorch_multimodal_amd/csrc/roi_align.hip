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

#include <algorithm>
#include <atomic>
#include <type_traits>

namespace {

// Optional per-channel affine + ReLU applied to a pooled value before it is stored: out = act(pooled * scale[c] + shift[c]).
// Lets a bias-free 1x1 convolution that FOLLOWS the pooling in the reference (layer4[0].conv1 / downsample[0] on pool5,
// lib/nets/resnet.py:98-127) run BEFORE it on the feature map - pooling and a 1x1 convolution are both linear and act on
// different axes, so they commute - with its folded BatchNorm and activation still applied after the pooling.
struct RoiEpilogue {
  const float* scale;   // [C] or nullptr (1)
  const float* shift;   // [C] or nullptr (0)
  int relu;
  __device__ __forceinline__ bool any() const { return scale || shift || relu; }
  __device__ __forceinline__ float4 apply(float4 v, int c4) const {
    if (scale) {
      const float4 s = reinterpret_cast<const float4*>(scale)[c4];
      v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
    }
    if (shift) {
      const float4 b = reinterpret_cast<const float4*>(shift)[c4];
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    return v;
  }
};

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
                                                         float* __restrict__ out, RoiEpilogue epi) {
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
    reinterpret_cast<float4*>(out)[item] = epi.apply(acc, c4);
  }
}


// ------------------------------------------------------------------------------------------------
// Planned separable form — the fast path (P == 7).
//
// Bilinear sampling factorises per axis — sample weight = wy(y) * wx(x), a sample is dropped when EITHER coordinate is
// outside [-1, size] — so a bin's average is
//     out[ph][pw] = 1/count * sum_y Wy[ph][y] * ( sum_x Wx[pw][x] * F[y][x] )
// with Wy / Wx the per-axis weights of a bin's samples accumulated per feature row / column.
//
// Kernel 1, roi_plan_kernel: one workgroup per RoI, one wave per (axis, bin).  Lane i evaluates sample i of the bin
// ONCE (coordinate arithmetic of the library kernel, built with -ffp-contract=off), then every pixel lane walks the
// samples in order through v_readlane: deterministic, no read-modify-write.  It writes the axis weight tables
// (Wx[P][.] relative to the bin's first column, Wy[P][.] by absolute row, zero outside a bin's range) and appends the
// RoI's work items to a COMPACT list (its position = the item count of the RoIs before it, which every workgroup
// recomputes from the boxes: no atomics, deterministic order).  Items = (RoI, row-bin piece, 256-channel slice):
// a LIGHT RoI is one piece (all P row bins in one pass over the window, each feature row read once); a HEAVY RoI (an
// untrained RPN proposes frame-wide boxes: 38 rows x 9 columns per bin = 340 loads per lane, a chain of 40 dependent
// load groups) gets one piece per row bin (rows at a bin border are read by both neighbours: <= (rows + P) / rows
// extra traffic, small exactly when the RoI is tall).
//
// Kernel 2, roi_align_fwd_planned: no LDS, no barrier.  One WAVE-ITEM = (item, column bin pw); lane l owns channels
// 4*(64*slice + l) .. +3, so every feature-map access of the wave is one contiguous 1 KB line and every per-pixel
// weight is wave-uniform: wx (lane j = column xlo + j) and wy[ph] (lane j = row yc + j) come from the plan and are
// broadcast with v_readlane.  The window is walked pixel by pixel (row-major) in groups of G independent 16-byte
// loads per lane (4 VALU per pixel: address, readlane, two v_pk_fma_f32); a finished row's column-weighted sum is
// folded into the row-bin accumulators.  The grid is PERSISTENT — as many waves as the chip holds at once, each
// looping over wave-items with a fixed stride — and software-pipelined across items: while item i is pooled, the
// tables of item i+1 and the descriptor of item i+2 are already in flight, so a wave's dependent chain per item is
// its pixel-load groups only.
// Workgroup b only takes items of slice b % nslices, so with nslices | 8 an XCD (b % 8) touches one channel slice of
// the map (1/nslices of it stays in its 4 MB L2).
// What bounded the earlier forms (profiles/r02_roi_align.md): LDS row-chunk kernel, 4800 workgroups each a chain of
// short dependent phases: 46 us; register form with per-lane redundant sample arithmetic and IEEE divisions: 2000
// VALU per wave, VALUBusy 50 %, 54 us; lane-parallel samples, one workgroup per (RoI, piece, slice): 35 us of which
// 13.6 us for the launch / scalar-load / exit skeleton of the 7200 unused pieces; persistent over a static item space
// without pipelining: 31 us (a wave lived 12 us on average, 31 us at worst: per-item latency chain x imbalance).
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bool axis_sample_pt(float start, float bin_size, int pbin, int i, int grid, int size, int& lo,
                                               int& hi, float& wlo, float& whi) {
  float v = start + pbin * bin_size + ((float)i + .5f) * bin_size / (float)grid;
  if (v < -1.0f || v > (float)size) return false;
  if (v <= 0.f) v = 0.f;
  lo = (int)v;
  if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; } else hi = lo + 1;
  whi = v - (float)lo;
  wlo = 1.f - whi;
  return true;
}

__device__ __forceinline__ float lane_bcast(float v, int lane) {   // lane is wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int lane_bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

constexpr int PLAN_P = 7;
constexpr int NO_PIXEL = -(1 << 30);
enum { ROI_SKIP = 0, ROI_LIGHT = 1, ROI_HEAVY = 2, ROI_DEAD = 3 };

struct RoiItem {                  // 32 words, everything wave-uniform a wave-item needs
  int r, sp, slice, flag;         // flag: ROI_LIGHT (all row bins), ROI_HEAVY (row bin sp only), ROI_DEAD (zero fill)
  int y0, y1;                     // feature rows touched by the item's row bins (y0 > y1: none)
  float inv_count;
  int bimg;
  int xlo[8], xhi[8];             // columns touched by column bin pw (lo > hi: none)
  int pad[8];
};
static_assert(sizeof(RoiItem) == 128, "item descriptor is 32 words");

struct RoiPlanHead {              // first 64 bytes of the workspace
  int total_items;
  int pad[15];
};

__host__ __device__ inline int plan_pad(int n) { return (n + 63) & ~63; }

// flag of RoI r from the box alone (every workgroup of the plan kernel evaluates it for the RoIs before its own).
// Branch-free: the box is loaded unconditionally so that the loads of the whole prefix pass are in flight together.
__device__ __forceinline__ int roi_flag(const float* __restrict__ rois, int r, int live, const int* __restrict__ level_of_roi,
                                        int level, float spatial_scale, int H, int W, int heavy_loads) {
  const float* roi = rois + (size_t)r * 5;
  const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
  const int lvl = level_of_roi ? level_of_roi[r] : level;
  const float roi_width = fmaxf(x2 * spatial_scale - x1 * spatial_scale, 1.0f);
  const float roi_height = fmaxf(y2 * spatial_scale - y1 * spatial_scale, 1.0f);
  const int rows_est = min((int)roi_height + 2, H), cols_est = min((int)(roi_width / (float)PLAN_P) + 2, W);
  int flag = rows_est * cols_est > heavy_loads ? ROI_HEAVY : ROI_LIGHT;
  flag = r >= live ? ROI_DEAD : flag;
  return lvl != level ? ROI_SKIP : flag;
}

// one workgroup per RoI, 2P waves: wave w < P builds column bin w, wave P + w row bin w
__global__ __launch_bounds__(64 * 2 * PLAN_P) void roi_plan_kernel(int H, int W, const float* __restrict__ rois,
                                                                   const int* __restrict__ roi_count, int num_rois,
                                                                   float spatial_scale, int sampling_ratio,
                                                                   const int* __restrict__ level_of_roi, int level,
                                                                   int nslices, int heavy_loads,
                                                                   RoiPlanHead* __restrict__ head,
                                                                   RoiItem* __restrict__ items, float* __restrict__ wxt,
                                                                   float* __restrict__ wyt) {
  constexpr int P = PLAN_P;
  __shared__ int s_count[2 * P];
  __shared__ int s_lo[2 * P], s_hi[2 * P];
  const int r = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int live = roi_count ? min(*roi_count, num_rois) : num_rois;
  // position of this RoI's items in the compact list: pieces of the RoIs before it
  int before = 0;
  for (int q = threadIdx.x; q < r; q += blockDim.x) {
    const int f = roi_flag(rois, q, live, level_of_roi, level, spatial_scale, H, W, heavy_loads);
    before += f == ROI_SKIP ? 0 : (f == ROI_HEAVY ? P : 1);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
  if (lane == 0) s_count[wave] = before;
  const int flag = roi_flag(rois, r, live, level_of_roi, level, spatial_scale, H, W, heavy_loads);
  int plo = 0, phi = -1;
  float inv_count = 0.f;
  int bimg = 0;
  if (flag == ROI_LIGHT || flag == ROI_HEAVY) {
    const float* roi = rois + (size_t)r * 5;
    bimg = (int)roi[0];
    const float roi_start_w = roi[1] * spatial_scale, roi_start_h = roi[2] * spatial_scale;
    const float roi_end_w = roi[3] * spatial_scale, roi_end_h = roi[4] * spatial_scale;
    const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.0f), roi_height = fmaxf(roi_end_h - roi_start_h, 1.0f);
    const float bin_size_h = roi_height / (float)P, bin_size_w = roi_width / (float)P;
    const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)P);
    const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)P);
    inv_count = 1.0f / (float)(grid_h * grid_w);
    const bool is_x = wave < P;
    const int bin = is_x ? wave : wave - P;
    const float start = is_x ? roi_start_w : roi_start_h, bin_size = is_x ? bin_size_w : bin_size_h;
    const int grid = __builtin_amdgcn_readfirstlane(is_x ? grid_w : grid_h), size = is_x ? W : H;
    float* table = is_x ? wxt + ((size_t)r * P + bin) * plan_pad(W) : wyt + ((size_t)r * P + bin) * plan_pad(H);
    // lane-parallel sample evaluation; invalid -> NO_PIXEL
    auto sample = [&](int i, int& lo, int& hi, float& wl, float& wh) {
      lo = hi = NO_PIXEL;
      wl = wh = 0.f;
      if (i < grid) {
        int l, hh;
        if (axis_sample_pt(start, bin_size, bin, i, grid, size, l, hh, wl, wh)) { lo = l; hi = hh; }
      }
    };
    // sample coordinates are non-decreasing, so the first valid sample has the smallest first pixel and the last valid
    // sample the largest second pixel
    plo = size;
    for (int i0 = 0; i0 < grid; i0 += 64) {
      int lo, hi;
      float wl, wh;
      sample(i0 + lane, lo, hi, wl, wh);
      const unsigned long long m = __ballot(lo != NO_PIXEL);
      if (m) {
        plo = min(plo, lane_bcast(lo, (int)__builtin_ctzll(m)));
        phi = max(phi, lane_bcast(hi, 63 - (int)__builtin_clzll(m)));
      }
    }
    // dense weight row: pixel lane accumulates the samples that touch it, in sample order.  Column tables are
    // relative to the bin's first column, row tables are indexed by absolute row.
    const int origin = is_x ? (phi >= plo ? plo : 0) : 0;
    for (int p0 = 0; p0 < plan_pad(size); p0 += 64) {
      const int px = origin + p0 + lane;
      float wgt = 0.f;
      if (origin + p0 <= phi && origin + p0 + 63 >= plo) {
        for (int i0 = 0; i0 < grid; i0 += 64) {
          int slo, shi;
          float swl, swh;
          sample(i0 + lane, slo, shi, swl, swh);
          const int n = min(64, grid - i0);
          for (int i = 0; i < n; ++i) {
            wgt += (lane_bcast(slo, i) == px) ? lane_bcast(swl, i) : 0.f;
            wgt += (lane_bcast(shi, i) == px) ? lane_bcast(swh, i) : 0.f;
          }
        }
      }
      table[p0 + lane] = wgt;
    }
  }
  if (lane == 0) { s_lo[wave] = plo; s_hi[wave] = phi; }
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 2 * P; ++w) base += s_count[w];
  const int pieces = flag == ROI_SKIP ? 0 : (flag == ROI_HEAVY ? P : 1);
  if (r == num_rois - 1 && threadIdx.x == 0) head->total_items = (base + pieces) * nslices;
  // item descriptors: thread t writes (piece, slice) = (t / nslices, t % nslices)
  for (int t = threadIdx.x; t < pieces * nslices; t += blockDim.x) {
    const int sp = t / nslices, slice = t - sp * nslices;
    RoiItem it;
    it.r = r; it.sp = sp; it.slice = slice; it.flag = flag;
    it.inv_count = inv_count;
    it.bimg = bimg;
    int y0 = H, y1 = -1;
#pragma unroll
    for (int ph = 0; ph < P; ++ph)
      if (flag != ROI_HEAVY || ph == sp) { y0 = min(y0, s_lo[P + ph]); y1 = max(y1, s_hi[P + ph]); }
    it.y0 = flag == ROI_DEAD ? 0 : y0;
    it.y1 = flag == ROI_DEAD ? -1 : y1;
#pragma unroll
    for (int pw = 0; pw < 8; ++pw) {
      it.xlo[pw] = pw < P ? s_lo[pw] : 0;
      it.xhi[pw] = pw < P ? s_hi[pw] : -1;
      it.pad[pw] = 0;
    }
    items[(size_t)(base + sp) * nslices + slice] = it;
  }
}

template <int G, bool EPI>
__global__ __launch_bounds__(256) void roi_align_fwd_planned(const float* __restrict__ feat, int H, int W, int C4,
                                                             int S4, int off4, int plan_slices,
                                                             int nslices, const RoiPlanHead* __restrict__ head,
                                                             const RoiItem* __restrict__ items,
                                                             const float* __restrict__ wxt, const float* __restrict__ wyt,
                                                             float* __restrict__ out, RoiEpilogue epi) {
  constexpr int P = PLAN_P;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpad = plan_pad(W), hpad = plan_pad(H);
  // C4 = float4 channel groups this launch pools and writes; the map has S4 groups per pixel and the launch reads groups
  // [off4, off4 + C4) of it (S4 == C4, off4 == 0: the whole map).  plan_slices: item descriptors per piece in the plan
  // (== nslices when the plan was built for this launch alone, 1 for a plan shared by launches over channel sub-ranges)
  const int slice = blockIdx.x % nslices;              // this workgroup's channel slice
  const int n_slice = head->total_items / plan_slices; // items of one slice
  const int nt = n_slice * P;                          // wave-items of the slice: t -> (item t / P, column bin t % P)
  const int stride = (gridDim.x / nslices) * 4;        // waves working on the slice
  const int c4 = slice * 64 + lane;
  const bool active = c4 < C4;
  const int cl = min(c4, C4 - 1);                      // inactive lanes read the last channel group and never store
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // epilogue terms of this lane's four channels (the slice of a workgroup never changes)
  float4 esc = make_float4(1.f, 1.f, 1.f, 1.f), esh = zero4;
  if (EPI) {
    if (epi.scale) esc = reinterpret_cast<const float4*>(epi.scale)[cl];
    if (epi.shift) esh = reinterpret_cast<const float4*>(epi.shift)[cl];
  }
  auto finish = [&](float4 v) {
    if (EPI) {
      v.x = v.x * esc.x + esh.x; v.y = v.y * esc.y + esh.y; v.z = v.z * esc.z + esh.z; v.w = v.w * esc.w + esh.w;
      if (epi.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    }
    return v;
  };

  struct Desc { int r, sp, flag, y0, y1, bimg, xlo, xhi, pw; float inv_count; };
  struct Tab { float wx; float wy[P]; };
  auto load_desc = [&](int t, Desc& d) {               // scalar loads: addresses depend on t only
    d.flag = ROI_SKIP;
    if (t < nt) {
      const int e = t / P;
      d.pw = t - e * P;
      const RoiItem* it = items + (size_t)e * plan_slices + (plan_slices == 1 ? 0 : slice);
      d.r = it->r; d.sp = it->sp; d.flag = it->flag; d.y0 = it->y0; d.y1 = it->y1; d.bimg = it->bimg;
      d.inv_count = it->inv_count;
      d.xlo = it->xlo[d.pw]; d.xhi = it->xhi[d.pw];
    }
  };
  auto load_tab = [&](const Desc& d, Tab& tb) {        // vector loads: first column chunk, first row chunk
    if (d.flag == ROI_LIGHT || d.flag == ROI_HEAVY) {
      tb.wx = wxt[((size_t)d.r * P + d.pw) * wpad + lane];
      const int yc = d.y0 & ~63;
#pragma unroll
      for (int ph = 0; ph < P; ++ph) {
        const int bin = d.flag == ROI_HEAVY ? d.sp : ph;
        tb.wy[ph] = (d.flag == ROI_LIGHT || ph == 0) ? wyt[((size_t)d.r * P + bin) * hpad + min(yc + lane, hpad - 1)] : 0.f;
      }
    }
  };

  int t = (blockIdx.x / nslices) * 4 + wave;
  Desc d_cur, d_nxt;
  Tab t_cur, t_nxt;
  load_desc(t, d_cur);
  load_desc(t + stride, d_nxt);
  load_tab(d_cur, t_cur);
  for (; t < nt; t += stride) {
    load_tab(d_nxt, t_nxt);                            // tables of the next wave-item
    Desc d_nn;
    load_desc(t + 2 * stride, d_nn);                   // descriptor of the one after
    // ---- pool wave-item (d_cur, t_cur) ---------------------------------------------------------------
    {
      const Desc& d = d_cur;
      const int pw = d.pw;
      const bool heavy = d.flag == ROI_HEAVY;
      const int ph_begin = heavy ? d.sp : 0, nph = heavy ? 1 : P;
      float4* ob = reinterpret_cast<float4*>(out) + ((size_t)d.r * P * P + pw) * C4 + c4;   // + ph * P * C4
      if (d.flag == ROI_DEAD) {
        if (active)
#pragma unroll
          for (int ph = 0; ph < P; ++ph) ob[(size_t)ph * P * C4] = finish(zero4);
      } else {
        f32x2 acc[P][2];
#pragma unroll
        for (int ph = 0; ph < P; ++ph) acc[ph][0] = acc[ph][1] = f32x2{0.f, 0.f};
        const int ncols_all = d.xhi - d.xlo + 1;
        if (ncols_all > 0 && d.y1 >= d.y0) {
          const float4* fbu = reinterpret_cast<const float4*>(feat) + (size_t)d.bimg * H * W * S4 + off4;   // wave-uniform
          for (int xc = 0; xc < ncols_all; xc += 64) {           // column chunks of 64 (one weight per lane)
            const int ncols = min(64, ncols_all - xc), xlo = d.xlo + xc;
            const float wx = xc == 0 ? t_cur.wx : wxt[((size_t)d.r * P + pw) * wpad + xc + lane];
            const int row_skip = (W - ncols) * S4;  // float4 units from the end of a window row to the next row's start
            for (int yc = d.y0 & ~63; yc <= d.y1; yc += 64) {   // aligned row chunks of 64 (one weight per lane)
              const int ys = max(yc, d.y0), ye = min(yc + 63, d.y1);
              float wy[P];
#pragma unroll
              for (int ph = 0; ph < P; ++ph) {
                wy[ph] = t_cur.wy[ph];
                if (yc != (d.y0 & ~63) && ph < nph)
                  wy[ph] = wyt[((size_t)d.r * P + ph_begin + ph) * hpad + min(yc + lane, hpad - 1)];
              }
              const int total = (ye - ys + 1) * ncols;
              f32x2 T0 = {0.f, 0.f}, T1 = {0.f, 0.f};
              int cy = ys - yc, cx = 0;                 // consume cursor: row relative to yc, column relative to xlo
              int lx = 0, loff = (ys * W + xlo) * S4;   // load cursor: column and float4 offset (wave-uniform)
              auto consume = [&](const float4& v) {
                const float wgt = lane_bcast(wx, cx);
                const f32x2 w2 = {wgt, wgt};
                T0 = __builtin_elementwise_fma(w2, f32x2{v.x, v.y}, T0);
                T1 = __builtin_elementwise_fma(w2, f32x2{v.z, v.w}, T1);
                if (++cx == ncols) {
#pragma unroll
                  for (int ph = 0; ph < P; ++ph) {
                    if (ph >= nph) continue;
                    const float wr = lane_bcast(wy[ph], cy);
                    const f32x2 r2 = {wr, wr};
                    acc[ph][0] = __builtin_elementwise_fma(r2, T0, acc[ph][0]);
                    acc[ph][1] = __builtin_elementwise_fma(r2, T1, acc[ph][1]);
                  }
                  T0 = T1 = f32x2{0.f, 0.f};
                  cx = 0;
                  ++cy;
                }
              };
              int base = 0;
              for (; base + G <= total; base += G) {    // full groups: G independent loads in flight per lane
                float4 v[G];
#pragma unroll
                for (int k = 0; k < G; ++k) {
                  v[k] = (fbu + loff)[cl];
                  loff += S4;
                  if (++lx == ncols) { lx = 0; loff += row_skip; }
                }
#pragma unroll
                for (int k = 0; k < G; ++k) consume(v[k]);
              }
              if (base < total) {                       // tail group
                const int n = total - base;
                float4 v[G];
#pragma unroll
                for (int k = 0; k < G - 1; ++k) {
                  v[k] = zero4;
                  if (k < n) {
                    v[k] = (fbu + loff)[cl];
                    loff += S4;
                    if (++lx == ncols) { lx = 0; loff += row_skip; }
                  }
                }
#pragma unroll
                for (int k = 0; k < G - 1; ++k)
                  if (k < n) consume(v[k]);
              }
            }
          }
        }
        if (active)
#pragma unroll
          for (int ph = 0; ph < P; ++ph)
            if (ph < nph) {
              const float4 v = make_float4(acc[ph][0][0] * d.inv_count, acc[ph][0][1] * d.inv_count,
                                           acc[ph][1][0] * d.inv_count, acc[ph][1][1] * d.inv_count);
              ob[(size_t)(ph_begin + ph) * P * C4] = finish(v);
            }
      }
    }
    d_cur = d_nxt;
    t_cur = t_nxt;
    d_nxt = d_nn;
  }
}


// ------------------------------------------------------------------------------------------------
// Backward of the planned separable form (training: torchvision's roi_align autograd, aligned=False).
//   dfeat[y][x] += 1/count * sum_ph sum_pw Wy[ph][y] * Wx[pw][x] * dout[ph][pw]
// The sample-by-sample backward issues 4 float atomics per sample (49 bins x grid_h x grid_w x 4 per RoI and channel); here
// a wave-item (item, column bin pw) first folds the row bins of a window row into T = sum_ph Wy[ph][y] * dout[ph][pw]
// (registers, wave-uniform weights from the plan tables) and then adds Wx[pw][x] * T once per window pixel of that column
// bin: one atomic per (RoI, pixel of a column bin, channel) whatever the sampling grid.  Same plan kernel, same item
// list and tables as the forward; float atomics because windows of different RoIs (and neighbouring column bins) overlap.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void roi_align_bwd_planned(const float* __restrict__ dout, int H, int W, int C4, int nslices,
                                                             const RoiPlanHead* __restrict__ head,
                                                             const RoiItem* __restrict__ items,
                                                             const float* __restrict__ wxt, const float* __restrict__ wyt,
                                                             float* __restrict__ dfeat) {
  constexpr int P = PLAN_P;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpad = plan_pad(W), hpad = plan_pad(H);
  const int slice = blockIdx.x % nslices;
  const int n_slice = head->total_items / nslices;
  const int nt = n_slice * P;
  const int stride = (gridDim.x / nslices) * 4;
  for (int t = (blockIdx.x / nslices) * 4 + wave; t < nt; t += stride) {
    const int e = t / P, pw = t - e * P;
    const RoiItem* it = items + (size_t)e * nslices + slice;
    const int flag = it->flag;
    if (flag != ROI_LIGHT && flag != ROI_HEAVY) continue;
    const int r = it->r, sp = it->sp, y0 = it->y0, y1 = it->y1, bimg = it->bimg;
    const float inv_count = it->inv_count;
    const int xlo_all = it->xlo[pw], xhi_all = it->xhi[pw];
    const bool heavy = flag == ROI_HEAVY;
    const int ph_begin = heavy ? sp : 0, nph = heavy ? 1 : P;
    const int ncols_all = xhi_all - xlo_all + 1;
    if (ncols_all <= 0 || y1 < y0) continue;
    // the item's output gradients, scaled by 1/count once.  Lane l owns channels slice*256 + q*64 + l (q = 0..3): every
    // atomic instruction of the wave then covers 256 contiguous bytes of a pixel (two cache lines), not 64 lanes x 16 bytes
    float g[P][4];
    const int cbase = slice * 256 + lane;
    const int C = C4 * 4;
    const float* gb = dout + ((size_t)r * P * P + pw) * C;
#pragma unroll
    for (int ph = 0; ph < P; ++ph)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = cbase + q * 64;
        g[ph][q] = (ph < nph && c < C) ? gb[(size_t)(ph_begin + ph) * P * C + c] * inv_count : 0.f;
      }
    float* fb = dfeat + (size_t)bimg * H * W * C;
    for (int xc = 0; xc < ncols_all; xc += 64) {
      const int ncols = min(64, ncols_all - xc), xlo = xlo_all + xc;
      const float wx = wxt[((size_t)r * P + pw) * wpad + xc + lane];
      for (int yc = y0 & ~63; yc <= y1; yc += 64) {
        const int ys = max(yc, y0), ye = min(yc + 63, y1);
        float wy[P];
#pragma unroll
        for (int ph = 0; ph < P; ++ph)
          wy[ph] = ph < nph ? wyt[((size_t)r * P + ph_begin + ph) * hpad + min(yc + lane, hpad - 1)] : 0.f;
        for (int y = ys; y <= ye; ++y) {
          float T[4] = {0.f, 0.f, 0.f, 0.f};
          bool any = false;
#pragma unroll
          for (int ph = 0; ph < P; ++ph) {
            if (ph >= nph) continue;
            const float wr = lane_bcast(wy[ph], y - yc);
            any = any || wr != 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) T[q] += wr * g[ph][q];
          }
          if (!any) continue;                                   // wave-uniform: a row no row bin of the item touches
          float* row = fb + ((size_t)y * W + xlo) * C + cbase;
          for (int x = 0; x < ncols; ++x) {
            const float wc = lane_bcast(wx, x);
            if (wc == 0.f) continue;                            // wave-uniform
            float* px = row + (size_t)x * C;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (cbase + q * 64 < C) atomicAdd(px + q * 64, wc * T[q]);
          }
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Map-resident form — the fastest path (P == 7, the map slice fits the LDS).
//
// What bounded the planned pair above: every RoI reads its own window through the vector L1 (326 MB per launch for the
// 300 RoIs of a 1000x600 frame against a 9.8 MB map), one dependent load chain per work item, a static schedule, and a
// separate plan launch (6.3 us).  Here a workgroup keeps the WHOLE feature map for 16 channels in LDS (38 x 63 pixels x
// 64 B = 150 KB of the CU's 160 KB) and pools every RoI of its share from there: the map leaves L2 once per workgroup
// (4 workgroups share a slice), all window re-reads are LDS reads (256 B/clk/CU instead of 64 B/clk through the L1), the
// plan is built per RoI pair by the consuming wave, and RoI pairs are dealt dynamically from an LDS counter.
//
//   grid    (C/16 channel slices x G RoI groups, images); workgroup = 8 waves.  Slices that share 128-byte output lines
//           sit on the same XCD (speed only).
//   phase 1 the slice is copied global -> LDS with LDS-DMA (global_load_lds_dwordx4: 16 pixels x 64 B per wave
//           instruction), pixel s stored at slot s ^ swz(s >> 2) (a permutation inside each group of 4 slots, so that
//           the pixels 4 lane groups read together do not all sit in the same quarter of the banks).
//   phase 2 a wave takes a PAIR of RoIs (one per half wave).  14 of its 16 four-lane groups are (RoI, column bin pw)
//           tasks: lane q of a group owns channels 4q..4q+3, a group reads one pixel = 64 B per ds_read_b128.
//           tables: lanes evaluate the samples of a bin in parallel (the library's coordinate arithmetic) and add their
//           weights into compact per-bin tables in LDS (ds_add_f32 after a zero store; one wave = program order).
//           pooling: rows of the window ascending; inside a row the columns of bin pw ascending:
//           rowsum = sum_x Wx[pw][x] F[y][x]; then acc[ph] += Wy[ph][y] * rowsum for the 7 row bins (Wy dense over the
//           window rows, one lane per row, broadcast with v_readlane).  Each map pixel of a bin column range is read once
//           per (RoI, pw), whatever the number of row bins it feeds.
//           A RoI whose bin ranges exceed the compact tables (RoIs far larger than the map) is pooled sample by sample
//           from the LDS map (same arithmetic as the generic kernel).
// Output: 64-byte segments (16 channels of a bin), the fabric's request size.
// ------------------------------------------------------------------------------------------------
constexpr int RES_WAVES = 8;
constexpr int RES_FX = 11;        // columns a column bin may touch (map width <= 63: bin <= 9 px -> 11 pixels)
constexpr int RES_FY = 8;         // rows a row bin may touch
struct ResRoiTab {
  float wx[PLAN_P][RES_FX];
  float wy[PLAN_P][RES_FY];
  short xlo[PLAN_P], ylo[PLAN_P];
  unsigned char xcnt[PLAN_P], ycnt[PLAN_P];
  short pad[1];
};
static_assert(sizeof(ResRoiTab) % 16 == 0, "table block keeps 16-byte alignment");

__host__ __device__ inline size_t res_map_bytes(int h, int w) { return (size_t)h * w * 64; }
__host__ __device__ inline size_t res_lds_bytes(int h, int w) {
  // the tail of the last 1 KB DMA chunk (up to 15 pixel slots past the map) overlaps the table area, which is first
  // written after the barrier that ends phase 1
  return frcnn::align_up(res_map_bytes(h, w), 16) + (size_t)RES_WAVES * 2 * sizeof(ResRoiTab) + 16;
}

// Column x of a row is stored at column x ^ h(x >> 2) of the same row (h in 0..3: a permutation inside each aligned
// group of 4 columns, an involution; the last, partial group of a row keeps its order).  The four tasks a ds_read_b128
// serves together are column bins of ONE RoI in ONE row; frame-wide RoIs have bins ~8 px wide, so without the
// permutation their first columns all fall into the same quarter of the banks (4-way conflict).
__device__ __forceinline__ int res_col(int x, int W) {
  const int t = x >> 2;
  return x < (W & ~3) ? x ^ ((t ^ (t >> 1)) & 3) : x;
}

__global__ __launch_bounds__(64 * RES_WAVES) void roi_align_fwd_resident(
    const float* __restrict__ feat, int H, int W, int C, const float* __restrict__ rois, const int* __restrict__ roi_count,
    int num_rois, int rois_per_image, float spatial_scale, int sampling_ratio, const int* __restrict__ level_of_roi,
    int level, int nslices, int ngroups, float* __restrict__ out, int dbg) {
  constexpr int P = PLAN_P;
  extern __shared__ __attribute__((aligned(16))) unsigned char res_smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int img = blockIdx.y;
  // (slice, group) of this workgroup: with nslices % 8 == 0 XCD x (= blockIdx.x % 8) owns slices [x*nslices/8, ...)
  int slice, group;
  if ((nslices & 7) == 0) {
    const int per_xcd = nslices >> 3, k = blockIdx.x >> 3;
    slice = (blockIdx.x & 7) * per_xcd + k % per_xcd;
    group = k / per_xcd;
  } else {
    slice = blockIdx.x % nslices;
    group = blockIdx.x / nslices;
  }
  const int npix = H * W;
  float* smap = reinterpret_cast<float*>(res_smem);
  unsigned char* tab_area = res_smem + frcnn::align_up(res_map_bytes(H, W), 16);
  int* counter = reinterpret_cast<int*>(tab_area + (size_t)RES_WAVES * 2 * sizeof(ResRoiTab));
  if (threadIdx.x == 0) *counter = 0;

  // ---- phase 1: map slice -> LDS ---------------------------------------------------------------------------
  {
    const float* fimg = feat + (size_t)img * npix * C + slice * 16 + (lane & 3) * 4;
    const int nchunks = (dbg & 1) ? 0 : (npix + 15) >> 4;
    for (int c = wave; c < nchunks; c += RES_WAVES) {
      const int slot = min(c * 16 + (lane >> 2), npix - 1);   // slots past the map repeat the last pixel (never read)
      const int sy = slot / W, sx = slot - sy * W;
      const int s = sy * W + res_col(sx, W);
      typedef const __attribute__((address_space(1))) void* gptr;
      typedef __attribute__((address_space(3))) void* lptr;
      __builtin_amdgcn_global_load_lds((gptr)(fimg + (size_t)s * C), (lptr)(smap + (size_t)c * 256), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  // ---- phase 2 -----------------------------------------------------------------------------------------------
  const int half = lane >> 5, sub = lane & 31;
  const int pw = (lane >> 2) & 7, q = lane & 3;
  const bool slot_live = pw < P;
  const int pwc = min(pw, P - 1);
  const int r_first = rois_per_image > 0 ? img * rois_per_image : 0;
  const int r_end = rois_per_image > 0 ? min(num_rois, r_first + rois_per_image) : num_rois;
  const int live_n = roi_count ? min(roi_count[rois_per_image > 0 ? img : 0], r_end - r_first) : r_end - r_first;
  const int n_mine = (r_end - r_first - group + ngroups - 1) / ngroups;       // RoIs r_first + group + ngroups * k
  const int npairs = (max(n_mine, 0) + 1) >> 1;
  ResRoiTab* tabs = reinterpret_cast<ResRoiTab*>(tab_area) + wave * 2;
  ResRoiTab* T = tabs + half;
  const int C4 = C >> 2;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool is_x = sub < 16;
  const int si = sub & 15;                                    // sample index of this lane in the table passes

  for (;;) {
    int p = 0;
    if (lane == 0) p = atomicAdd(counter, 1);
    p = __builtin_amdgcn_readfirstlane(p);
    if (p >= npairs) break;
    // ---- RoI of this half wave ----
    const int k = 2 * p + half;
    const int r = r_first + group + ngroups * k;
    enum { ST_SKIP = 0, ST_TABLE = 1, ST_ZERO = 2, ST_DIRECT = 3 };
    int state = ST_SKIP;
    float rsw = 0.f, rsh = 0.f, bw = 1.f, bh = 1.f, inv_count = 0.f;
    int gw = 0, gh = 0;
    if (k < n_mine) {
      const float* roi = rois + (size_t)r * 5;
      const bool lvl_ok = !level_of_roi || level_of_roi[r] == level;
      const bool dead = (r - r_first) >= live_n;
      const int bimg = (int)roi[0];
      if (lvl_ok && dead) state = (img == 0 || rois_per_image > 0) ? ST_ZERO : ST_SKIP;
      else if (lvl_ok && (rois_per_image > 0 || bimg == img)) {
        state = ST_TABLE;
        rsw = roi[1] * spatial_scale; rsh = roi[2] * spatial_scale;
        const float rew = roi[3] * spatial_scale, reh = roi[4] * spatial_scale;
        const float roi_width = fmaxf(rew - rsw, 1.0f), roi_height = fmaxf(reh - rsh, 1.0f);
        bh = roi_height / (float)P; bw = roi_width / (float)P;
        gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)P);
        gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)P);
        inv_count = 1.0f / (float)(gh * gw);
        if (gw > 16 || gh > 16) state = ST_DIRECT;
      }
    }
    if (__ballot(state != ST_SKIP) == 0ull) continue;

    // ---- compact weight tables: pass t builds column bin t (lanes sub 0..15 = samples) and row bin t (sub 16..31) ----
    if (!(dbg & 2)) {
      const float start = is_x ? rsw : rsh, bin = is_x ? bw : bh;
      const int grid = is_x ? gw : gh, size = is_x ? W : H, cap = is_x ? RES_FX : RES_FY;
      bool over = false;
#pragma unroll
      for (int t = 0; t < P; ++t) {
        int lo = 0, hi = 0;
        float wl = 0.f, wh = 0.f;
        bool valid = false;
        if (state == ST_TABLE && si < grid) valid = axis_sample_pt(start, bin, t, si, grid, size, lo, hi, wl, wh);
        const unsigned long long m = __ballot(valid);
        const unsigned f = (unsigned)(m >> (lane & 48)) & 0xFFFFu;       // the 16 lanes of my (RoI, axis)
        const int first = f ? __builtin_ctz(f) : 0, last = f ? 31 - __builtin_clz(f) : 0;
        // sample coordinates are non-decreasing: the first valid sample has the smallest pixel, the last the largest
        const int lo_first = __shfl(lo, (lane & 48) + first), hi_last = __shfl(hi, (lane & 48) + last);
        const int cnt = f ? hi_last - lo_first + 1 : 0;
        over = over || cnt > cap;
        float* row = is_x ? T->wx[t] : T->wy[t];
        if (si < cap) row[si] = 0.f;
        if (si == 0) {
          (is_x ? T->xlo : T->ylo)[t] = (short)lo_first;
          (is_x ? T->xcnt : T->ycnt)[t] = (short)min(cnt, cap);
        }
        if (valid && cnt <= cap) {
          atomicAdd(&row[lo - lo_first], wl);                 // ds_add_f32: same-wave LDS operations stay in order
          atomicAdd(&row[hi - lo_first], wh);
        }
      }
      // a bin that does not fit the compact tables sends its whole RoI to the sample-by-sample path
      const unsigned long long mo = __ballot(over);
      if (state == ST_TABLE && ((mo >> (lane & 32)) & 0xFFFFFFFFull) != 0ull) state = ST_DIRECT;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    f32x2 acc[P][2];
#pragma unroll
    for (int ph = 0; ph < P; ++ph) acc[ph][0] = acc[ph][1] = f32x2{0.f, 0.f};

    // ---- table path ----
    if (!(dbg & 8) && __ballot(state == ST_TABLE) != 0ull) {
      // window rows of my RoI and the dense row-weight registers: lane l = row y0 + l
      int y0 = H, y1 = -1;
#pragma unroll
      for (int ph = 0; ph < P; ++ph) {
        const int c = T->ycnt[ph], l = T->ylo[ph];
        if (c > 0) { y0 = min(y0, l); y1 = max(y1, l + c - 1); }
      }
      const bool tab = state == ST_TABLE;
      const int nrows = tab && y1 >= y0 ? y1 - y0 + 1 : 0;
      const int y0a = __builtin_amdgcn_readlane(y0, 0), y0b = __builtin_amdgcn_readlane(y0, 32);
      float ta[P], tb[P];
#pragma unroll
      for (int ph = 0; ph < P; ++ph) {
        const int ka = lane - ((int)tabs[0].ylo[ph] - y0a), kb = lane - ((int)tabs[1].ylo[ph] - y0b);
        ta[ph] = (ka >= 0 && ka < (int)tabs[0].ycnt[ph]) ? tabs[0].wy[ph][ka] : 0.f;
        tb[ph] = (kb >= 0 && kb < (int)tabs[1].ycnt[ph]) ? tabs[1].wy[ph][kb] : 0.f;
      }
      const int my_xlo = T->xlo[pwc];
      const int my_xcnt = (tab && slot_live) ? (int)T->xcnt[pwc] : 0;
      // per column of my bin: weight and LDS byte offset inside a row (q-th 16 bytes of the pixel).  Columns past the
      // bin's range have weight 0 and repeat the last column's address, so the row loop needs no lane predicate.
      float wxr[RES_FX];
      int off[RES_FX];
#pragma unroll
      for (int j = 0; j < RES_FX; ++j) {
        wxr[j] = j < my_xcnt ? T->wx[pwc][j] : 0.f;
        const int x = my_xcnt > 0 ? my_xlo + min(j, my_xcnt - 1) : 0;
        off[j] = res_col(x, W) * 64 + q * 16;
        // a four-lane group without a task (slots 7 and 15) repeats the addresses of slot 4 / 12, which the same
        // ds_read_b128 service group reads anyway: identical addresses are one bank access
        const int borrowed = __shfl(off[j], lane - 12);
        off[j] = slot_live ? off[j] : borrowed;
      }
      int cnt_max = my_xcnt, rows_max = nrows;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        cnt_max = max(cnt_max, __shfl_xor(cnt_max, o));
        rows_max = max(rows_max, __shfl_xor(rows_max, o));
      }
      cnt_max = __builtin_amdgcn_readfirstlane(cnt_max);
      rows_max = __builtin_amdgcn_readfirstlane(rows_max);
      const int ybase = nrows > 0 ? y0 : 0;
      const unsigned char* smap_b = reinterpret_cast<const unsigned char*>(smap);
      auto rows = [&](auto nj_tag) {
        constexpr int NJ = decltype(nj_tag)::value;
        for (int i = 0; i < rows_max; ++i) {
          // a half wave whose window is shorter re-reads its last row; the row weights past its window are zero
          const int rb = min(ybase + i, H - 1) * W * 64;
          float4 v[NJ];
#pragma unroll
          for (int j = 0; j < NJ; ++j) v[j] = *reinterpret_cast<const float4*>(smap_b + rb + off[j]);
          f32x2 r0 = {0.f, 0.f}, r1 = {0.f, 0.f};
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const f32x2 w2 = {wxr[j], wxr[j]};
            r0 = __builtin_elementwise_fma(w2, f32x2{v[j].x, v[j].y}, r0);
            r1 = __builtin_elementwise_fma(w2, f32x2{v[j].z, v[j].w}, r1);
          }
#pragma unroll
          for (int ph = 0; ph < P; ++ph) {
            const float wa = lane_bcast(ta[ph], i), wb = lane_bcast(tb[ph], i);
            const float wr = half ? wb : wa;
            const f32x2 w2 = {wr, wr};
            acc[ph][0] = __builtin_elementwise_fma(w2, r0, acc[ph][0]);
            acc[ph][1] = __builtin_elementwise_fma(w2, r1, acc[ph][1]);
          }
        }
      };
      if (cnt_max <= 4) rows(std::integral_constant<int, 4>{});
      else if (cnt_max <= 8) rows(std::integral_constant<int, 8>{});
      else rows(std::integral_constant<int, RES_FX>{});
    }

    // ---- sample-by-sample path (bins too large for the tables) ----
    if (__ballot(state == ST_DIRECT) != 0ull) {
      if (state == ST_DIRECT && slot_live) {
        for (int ph = 0; ph < P; ++ph) {
          float4 a = zero4;
          for (int iy = 0; iy < gh; ++iy) {
            int yl, yh;
            float hy, ly;
            if (!axis_sample_pt(rsh, bh, ph, iy, gh, H, yl, yh, hy, ly)) continue;
            for (int ix = 0; ix < gw; ++ix) {
              int xl, xh;
              float hx, lx;
              if (!axis_sample_pt(rsw, bw, pw, ix, gw, W, xl, xh, hx, lx)) continue;
              const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
              const float4 v1 = *reinterpret_cast<const float4*>(smap + (size_t)(yl * W + res_col(xl, W)) * 16 + q * 4);
              const float4 v2 = *reinterpret_cast<const float4*>(smap + (size_t)(yl * W + res_col(xh, W)) * 16 + q * 4);
              const float4 v3 = *reinterpret_cast<const float4*>(smap + (size_t)(yh * W + res_col(xl, W)) * 16 + q * 4);
              const float4 v4 = *reinterpret_cast<const float4*>(smap + (size_t)(yh * W + res_col(xh, W)) * 16 + q * 4);
              a.x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
              a.y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
              a.z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
              a.w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
            }
          }
          // (dynamic ph: written through a switch-free select chain below would cost registers; store directly)
          float4* ob = reinterpret_cast<float4*>(out) + ((size_t)(r * P + ph) * P + pw) * C4 + slice * 4 + q;
          *ob = make_float4(a.x * inv_count, a.y * inv_count, a.z * inv_count, a.w * inv_count);
        }
      }
    }

    // ---- stores: bins (ph, pw) of my RoI, 16 channels = 64 B per four-lane group ----
    if (!(dbg & 4) && slot_live && (state == ST_TABLE || state == ST_ZERO)) {
      float4* ob = reinterpret_cast<float4*>(out) + ((size_t)r * P * P + pw) * C4 + slice * 4 + q;
#pragma unroll
      for (int ph = 0; ph < P; ++ph)
        ob[(size_t)ph * P * C4] = make_float4(acc[ph][0][0] * inv_count, acc[ph][0][1] * inv_count,
                                              acc[ph][1][0] * inv_count, acc[ph][1][1] * inv_count);
    }
  }
}

size_t plan_bytes(int h, int w, int c, int num_rois) {
  const int nslices = (c / 4 + 63) / 64;
  return sizeof(RoiPlanHead) + (size_t)num_rois * PLAN_P * nslices * sizeof(RoiItem) +
         (size_t)num_rois * PLAN_P * (plan_pad(w) + plan_pad(h)) * sizeof(float);
}

bool planned_ok(int pooled) { return pooled == PLAN_P; }

}  // namespace

// tuning hook: 0 = automatic (map-resident kernel when the map slice fits the LDS, else the planned pair), 1 = generic,
// 2 = generic with XCD channel slices, 3 / 4 = planned kernel with 8 / 4 loads in flight per lane (needs a workspace),
// 5 = map-resident kernel (an error when it does not apply)
static int g_roi_variant = 0;
// estimated loads per lane above which a RoI is split into its P row bins (variant >= 100 sets it: tuning only)
static int g_roi_heavy_loads = 64;
constexpr bool RES_AUTO = false;   // flipped when the map-resident kernel beats the planned pair (profiles/r03_roi_align.md)
static int g_res_dbg = 0;      // ablation mask of the map-resident kernel (variant 1000 + mask): 1 no map load, 2 no tables,
                               // 4 no stores, 8 no row loop - timing experiments only, results are then wrong
extern "C" int frcnn_roi_align_set_variant(int v) {
  if (v >= 1000) { g_res_dbg = v - 1000; return FRCNN_OK; }
  if (v >= 100) { g_roi_heavy_loads = v; return FRCNN_OK; }
  g_roi_variant = v;
  return FRCNN_OK;
}

unsigned long long frcnn::roi_settings_word() {
  return (unsigned long long)(unsigned)g_roi_variant | ((unsigned long long)(unsigned)g_roi_heavy_loads << 16) |
         ((unsigned long long)(unsigned)g_res_dbg << 40);
}

static bool resident_ok(int h, int w, int c, int pooled) {
  return pooled == PLAN_P && c % 16 == 0 && h <= 64 && w <= 1024 && res_lds_bytes(h, w) <= (size_t)160 * 1024;
}

extern "C" size_t frcnn_roi_align_fwd_ws_bytes(int h, int w, int c, int num_rois, int pooled) {
  if (h <= 0 || w <= 0 || c <= 0 || num_rois <= 0 || !planned_ok(pooled)) return 0;
  return plan_bytes(h, w, c, num_rois);
}

extern "C" int frcnn_roi_align_fwd_affine(const float* feat, int n, int h, int w, int c, const float* rois,
                                          const int* roi_count, int num_rois, int rois_per_image, int pooled,
                                          float spatial_scale, int sampling_ratio, const int* level_of_roi, int level,
                                          float* out, const float* scale, const float* shift, int relu, void* ws,
                                          size_t ws_bytes, void* stream_) {
  const RoiEpilogue epi{scale, shift, relu};
  const bool has_epi = scale || shift || relu;
  FRCNN_REQUIRE(feat && rois && out && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && num_rois > 0 && pooled > 0,
                "roi_align_fwd: bad arguments (c%%4==0)");
  FRCNN_REQUIRE(rois_per_image == 0 || (rois_per_image > 0 && (long)rois_per_image * n >= num_rois),
                "roi_align_fwd: rois_per_image %d x %d images < %d rois", rois_per_image, n, num_rois);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FRCNN_REQUIRE(!(has_epi && g_roi_variant == 5), "roi_align_fwd_affine: the map-resident kernel has no epilogue");
  if (g_roi_variant == 5 && !resident_ok(h, w, c, pooled))
    return frcnn::fail(FRCNN_ERR_ARG, "roi_align_fwd: the map-resident kernel needs pooled 7, c %% 16 == 0, h <= 64 and "
                       "h*w*64 B of LDS (got %dx%dx%d)", h, w, c);
  if ((g_roi_variant == 5 || (g_roi_variant == 0 && RES_AUTO && !has_epi)) && resident_ok(h, w, c, pooled)) {
    const size_t lds = res_lds_bytes(h, w);
    static std::atomic<size_t> configured{0};
    if (lds > configured.load()) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&roi_align_fwd_resident),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return frcnn::fail(FRCNN_ERR_LAUNCH, "roi_align_fwd: set LDS size: %s", hipGetErrorString(e));
      configured.store(lds);
    }
    const int nslices = c / 16;
    // one workgroup per CU in total: RoI groups share a channel slice (each reloads the map slice from L2)
    const int per_image_rois = rois_per_image > 0 ? rois_per_image : num_rois;
    int ngroups = std::max(1, 256 / (nslices * n));
    ngroups = std::max(1, std::min(ngroups, per_image_rois / 16));
    hipLaunchKernelGGL(roi_align_fwd_resident, dim3((unsigned)(nslices * ngroups), (unsigned)n), dim3(64 * RES_WAVES), lds,
                       stream, feat, h, w, c, rois, roi_count, num_rois, rois_per_image, spatial_scale, sampling_ratio,
                       level_of_roi, level, nslices, ngroups, out, g_res_dbg);
    return frcnn::check_launch("roi_align_fwd_resident");
  }
  // the kernels below index the image through the RoI's batch column and read one live count
  FRCNN_REQUIRE(rois_per_image == 0 || n == 1, "roi_align_fwd: per-image RoI blocks need the map-resident kernel");
  const bool want_planned = g_roi_variant == 0 || g_roi_variant == 3 || g_roi_variant == 4;
  if (want_planned && planned_ok(pooled) && ws && ws_bytes >= plan_bytes(h, w, c, num_rois)) {
    const int c4 = c / 4, nslices = (c4 + 63) / 64;
    RoiPlanHead* head = static_cast<RoiPlanHead*>(ws);
    RoiItem* items = reinterpret_cast<RoiItem*>(head + 1);
    float* wxt = reinterpret_cast<float*>(items + (size_t)num_rois * PLAN_P * nslices);
    float* wyt = wxt + (size_t)num_rois * PLAN_P * plan_pad(w);
    hipLaunchKernelGGL(roi_plan_kernel, dim3((unsigned)num_rois), dim3(64 * 2 * PLAN_P), 0, stream, h, w, rois, roi_count,
                       num_rois, spatial_scale, sampling_ratio, level_of_roi, level, nslices, g_roi_heavy_loads, head, items, wxt,
                       wyt);
    int rc = frcnn::check_launch("roi_plan_kernel");
    if (rc != FRCNN_OK) return rc;
    // persistent grid of 4-wave workgroups, as many as are resident at once: 88 VGPRs with 4 loads in flight per lane =
    // 5 waves per SIMD = 5 workgroups per CU (measured best: 22.9 us; 8 loads in flight need 104 VGPRs = 4 per CU: 25.5 us).  A multiple of nslices so that a workgroup keeps its slice;
    // never more waves than wave-items in the worst case
    const long max_wave_items = (long)num_rois * nslices * PLAN_P * PLAN_P;
    const bool g8 = g_roi_variant == 3;                  // variant 3: 8 loads in flight (104 VGPRs, 4 workgroups per CU)
    long nwg = 256 * ((g8 || has_epi) ? 4 : 5);      // the epilogue form holds 108 VGPRs: 4 waves per SIMD
    nwg = std::min(nwg, (max_wave_items + 3) / 4);
    nwg = std::max<long>(nslices, nwg / nslices * nslices);
    if (!g8 && !has_epi)
      hipLaunchKernelGGL((roi_align_fwd_planned<4, false>), dim3((unsigned)nwg), dim3(256), 0, stream, feat, h, w, c4, c4, 0, nslices, nslices,
                         head, items, wxt, wyt, out, epi);
    else if (!g8)
      hipLaunchKernelGGL((roi_align_fwd_planned<4, true>), dim3((unsigned)nwg), dim3(256), 0, stream, feat, h, w, c4, c4, 0, nslices, nslices,
                         head, items, wxt, wyt, out, epi);
    else if (!has_epi)
      hipLaunchKernelGGL((roi_align_fwd_planned<8, false>), dim3((unsigned)nwg), dim3(256), 0, stream, feat, h, w, c4, c4, 0, nslices, nslices,
                         head, items, wxt, wyt, out, epi);
    else
      hipLaunchKernelGGL((roi_align_fwd_planned<8, true>), dim3((unsigned)nwg), dim3(256), 0, stream, feat, h, w, c4, c4, 0, nslices, nslices,
                         head, items, wxt, wyt, out, epi);
    return frcnn::check_launch("roi_align_fwd_planned");
  }
  const size_t total = (size_t)num_rois * pooled * pooled * (c / 4);
  if (g_roi_variant != 1 && c % 32 == 0) {   // 8 channel slices of c/8 channels, one per XCD; grid = multiple of 8
    const size_t per_slice_blocks = std::min<size_t>((total / 8 + 255) / 256, (size_t)1 << 17);
    hipLaunchKernelGGL(roi_align_fwd_nhwc<true>, dim3((unsigned)(per_slice_blocks * 8)), dim3(256), 0, stream, feat, h, w,
                       c / 4, rois, roi_count, num_rois, pooled, spatial_scale, sampling_ratio, level_of_roi, level, out, epi);
    return frcnn::check_launch("roi_align_fwd_nhwc<xcd>");
  }
  const size_t blocks = std::min<size_t>((total + 255) / 256, (size_t)1 << 20);
  hipLaunchKernelGGL(roi_align_fwd_nhwc<false>, dim3((unsigned)blocks), dim3(256), 0, stream, feat, h, w, c / 4, rois,
                     roi_count, num_rois, pooled, spatial_scale, sampling_ratio, level_of_roi, level, out, epi);
  return frcnn::check_launch("roi_align_fwd_nhwc");
}

extern "C" int frcnn_roi_align_fwd(const float* feat, int n, int h, int w, int c, const float* rois, const int* roi_count,
                                   int num_rois, int rois_per_image, int pooled, float spatial_scale, int sampling_ratio,
                                   const int* level_of_roi, int level, float* out, void* ws, size_t ws_bytes,
                                   void* stream_) {
  return frcnn_roi_align_fwd_affine(feat, n, h, w, c, rois, roi_count, num_rois, rois_per_image, pooled, spatial_scale,
                                    sampling_ratio, level_of_roi, level, out, nullptr, nullptr, 0, ws, ws_bytes, stream_);
}

// RoIAlign of ONE map into TWO outputs over channel ranges, through ONE plan: out1 = act1(pool(feat[..., :split_c]) *
// scale + shift), out2 = act2(pool(feat[..., split_c:]) * scale + shift) (scale / shift index the map's channels).  The
// inference path pools layer4[0]'s two projections this way (Network._layer4_projected: conv1 and downsample[0] run as one
// 1024 -> 512 + 2048 convolution; 512 channels with BatchNorm + ReLU, 2048 with BatchNorm).  One plan launch instead of two;
// the pooling stays two launches because a workgroup's channel slice is bound to its XCD (blockIdx % 8): 2 and 8 slices of
// 256 channels keep each XCD on one slice of the map (L2-resident), 10 slices in one launch would put the whole 24.5 MB
// map through every 4 MB L2.  ws: frcnn_roi_align_fwd_ws_bytes(h, w, 4, num_rois, 7) (one descriptor per piece).
extern "C" int frcnn_roi_align_fwd_split(const float* feat, int h, int w, int c, const float* rois, const int* roi_count,
                                         int num_rois, int pooled, float spatial_scale, int sampling_ratio, int split_c,
                                         float* out1, float* out2, const float* scale, const float* shift, int relu1,
                                         int relu2, void* ws, size_t ws_bytes, void* stream_) {
  FRCNN_REQUIRE(feat && rois && out1 && out2 && h > 0 && w > 0 && c > 0 && num_rois > 0 && planned_ok(pooled) &&
                    split_c > 0 && split_c < c && split_c % 256 == 0 && (c - split_c) % 4 == 0,
                "roi_align_fwd_split: bad arguments (pooled 7, 0 < split_c < c, split_c %% 256 == 0, c %% 4 == 0)");
  if (!ws || ws_bytes < plan_bytes(h, w, 4, num_rois))
    return frcnn::fail(FRCNN_ERR_WS, "roi_align_fwd_split: workspace %zu < %zu bytes", ws_bytes, plan_bytes(h, w, 4, num_rois));
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  RoiPlanHead* head = static_cast<RoiPlanHead*>(ws);
  RoiItem* items = reinterpret_cast<RoiItem*>(head + 1);
  float* wxt = reinterpret_cast<float*>(items + (size_t)num_rois * PLAN_P);
  float* wyt = wxt + (size_t)num_rois * PLAN_P * plan_pad(w);
  hipLaunchKernelGGL(roi_plan_kernel, dim3((unsigned)num_rois), dim3(64 * 2 * PLAN_P), 0, stream, h, w, rois, roi_count,
                     num_rois, spatial_scale, sampling_ratio, (const int*)nullptr, -1, 1, g_roi_heavy_loads, head, items, wxt, wyt);
  int rc = frcnn::check_launch("roi_plan_kernel");
  if (rc != FRCNN_OK) return rc;
  const int s4 = c / 4;
  for (int part = 0; part < 2; ++part) {
    const int off4 = part == 0 ? 0 : split_c / 4, c4 = part == 0 ? split_c / 4 : s4 - split_c / 4;
    const int nslices = (c4 + 63) / 64;
    const RoiEpilogue epi{scale ? scale + 4 * off4 : nullptr, shift ? shift + 4 * off4 : nullptr, part == 0 ? relu1 : relu2};
    const long max_wave_items = (long)num_rois * nslices * PLAN_P * PLAN_P;
    long nwg = std::min<long>(256 * 4, (max_wave_items + 3) / 4);
    nwg = std::max<long>(nslices, nwg / nslices * nslices);
    hipLaunchKernelGGL((roi_align_fwd_planned<4, true>), dim3((unsigned)nwg), dim3(256), 0, stream, feat, h, w, c4, s4, off4, 1,
                       nslices, head, items, wxt, wyt, part == 0 ? out1 : out2, epi);
    rc = frcnn::check_launch("roi_align_fwd_planned");
    if (rc != FRCNN_OK) return rc;
  }
  return FRCNN_OK;
}

// Backward through the plan (see roi_align_bwd_planned): dfeat (n,H,W,C) += scatter(dout (R,P,P,C)); dfeat is zero-filled
// by the caller.  ws: frcnn_roi_align_fwd_ws_bytes of the same shape; returns FRCNN_ERR_ARG when the shape has no planned
// form (pooled != 7, c % 4 != 0) - the caller then uses frcnn_roi_align_bwd.
extern "C" int frcnn_roi_align_bwd_planned(const float* dout, int h, int w, int c, const float* rois, const int* roi_count,
                                           int num_rois, int pooled, float spatial_scale, int sampling_ratio,
                                           const int* level_of_roi, int level, float* dfeat, void* ws, size_t ws_bytes,
                                           void* stream_) {
  FRCNN_REQUIRE(dout && rois && dfeat && h > 0 && w > 0 && c > 0 && c % 4 == 0 && num_rois > 0 && planned_ok(pooled),
                "roi_align_bwd_planned: bad arguments (pooled 7, c%%4==0)");
  if (!ws || ws_bytes < plan_bytes(h, w, c, num_rois))
    return frcnn::fail(FRCNN_ERR_WS, "roi_align_bwd_planned: workspace %zu < %zu bytes", ws_bytes, plan_bytes(h, w, c, num_rois));
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int c4 = c / 4, nslices = (c4 + 63) / 64;
  RoiPlanHead* head = static_cast<RoiPlanHead*>(ws);
  RoiItem* items = reinterpret_cast<RoiItem*>(head + 1);
  float* wxt = reinterpret_cast<float*>(items + (size_t)num_rois * PLAN_P * nslices);
  float* wyt = wxt + (size_t)num_rois * PLAN_P * plan_pad(w);
  hipLaunchKernelGGL(roi_plan_kernel, dim3((unsigned)num_rois), dim3(64 * 2 * PLAN_P), 0, stream, h, w, rois, roi_count,
                     num_rois, spatial_scale, sampling_ratio, level_of_roi, level, nslices, g_roi_heavy_loads, head, items, wxt,
                     wyt);
  int rc = frcnn::check_launch("roi_plan_kernel");
  if (rc != FRCNN_OK) return rc;
  const long max_wave_items = (long)num_rois * nslices * PLAN_P * PLAN_P;
  long nwg = std::min<long>(256 * 5, (max_wave_items + 3) / 4);
  nwg = std::max<long>(nslices, nwg / nslices * nslices);
  hipLaunchKernelGGL(roi_align_bwd_planned, dim3((unsigned)nwg), dim3(256), 0, stream, dout, h, w, c4, nslices, head, items,
                     wxt, wyt, dfeat);
  return frcnn::check_launch("roi_align_bwd_planned");
}
