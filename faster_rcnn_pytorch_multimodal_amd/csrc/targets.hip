// Training targets on the device: pairwise IoU, RPN anchor targets, second-stage RoI sampling.
// Replaces lib/utils/bbox.py:5-33 (bbox_overlaps, +1 area convention), lib/layer_utils/anchor_target_layer.py:22-165
// and lib/layer_utils/proposal_target_layer.py:22-262.  Integer/latency work: no matrix cores; counts are integer
// atomics (order-independent), maxima are atomicMax on the (non-negative) float bit patterns, random
// sub-sampling draws a counter-based hash key per candidate and takes the smallest keys (what
// torch.randperm(n)[:k] does in distribution; the reference's own draws depend on the torch RNG stream and are
// not reproducible across devices either).  Built with -ffp-contract=off.
#include "common.h"
#include "box_math.h"
#include "rng.h"

#include <atomic>

using namespace frcnn;

namespace {

// IoU with the +1 convention of lib/utils/bbox.py:22-32
__device__ __forceinline__ float iou_plus1(const float* a, const float* b) {
  const float aa = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
  const float ab = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
  const float iw = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]) + 1.f, 0.f);
  const float ih = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]) + 1.f, 0.f);
  const float ua = aa + ab - iw * ih;
  return iw * ih / ua;
}

// bbox_transform (lib/model/bbox_transform.py:52-70): centre deltas over the box diagonal, log size ratios
__device__ __forceinline__ void encode_box(const float* ex, const float* gt, float out[4]) {
  const float ew = ex[2] - ex[0] + 1.0f, eh = ex[3] - ex[1] + 1.0f;
  const float diag = sqrtf(ew * ew + eh * eh);
  const float ecx = ex[0] + 0.5f * ew, ecy = ex[1] + 0.5f * eh;
  const float gw = gt[2] - gt[0] + 1.0f, gh = gt[3] - gt[1] + 1.0f;
  const float gcx = gt[0] + 0.5f * gw, gcy = gt[1] + 0.5f * gh;
  out[0] = (gcx - ecx) / diag;
  out[1] = (gcy - ecy) / diag;
  out[2] = (float)log((double)(gw / ew));
  out[3] = (float)log((double)(gh / eh));
}


__global__ __launch_bounds__(256) void overlaps_kernel(const float* __restrict__ boxes, int box_ld, int n,
                                                      const float* __restrict__ query, int q_ld, int k,
                                                      float* __restrict__ out) {
  const size_t total = (size_t)n * k;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / k), q = (int)(i - (size_t)r * k);
    out[i] = iou_plus1(boxes + (size_t)r * box_ld, query + (size_t)q * q_ld);
  }
}

// ---------------------------------------------------------------------------------------------
// anchor_target_layer.  State in ws: gt_max[G] (uint bits), counters[4] = {fg, bg, examples, unused},
// max_ov[N], argmax[N] (int), keys_fg[N], keys_bg[N] (float scores for the top-k selection), keep[N] bytes.
// ---------------------------------------------------------------------------------------------
struct AtlFrame {
  float x_lo, x_hi, y_lo, y_hi;  // inside test: x1 >= x_lo, y1 >= y_lo, x2 < x_hi, y2 < y_hi  (:37-42)
};

constexpr int ATL_MAX_GT_LDS = 512;
constexpr unsigned ATL_LABEL_BLOCKS = 1024;   // atl_label_kernel: grid-stride over the anchors, one atomic pair per workgroup

__global__ __launch_bounds__(256) void atl_overlap_kernel(const float* __restrict__ anchors, int n,
                                                         const float* __restrict__ gt, int g,
                                                         const int* __restrict__ g_dev, AtlFrame fr,
                                                         float* __restrict__ max_ov, int* __restrict__ argmax,
                                                         unsigned* __restrict__ gt_max) {
  if (g_dev) g = max(1, min(g, *g_dev));   // live rows of a gt buffer padded to capacity g (a replayed hipGraph: g is baked in)
  // Per-gt maxima: a wave reduces with shuffles only when one of its anchors overlaps the gt box at all (most waves of a
  // 10^6-anchor pyramid level do not), wave leaders merge into an LDS table, and each workgroup issues ONE global atomicMax
  // per gt box at its end (a per-wave global atomic was 117 K same-address atomics for 937 500 anchors x 8 boxes: 423 us).
  // Every lane of a wave runs the same number of iterations (the ballot / shuffles need all lanes).
  __shared__ unsigned s_gmax[ATL_MAX_GT_LDS];
  const bool use_lds = g <= ATL_MAX_GT_LDS;
  for (int j = threadIdx.x; j < g && use_lds; j += blockDim.x) s_gmax[j] = 0u;
  __syncthreads();
  const int stride = gridDim.x * blockDim.x;
  const int iters = (n + stride - 1) / stride;
  for (int it = 0; it < iters; ++it) {
    const int i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
      const float4 a4 = reinterpret_cast<const float4*>(anchors)[i];
      a[0] = a4.x; a[1] = a4.y; a[2] = a4.z; a[3] = a4.w;
    }
    const bool inside = live && a[0] >= fr.x_lo && a[1] >= fr.y_lo && a[2] < fr.x_hi && a[3] < fr.y_hi;
    float best = -1.f;
    int arg = 0;
    for (int j = 0; j < g; ++j) {
      const float ov = inside ? iou_plus1(a, gt + (size_t)j * 5) : 0.f;
      if (inside && ov > best) { best = ov; arg = j; }      // first maximum, like argmax(dim=1)
      if (__ballot(ov > 0.f) == 0ull) continue;             // wave-uniform
      float wmax = ov;
      for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off));
      if ((threadIdx.x & 63) == 0) {                        // ov >= 0: the bit pattern orders like the value
        if (use_lds) atomicMax(&s_gmax[j], __float_as_uint(wmax));
        else atomicMax(gt_max + j, __float_as_uint(wmax));
      }
    }
    if (live) {
      max_ov[i] = inside ? best : -1.f;                     // -1 marks an anchor outside the frame
      argmax[i] = arg;
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < g && use_lds; j += blockDim.x)
    if (s_gmax[j] > 0u) atomicMax(gt_max + j, s_gmax[j]);
}

__global__ __launch_bounds__(256) void atl_label_kernel(const float* __restrict__ anchors, int n,
                                                       const float* __restrict__ gt, int g,
                                                       const int* __restrict__ g_dev,
                                                       const float* __restrict__ max_ov,
                                                       const unsigned* __restrict__ gt_max, float neg_ov, float pos_ov,
                                                       uint32_t seed, const uint32_t* __restrict__ seed_dev,
                                                       float* __restrict__ labels,
                                                       float* __restrict__ key_fg, float* __restrict__ key_bg,
                                                       int* __restrict__ counters) {
  if (seed_dev) seed += *seed_dev;      // per-step seed from device memory (a replayed hipGraph keeps `seed` itself)
  if (g_dev) g = max(1, min(g, *g_dev));
  const float eps = 1.1920929e-07f;  // torch.finfo(float32).eps (:62)
  int n_fg = 0, n_bg = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float mo = max_ov[i];
    float lab = -1.f;
    if (mo >= 0.f) {
      if (mo < neg_ov) lab = 0.f;                          // :66-69 (RPN_CLOBBER_POSITIVES off)
      const float4 a4 = reinterpret_cast<const float4*>(anchors)[i];
      const float a[4] = {a4.x, a4.y, a4.z, a4.w};
      for (int j = 0; j < g && mo > 0.f; ++j) {            // :63,73: every anchor tying a gt's best overlap
        const float gm = fmaxf(__uint_as_float(gt_max[j]), eps);   // (gm >= eps > 0: an anchor without any overlap ties nothing)
        if (iou_plus1(a, gt + (size_t)j * 5) == gm) { lab = 1.f; break; }
      }
      if (mo >= pos_ov) lab = 1.f;                         // :78
    }
    labels[i] = lab;
    // selection keys: candidates get a uniform key in (0,1], everything else -1 so it sorts last
    key_fg[i] = lab == 1.f ? (float)((rand_key(seed, 1, i) >> 8) + 1) * (1.0f / 16777216.0f) : -1.f;
    key_bg[i] = lab == 0.f ? (float)((rand_key(seed, 2, i) >> 8) + 1) * (1.0f / 16777216.0f) : -1.f;
    n_fg += lab == 1.f;
    n_bg += lab == 0.f;
  }
  // candidate counts: ONE global atomic per workgroup and label, and at most ATL_LABEL_BLOCKS workgroups.  Same-address
  // atomics retire ~10 ns apart: one per candidate (~250 K background anchors of a 1000x600 FPN frame) and then one per wave
  // (14 600 waves) both left the kernel at 167 us, all of it the atomic queue.  Integer sums, any order.
  __shared__ int s_cnt[2][4];
  for (int off = 32; off > 0; off >>= 1) {
    n_fg += __shfl_xor(n_fg, off);
    n_bg += __shfl_xor(n_bg, off);
  }
  if ((threadIdx.x & 63) == 0) {
    s_cnt[0][threadIdx.x >> 6] = n_fg;
    s_cnt[1][threadIdx.x >> 6] = n_bg;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int total = ((s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1]) + s_cnt[threadIdx.x][2]) + s_cnt[threadIdx.x][3];
    if (total) atomicAdd(counters + threadIdx.x, total);
  }
}

// keep[order[i]] = 1 for the first min(count, quota) entries that are real candidates (key > 0)
__global__ __launch_bounds__(256) void atl_mark_keep_kernel(const int64_t* __restrict__ order,
                                                           const float* __restrict__ sorted_keys, int top_n,
                                                           const int* __restrict__ counters, int which, int quota_total,
                                                           int num_fg_cap, uint8_t* __restrict__ keep) {
  // which = 0: fg, quota = num_fg_cap.  which = 1: bg, quota = quota_total - min(fg_count, num_fg_cap)  (:96-99)
  const int fg_kept = min(counters[0], num_fg_cap);
  const int quota = which == 0 ? num_fg_cap : quota_total - fg_kept;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < top_n; i += gridDim.x * blockDim.x)
    if (i < quota && sorted_keys[i] > 0.f) keep[order[i]] = 1;
}

__global__ __launch_bounds__(256) void atl_finalize_kernel(const float* __restrict__ anchors, int n,
                                                          const float* __restrict__ gt,
                                                          const int* __restrict__ argmax,
                                                          const float* __restrict__ max_ov,
                                                          const uint8_t* __restrict__ keep_fg,
                                                          const uint8_t* __restrict__ keep_bg,
                                                          const int* __restrict__ counters, int quota_total,
                                                          int num_fg_cap, float* __restrict__ labels,
                                                          float* __restrict__ targets, float* __restrict__ inside,
                                                          float* __restrict__ outside) {
  const int fg = counters[0], bg = counters[1];
  const int fg_kept = min(fg, num_fg_cap);
  const int bg_kept = min(bg, quota_total - fg_kept);
  const float wgt = 1.0f / (float)(fg_kept + bg_kept);      // :118-124 uniform example weights
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float lab = labels[i];
    if (lab == 1.f && fg > num_fg_cap && !keep_fg[i]) lab = -1.f;                   // :91-96
    if (lab == 0.f && bg > quota_total - fg_kept && !keep_bg[i]) lab = -1.f;        // :99-105
    labels[i] = lab;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (max_ov[i] >= 0.f) {                                  // every inside anchor gets targets (:110)
      const float4 a4 = reinterpret_cast<const float4*>(anchors)[i];
      const float a[4] = {a4.x, a4.y, a4.z, a4.w};
      float o[4];
      encode_box(a, gt + (size_t)argmax[i] * 5, o);
      t = make_float4(o[0], o[1], o[2], o[3]);
    }
    const float iw = lab == 1.f ? 1.f : 0.f;                 // RPN_BBOX_INSIDE_WEIGHTS (1,1,1,1)
    const float ow = lab >= 0.f ? wgt : 0.f;
    reinterpret_cast<float4*>(targets)[i] = t;
    reinterpret_cast<float4*>(inside)[i] = make_float4(iw, iw, iw, iw);
    reinterpret_cast<float4*>(outside)[i] = make_float4(ow, ow, ow, ow);
  }
}

// ---------------------------------------------------------------------------------------------
// proposal_target_layer for the image detector: one workgroup, R <= 4096 candidate RoIs.
// LDS: keys[2][npad] u64 (random key << 32 | roi index) for the fg and bg candidates.
// ---------------------------------------------------------------------------------------------
constexpr int PTL_THREADS = 1024;

struct PtlNorm {
  float means[7], stds[7];
};

// lidar_3d_bbox_transform (lib/model/bbox_transform.py:16-49): centre deltas over the RoI's BEV diagonal, z and the
// height from the RoI's 3-D anchor, yaw = the gt yaw itself.
__device__ __forceinline__ void encode_box_lidar(const float* roi, const float* anc, const float* gt7, float* o) {
  const float ln = roi[2] - roi[0] + 1.f, wd = roi[3] - roi[1] + 1.f, ht = anc[5];
  const float cx = roi[0] + ln / 2.0f, cy = roi[1] + wd / 2.0f, cz = anc[2];
  const float diag = sqrtf(ln * ln + wd * wd);
  o[0] = (gt7[0] - cx) / diag;
  o[1] = (gt7[1] - cy) / diag;
  o[2] = (gt7[2] - cz) / ht;
  o[3] = logf(gt7[3] / ln);
  o[4] = logf(gt7[4] / wd);
  o[5] = logf(gt7[5] / ht);
  o[6] = gt7[6];
}

// E = 4: image detector (gt rows [x1,y1,x2,y2,cls]).  E = 7: LiDAR detector - overlaps on the BEV rectangles `gt`,
// regression targets against `true_gt` rows [xc,yc,zc,l,w,h,ry,cls] and the RoI's 3-D anchor, which is carried along.
template <int E>
__global__ __launch_bounds__(PTL_THREADS) void ptl_kernel(const float* __restrict__ rois, const float* __restrict__ scores,
                                                         const int* __restrict__ roi_count, int num_rois,
                                                         const unsigned char* __restrict__ skip,
                                                         const float* __restrict__ anchors3d,
                                                         const float* __restrict__ true_gt,
                                                         float* __restrict__ out_anchors3d,
                                                         const float* __restrict__ gt, int g,
                                                         const int* __restrict__ g_dev, int num_classes,
                                                         int rois_per_frame, int fg_quota, float fg_thresh, float bg_hi,
                                                         float bg_lo, PtlNorm norm, uint32_t seed,
                                                         const uint32_t* __restrict__ seed_dev, int npad,
                                                         float* __restrict__ out_labels, float* __restrict__ out_rois,
                                                         float* __restrict__ out_scores, float* __restrict__ out_targets,
                                                         float* __restrict__ out_inside, float* __restrict__ out_outside,
                                                         int* __restrict__ out_assign, int* __restrict__ out_counts) {
  if (seed_dev) seed += *seed_dev;
  if (g_dev) g = max(1, min(g, *g_dev));
  extern __shared__ __attribute__((aligned(16))) unsigned char ptl_smem[];
  uint64_t* kfg = reinterpret_cast<uint64_t*>(ptl_smem);
  uint64_t* kbg = kfg + npad;
  int* assign = reinterpret_cast<int*>(kbg + npad);  // [num_rois]
  __shared__ int s_fg, s_bg;
  const int t = threadIdx.x;
  const int R = roi_count ? min(*roi_count, num_rois) : num_rois;
  if (t == 0) { s_fg = 0; s_bg = 0; }
  for (int i = t; i < npad; i += PTL_THREADS) { kfg[i] = ~0ull; kbg[i] = ~0ull; }
  __syncthreads();
  for (int i = t; i < R; i += PTL_THREADS) {
    if (skip && skip[i]) continue;       // TRAIN.IGNORE_DC: proposals inside a don't-care region are no candidates (:182-187)
    const float* b = rois + (size_t)i * 5 + 1;
    float best = -1.f;
    int arg = 0;
    for (int j = 0; j < g; ++j) {
      const float ov = iou_plus1(b, gt + (size_t)j * 5);
      if (ov > best) { best = ov; arg = j; }
    }
    assign[i] = arg;
    if (best >= fg_thresh) {                                   // :200
      kfg[i] = ((uint64_t)rand_key(seed, 3, i) << 32) | (uint32_t)i;
      atomicAdd(&s_fg, 1);
    } else if (best < bg_hi && best >= bg_lo) {               // :203-204
      kbg[i] = ((uint64_t)rand_key(seed, 4, i) << 32) | (uint32_t)i;
      atomicAdd(&s_bg, 1);
    }
  }
  __syncthreads();
  block_bitonic_sort(kfg, npad);   // candidates first, in random order
  block_bitonic_sort(kbg, npad);
  const int nfg_c = s_fg, nbg_c = s_bg;
  // :206-231 quotas
  int n_fg, n_bg;
  if (nfg_c > 0 && nbg_c > 0) { n_fg = min(fg_quota, nfg_c); n_bg = rois_per_frame - n_fg; }
  else if (nfg_c > 0) { n_fg = rois_per_frame; n_bg = 0; }
  else { n_fg = 0; n_bg = nbg_c > 0 ? rois_per_frame : 0; }
  if (t == 0) { out_counts[0] = n_fg; out_counts[1] = n_bg; out_counts[2] = nfg_c; out_counts[3] = nbg_c; }
  const int cols = E * num_classes;
  for (int j = t; j < rois_per_frame; j += PTL_THREADS) {
    int src = -1;
    bool is_fg = false;
    if (j < n_fg) {
      // mixed case: n_fg <= candidates, a random subset without replacement (:207-211); foreground-only case with
      // fewer candidates than rows: every row drawn WITH replacement (:217-221, torch_choice_replace :273-275)
      const int pick = n_fg <= nfg_c ? j : (int)(rand_key(seed, 5, j) % (uint32_t)nfg_c);
      src = (int)(kfg[pick] & 0xFFFFFFFFu);
      is_fg = true;
    } else if (j - n_fg < n_bg) {
      const int q = j - n_fg;
      const int pick = (nbg_c >= n_bg) ? q : (int)(rand_key(seed, 6, q) % (uint32_t)nbg_c);  // to_replace (:212)
      src = (int)(kbg[pick] & 0xFFFFFFFFu);
    }
    float lab = 0.f;
    float* tr = out_targets + (size_t)j * cols;
    float* ir = out_inside + (size_t)j * cols;
    float* orow = out_outside + (size_t)j * cols;
    for (int q = 0; q < cols; ++q) { tr[q] = 0.f; ir[q] = 0.f; orow[q] = 0.f; }
    float r5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float sc = 0.f;
    int as = 0;
    if (src >= 0) {
      for (int q = 0; q < 5; ++q) r5[q] = rois[(size_t)src * 5 + q];
      sc = scores ? scores[src] : 0.f;
      as = assign[src];
      const float* gb = gt + (size_t)as * 5;
      if (is_fg) lab = gb[4];                                  // :198,238: class of the assigned gt, bg rows -> 0
      const int c = (int)lab;
      if (c > 0 && c < num_classes) {
        float o[E];
        if (E == 7) encode_box_lidar(r5 + 1, anchors3d + (size_t)src * 7, true_gt + (size_t)as * 8, o);
        else encode_box(r5 + 1, gb, o);
        for (int q = 0; q < E; ++q) {                          // :142-163 normalised targets, :64-103 class slot
          tr[E * c + q] = (o[q] - norm.means[q]) / norm.stds[q];
          ir[E * c + q] = 1.f;
          orow[E * c + q] = 1.f;
        }
      }
    }
    if (E == 7)
      for (int q = 0; q < 7; ++q) out_anchors3d[(size_t)j * 7 + q] = src >= 0 ? anchors3d[(size_t)src * 7 + q] : 0.f;
    out_labels[j] = lab;
    for (int q = 0; q < 5; ++q) out_rois[(size_t)j * 5 + q] = r5[q];
    out_scores[j] = sc;
    out_assign[j] = as;
  }
}

int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
unsigned grid_for(size_t items, unsigned cap = 4096) { return (unsigned)std::min<size_t>((items + 255) / 256, cap); }

struct AtlLayout {
  size_t gt_max, counters, max_ov, argmax, key_fg, key_bg, keep_fg, keep_bg, order, sorted, sort_count, sort_ws, sort_ws_bytes, total;
};
AtlLayout atl_layout(int n, int g, int top_n) {
  AtlLayout l;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = align_up(o + bytes, 256); return at; };
  l.gt_max = take((size_t)std::max(g, 1) * 4);
  l.counters = take(16);
  l.max_ov = take((size_t)n * 4);
  l.argmax = take((size_t)n * 4);
  l.key_fg = take((size_t)n * 4);
  l.key_bg = take((size_t)n * 4);
  l.keep_fg = take((size_t)n);
  l.keep_bg = take((size_t)n);
  l.order = take((size_t)top_n * 8);
  l.sorted = take((size_t)top_n * 4);
  l.sort_count = take(16);
  l.sort_ws_bytes = frcnn_sort_topk_desc_ws_bytes(n, top_n);
  l.sort_ws = take(l.sort_ws_bytes);
  l.total = o;
  return l;
}

}  // namespace

extern "C" int frcnn_bbox_overlaps(const float* boxes, int box_ld, int n, const float* query, int query_ld, int k,
                                   float* overlaps, void* stream_) {
  FRCNN_REQUIRE(boxes && query && overlaps && n > 0 && k > 0 && box_ld >= 4 && query_ld >= 4, "bbox_overlaps: bad arguments");
  hipLaunchKernelGGL(overlaps_kernel, dim3(grid_for((size_t)n * k)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     boxes, box_ld, n, query, query_ld, k, overlaps);
  return check_launch("overlaps_kernel");
}

namespace {
// the two encoders as stand-alone calls (the target layers above apply them to their sampled rows in place)
__global__ __launch_bounds__(256) void encode_boxes_kernel(const float* __restrict__ ex, int ex_ld,
                                                          const float* __restrict__ gt, int gt_ld, int n,
                                                          float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float o[4];
  encode_box(ex + (size_t)i * ex_ld, gt + (size_t)i * gt_ld, o);
  *reinterpret_cast<float4*>(out + (size_t)i * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(256) void encode_boxes_lidar_kernel(const float* __restrict__ rois, int roi_ld,
                                                                const float* __restrict__ anchors3d,
                                                                const float* __restrict__ gt, int gt_ld, int n,
                                                                float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float o[7];
  encode_box_lidar(rois + (size_t)i * roi_ld, anchors3d + (size_t)i * 7, gt + (size_t)i * gt_ld, o);
  for (int q = 0; q < 7; ++q) out[(size_t)i * 7 + q] = o[q];
}
}  // namespace

extern "C" int frcnn_bbox_transform(const float* ex_rois, int ex_ld, const float* gt_rois, int gt_ld, int n, float* targets,
                                    void* stream_) {
  FRCNN_REQUIRE(ex_rois && gt_rois && targets && n > 0 && ex_ld >= 4 && gt_ld >= 4, "bbox_transform: bad arguments");
  hipLaunchKernelGGL(encode_boxes_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_), ex_rois,
                     ex_ld, gt_rois, gt_ld, n, targets);
  return check_launch("encode_boxes_kernel");
}

extern "C" int frcnn_lidar_bbox_transform(const float* ex_rois, int roi_ld, const float* ex_anchors_3d, const float* gt_rois,
                                          int gt_ld, int n, float* targets, void* stream_) {
  FRCNN_REQUIRE(ex_rois && ex_anchors_3d && gt_rois && targets && n > 0 && roi_ld >= 4 && gt_ld >= 7,
                "lidar_bbox_transform: bad arguments");
  hipLaunchKernelGGL(encode_boxes_lidar_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     ex_rois, roi_ld, ex_anchors_3d, gt_rois, gt_ld, n, targets);
  return check_launch("encode_boxes_lidar_kernel");
}

extern "C" size_t frcnn_anchor_target_layer_ws_bytes(int num_anchors_total, int num_gt, int rpn_batchsize) {
  if (num_anchors_total <= 0 || rpn_batchsize <= 0) return 0;
  const int top_n = std::min(rpn_batchsize, 16384);
  return atl_layout(num_anchors_total, num_gt, top_n).total;
}

extern "C" int frcnn_anchor_target_layer(const float* anchors, int n, const float* gt_boxes, int num_gt,
                                         const int* num_gt_dev, const float* info_host, int rpn_batchsize, float fg_fraction,
                                         float negative_overlap, float positive_overlap, uint32_t seed,
                                         const uint32_t* seed_dev, float* labels, float* targets, float* inside,
                                         float* outside, int* counts, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FRCNN_REQUIRE(anchors && gt_boxes && info_host && labels && targets && inside && outside && n > 0 && num_gt > 0 &&
                    rpn_batchsize > 0,
                "anchor_target_layer: bad arguments (needs at least one gt box)");
  const int top_n = std::min(rpn_batchsize, 16384);
  const AtlLayout l = atl_layout(n, num_gt, top_n);
  if (!ws || ws_bytes < l.total) return fail(FRCNN_ERR_WS, "anchor_target_layer: workspace %zu < %zu bytes", ws_bytes, l.total);
  char* base = static_cast<char*>(ws);
  unsigned* gt_max = reinterpret_cast<unsigned*>(base + l.gt_max);
  int* counters = reinterpret_cast<int*>(base + l.counters);
  float* max_ov = reinterpret_cast<float*>(base + l.max_ov);
  int* argmax = reinterpret_cast<int*>(base + l.argmax);
  float* key_fg = reinterpret_cast<float*>(base + l.key_fg);
  float* key_bg = reinterpret_cast<float*>(base + l.key_bg);
  uint8_t* keep_fg = reinterpret_cast<uint8_t*>(base + l.keep_fg);
  uint8_t* keep_bg = reinterpret_cast<uint8_t*>(base + l.keep_bg);
  int64_t* order = reinterpret_cast<int64_t*>(base + l.order);
  float* sorted = reinterpret_cast<float*>(base + l.sorted);
  int* sort_count = reinterpret_cast<int*>(base + l.sort_count);
  // gt_max, counters .. up to max_ov are contiguous at the start; keep flags are contiguous too
  hipError_t e = fill_bytes(base, 0, l.max_ov, stream);
  if (e == hipSuccess) e = fill_bytes(keep_fg, 0, l.order - l.keep_fg, stream);
  if (e != hipSuccess) return fail(FRCNN_ERR_LAUNCH, "anchor_target_layer: memset: %s", hipGetErrorString(e));
  const AtlFrame fr{info_host[0], info_host[1], info_host[2], info_host[3]};
  const unsigned grid = grid_for((size_t)n);
  hipLaunchKernelGGL(atl_overlap_kernel, dim3(grid), dim3(256), 0, stream, anchors, n, gt_boxes, num_gt, num_gt_dev, fr, max_ov,
                     argmax, gt_max);
  int rc = check_launch("atl_overlap_kernel");
  if (rc != FRCNN_OK) return rc;
  hipLaunchKernelGGL(atl_label_kernel, dim3(std::min(grid, ATL_LABEL_BLOCKS)), dim3(256), 0, stream, anchors, n, gt_boxes, num_gt, num_gt_dev, max_ov,
                     gt_max, negative_overlap, positive_overlap, seed, seed_dev, labels, key_fg, key_bg, counters);
  rc = check_launch("atl_label_kernel");
  if (rc != FRCNN_OK) return rc;
  const int num_fg_cap = (int)(fg_fraction * (float)rpn_batchsize);   // :91
  if (rpn_batchsize <= 16384) {
    // random sub-sampling: the `quota` candidates with the largest random keys survive
    for (int which = 0; which < 2; ++which) {
      rc = frcnn_sort_topk_desc(which == 0 ? key_fg : key_bg, n, top_n, order, sorted, sort_count,
                                l.sort_ws_bytes ? base + l.sort_ws : nullptr, l.sort_ws_bytes, stream_);
      if (rc != FRCNN_OK) return rc;
      hipLaunchKernelGGL(atl_mark_keep_kernel, dim3(grid_for((size_t)top_n)), dim3(256), 0, stream, order, sorted, top_n,
                         counters, which, rpn_batchsize, num_fg_cap, which == 0 ? keep_fg : keep_bg);
      rc = check_launch("atl_mark_keep_kernel");
      if (rc != FRCNN_OK) return rc;
    }
  }
  // (rpn_batchsize > 16384 only makes sense as "no sub-sampling": the caps then exceed any candidate count)
  hipLaunchKernelGGL(atl_finalize_kernel, dim3(grid), dim3(256), 0, stream, anchors, n, gt_boxes, argmax, max_ov, keep_fg,
                     keep_bg, counters, rpn_batchsize, num_fg_cap, labels, targets, inside, outside);
  rc = check_launch("atl_finalize_kernel");
  if (rc != FRCNN_OK) return rc;
  if (counts) {
    e = copy_bytes(counts, counters, 2 * sizeof(int), stream);
    if (e != hipSuccess) return fail(FRCNN_ERR_LAUNCH, "anchor_target_layer: copy counts: %s", hipGetErrorString(e));
  }
  return FRCNN_OK;
}

namespace {
template <int E>
int launch_ptl(const float* rois, const float* roi_scores, const int* roi_count, int num_rois,
               const unsigned char* skip, const float* anchors3d,
               const float* true_gt, float* out_anchors3d, const float* gt_boxes, int num_gt, const int* num_gt_dev,
               int num_classes, int rois_per_frame, float fg_fraction, float fg_thresh, float bg_thresh_hi, float bg_thresh_lo,
               const float* means_host, const float* stds_host, uint32_t seed, const uint32_t* seed_dev, float* labels,
               float* out_rois,
               float* out_scores, float* targets, float* inside, float* outside, int* gt_assignment, int* counts,
               void* stream_) {
  PtlNorm norm;
  for (int q = 0; q < 7; ++q) { norm.means[q] = q < E ? means_host[q] : 0.f; norm.stds[q] = q < E ? stds_host[q] : 1.f; }
  const int npad = next_pow2(std::max(num_rois, 2));
  const size_t lds = (size_t)npad * 16 + (size_t)num_rois * 4;
  static std::atomic<size_t> configured{0};
  if (lds > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ptl_kernel<E>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(FRCNN_ERR_LAUNCH, "proposal_target_layer: set LDS size: %s", hipGetErrorString(e));
    configured = lds;
  }
  const int fg_quota = (int)lrintf(fg_fraction * (float)rois_per_frame);   // int(round(...)) (:44-45)
  hipLaunchKernelGGL(ptl_kernel<E>, dim3(1), dim3(PTL_THREADS), lds, static_cast<hipStream_t>(stream_), rois, roi_scores,
                     roi_count, num_rois, skip, anchors3d, true_gt, out_anchors3d, gt_boxes, num_gt, num_gt_dev, num_classes,
                     rois_per_frame, fg_quota, fg_thresh, bg_thresh_hi, bg_thresh_lo, norm, seed, seed_dev, npad, labels, out_rois,
                     out_scores, targets, inside, outside, gt_assignment, counts);
  return check_launch("ptl_kernel");
}
}  // namespace

extern "C" int frcnn_proposal_target_layer(const float* rois, const float* roi_scores, const int* roi_count,
                                           int num_rois, const float* gt_boxes, int num_gt, const int* num_gt_dev,
                                           int num_classes, int rois_per_frame, float fg_fraction, float fg_thresh,
                                           float bg_thresh_hi,
                                           float bg_thresh_lo, const float* means_host, const float* stds_host,
                                           uint32_t seed, const uint32_t* seed_dev, float* labels, float* out_rois,
                                           float* out_scores,
                                           float* targets, float* inside, float* outside, int* gt_assignment,
                                           int* counts, const unsigned char* skip_mask, void* stream_) {
  FRCNN_REQUIRE(rois && gt_boxes && means_host && stds_host && labels && out_rois && out_scores && targets && inside &&
                    outside && gt_assignment && counts && num_rois > 0 && num_rois <= 4096 && num_gt > 0 &&
                    num_classes > 1 && rois_per_frame > 0,
                "proposal_target_layer: bad arguments (num_rois <= 4096, at least one gt box)");
  return launch_ptl<4>(rois, roi_scores, roi_count, num_rois, skip_mask, nullptr, nullptr, nullptr, gt_boxes, num_gt, num_gt_dev, num_classes,
                       rois_per_frame, fg_fraction, fg_thresh, bg_thresh_hi, bg_thresh_lo, means_host, stds_host, seed, seed_dev,
                       labels, out_rois, out_scores, targets, inside, outside, gt_assignment, counts, stream_);
}

extern "C" int frcnn_proposal_target_layer_lidar(const float* rois, const float* roi_scores, const int* roi_count,
                                                 int num_rois, const float* anchors3d, const float* gt_boxes,
                                                 const float* true_gt_boxes, int num_gt, const int* num_gt_dev,
                                                 int num_classes, int rois_per_frame, float fg_fraction, float fg_thresh,
                                                 float bg_thresh_hi, float bg_thresh_lo, const float* means_host,
                                                 const float* stds_host, uint32_t seed, const uint32_t* seed_dev,
                                                 float* labels, float* out_rois,
                                                 float* out_scores, float* out_anchors3d, float* targets, float* inside,
                                                 float* outside, int* gt_assignment, int* counts,
                                                 const unsigned char* skip_mask, void* stream_) {
  FRCNN_REQUIRE(rois && anchors3d && gt_boxes && true_gt_boxes && means_host && stds_host && labels && out_rois &&
                    out_scores && out_anchors3d && targets && inside && outside && gt_assignment && counts &&
                    num_rois > 0 && num_rois <= 4096 && num_gt > 0 && num_classes > 1 && rois_per_frame > 0,
                "proposal_target_layer_lidar: bad arguments (num_rois <= 4096, at least one gt box)");
  return launch_ptl<7>(rois, roi_scores, roi_count, num_rois, skip_mask, anchors3d, true_gt_boxes, out_anchors3d, gt_boxes, num_gt,
                       num_gt_dev, num_classes, rois_per_frame, fg_fraction, fg_thresh, bg_thresh_hi, bg_thresh_lo, means_host,
                       stds_host, seed, seed_dev, labels, out_rois, out_scores, targets, inside, outside, gt_assignment, counts,
                       stream_);
}

// ------------------------------------------------------------------------------------------------
// The RPN losses read the anchors the anchor target layer labelled - at most cfg.TRAIN.RPN_BATCHSIZE of them
// (lib/layer_utils/anchor_target_layer.py:91-107) - so the gradient of the RPN head is non-zero on at most that many
// PIXELS of the feature map.  These three kernels let the training step run the RPN's differentiable pass on those pixels
// only (nets/network.py: _rpn_losses_on_labelled_pixels): the list of pixels that carry a labelled anchor, the R x S input
// patches around them, and the scatter of the patch gradients back into the map.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int AP_THREADS = 1024;

// flag[px] = 1 when any of the pixel's A labels is != -1 (one thread per pixel, chip-wide)
__global__ __launch_bounds__(256) void labelled_flags_kernel(const float* __restrict__ labels, int hw, int A,
                                                            int* __restrict__ flag) {
  const int px = blockIdx.x * blockDim.x + threadIdx.x;
  if (px >= hw) return;
  int on = 0;
  for (int a = 0; a < A; ++a) on |= labels[(size_t)px * A + a] != -1.0f ? 1 : 0;   // no short circuit: independent loads
  flag[px] = on;
}

// compaction of the flags (ascending); idx[i] = -1 for i >= count[0].  count[0] = min(total, cap), count[1] = total
__global__ __launch_bounds__(AP_THREADS) void labelled_pixels_kernel(const int* __restrict__ flag, int hw, int cap,
                                                                     int64_t* __restrict__ idx, int* __restrict__ count) {
  __shared__ int s_wave[AP_THREADS / 64];
  __shared__ int s_base;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) s_base = 0;
  for (int i = t; i < cap; i += AP_THREADS) idx[i] = -1;
  __syncthreads();
  for (int p0 = 0; p0 < hw; p0 += AP_THREADS) {
    const int px = p0 + t;
    const bool on = px < hw && flag[px] != 0;
    const unsigned long long m = __ballot(on);
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int before = s_base;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    const int pos = before + __popcll(m & ((1ull << lane) - 1ull));
    if (on && pos < cap) idx[pos] = px;
    __syncthreads();
    if (t == 0) {
      int tot = s_base;
      for (int w = 0; w < AP_THREADS / 64; ++w) tot += s_wave[w];
      s_base = tot;
    }
    __syncthreads();
  }
  if (t == 0) { count[0] = min(s_base, cap); count[1] = s_base; }
}

// out[i][r][s][:] = x[py + r - pad][px + s - pad][:] (zero outside the map and for i >= count); C4 = C / 4
__global__ __launch_bounds__(256) void gather_patches_kernel(const float* __restrict__ x, int H, int W, int C4,
                                                            const int64_t* __restrict__ idx, const int* __restrict__ count,
                                                            int cap, int R, int S, int pad, float* __restrict__ out) {
  const size_t total = (size_t)cap * R * S * C4;
  const int live = min(*count, cap);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t q = i / C4;
    const int s = (int)(q % S); q /= S;
    const int r = (int)(q % R);
    const int row = (int)(q / R);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < live) {
      const int p = (int)idx[row];
      const int y = p / W + r - pad, xx = p % W + s - pad;
      if (p >= 0 && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W)
        v = reinterpret_cast<const float4*>(x)[((size_t)y * W + xx) * C4 + c4];
    }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

// dx[py + r - pad][px + s - pad][:] += d[i][r][s][:]; patches of neighbouring pixels overlap -> float atomics
__global__ __launch_bounds__(256) void scatter_add_patches_kernel(const float* __restrict__ d, int H, int W, int C,
                                                                 const int64_t* __restrict__ idx,
                                                                 const int* __restrict__ count, int cap, int R, int S, int pad,
                                                                 float* __restrict__ dx) {
  const int live = min(*count, cap);
  const size_t total = (size_t)live * R * S * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t q = i / C;
    const int s = (int)(q % S); q /= S;
    const int r = (int)(q % R);
    const int row = (int)(q / R);
    const int p = (int)idx[row];
    const int y = p / W + r - pad, xx = p % W + s - pad;
    if (p >= 0 && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W)
      atomicAdd(dx + ((size_t)y * W + xx) * C + c, d[i]);
  }
}
}  // namespace

extern "C" size_t frcnn_labelled_pixels_ws_bytes(int hw) { return hw > 0 ? (size_t)hw * sizeof(int) : 0; }

extern "C" int frcnn_labelled_pixels(const float* labels, int hw, int num_anchors, int cap, int64_t* idx, int* count, void* ws,
                                     size_t ws_bytes, void* stream_) {
  FRCNN_REQUIRE(labels && idx && count && hw > 0 && num_anchors > 0 && cap > 0, "labelled_pixels: bad arguments");
  if (!ws || ws_bytes < frcnn_labelled_pixels_ws_bytes(hw))
    return fail(FRCNN_ERR_WS, "labelled_pixels: workspace %zu < %zu bytes", ws_bytes, frcnn_labelled_pixels_ws_bytes(hw));
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int* flag = static_cast<int*>(ws);
  hipLaunchKernelGGL(labelled_flags_kernel, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, stream, labels, hw, num_anchors,
                     flag);
  int rc = check_launch("labelled_flags_kernel");
  if (rc != FRCNN_OK) return rc;
  hipLaunchKernelGGL(labelled_pixels_kernel, dim3(1), dim3(AP_THREADS), 0, stream, flag, hw, cap, idx, count);
  return check_launch("labelled_pixels_kernel");
}

extern "C" int frcnn_gather_patches(const float* x, int h, int w, int c, const int64_t* idx, const int* count, int cap, int r,
                                    int s, int pad, float* out, void* stream_) {
  FRCNN_REQUIRE(x && idx && count && out && h > 0 && w > 0 && c > 0 && c % 4 == 0 && cap > 0 && r > 0 && s > 0 && pad >= 0,
                "gather_patches: bad arguments (c%%4==0)");
  const size_t total = (size_t)cap * r * s * (c / 4);
  hipLaunchKernelGGL(gather_patches_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), x, h, w, c / 4, idx, count, cap, r, s, pad, out);
  return check_launch("gather_patches_kernel");
}

extern "C" int frcnn_scatter_add_patches(const float* d, int h, int w, int c, const int64_t* idx, const int* count, int cap,
                                         int r, int s, int pad, float* dx, void* stream_) {
  FRCNN_REQUIRE(d && idx && count && dx && h > 0 && w > 0 && c > 0 && cap > 0 && r > 0 && s > 0 && pad >= 0,
                "scatter_add_patches: bad arguments");
  const size_t total = (size_t)cap * r * s * c;
  hipLaunchKernelGGL(scatter_add_patches_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), d, h, w, c, idx, count, cap, r, s, pad, dx);
  return check_launch("scatter_add_patches_kernel");
}
