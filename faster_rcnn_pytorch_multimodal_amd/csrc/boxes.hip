// RPN proposal stage and per-class detection filter: anchors, decode + clip, top-k sort, NMS.
// Built with -ffp-contract=off (see box_math.h).  These are HBM/latency-bound integer/bit kernels:
// wave64 ballots/readlanes and LDS sorting, no matrix cores.
#include "common.h"

#include <atomic>
#include "box_math.h"

using namespace frcnn;

namespace {

// ------------------------------------------------------------------------------------------------
// generate_anchors_pre — lib/layer_utils/snippets.py:27-37.  numpy adds the float64 base table to the
// integer shift grid and casts ONCE to float32; do the same (double add, one rounding).
// Layout (H, W, A) flattened, A fastest (snippets.py:35-37).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void anchors_kernel(const double* __restrict__ base, int A, int H, int W,
                                                     int stride, float* __restrict__ out) {
  const int total = H * W * A;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int a = i % A;
    const int pix = i / A;
    const int x = pix % W, y = pix / W;
    const double sx = (double)(x * stride), sy = (double)(y * stride);
    float4 v;
    v.x = (float)(base[a * 4 + 0] + sx);
    v.y = (float)(base[a * 4 + 1] + sy);
    v.z = (float)(base[a * 4 + 2] + sx);
    v.w = (float)(base[a * 4 + 3] + sy);
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// GridAnchor3dGenerator — lib/layer_utils/generate_3d_anchors.py:15-118, plus the axis-aligned BEV box of
// every 3-D anchor (lib/utils/bbox.py:256-293, clip=False) that the RPN regresses against.
// `base` holds one row per anchor type t = size*R + rot (the order of the meshgrid, rot fastest):
//   [lo_x, lo_y, hi_x, hi_y, z, l, w, h, ry]   lo/hi = float32 BEV half extents of the rotated box.
// Layout (H, W, T): anchors3d[i] = [x, y, z, l, w, h, ry], anchors2d[i] = [lo_x+x, lo_y+y, hi_x+x, hi_y+y]
// with x = float(col*stride), y = float(row*stride) and fp32 adds like the reference's float32 arrays.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void anchors3d_kernel(const float* __restrict__ base, int T, int H, int W,
                                                       int stride, float* __restrict__ a3, float* __restrict__ a2) {
  const int total = H * W * T;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int t = i % T;
    const int pix = i / T;
    const float xc = (float)((pix % W) * stride), yc = (float)((pix / W) * stride);
    const float* b = base + t * 9;
    float* o = a3 + (size_t)i * 7;
    o[0] = xc; o[1] = yc; o[2] = b[4]; o[3] = b[5]; o[4] = b[6]; o[5] = b[7]; o[6] = b[8];
    reinterpret_cast<float4*>(a2)[i] = make_float4(b[0] + xc, b[1] + yc, b[2] + xc, b[3] + yc);
  }
}

// ------------------------------------------------------------------------------------------------
// proposal_layer.py:32-36: fg score, bbox_transform_inv, clip_boxes for anchor i = pix*A + a.
// ------------------------------------------------------------------------------------------------
struct ClipInfo {
  float x_lo, x_hi, y_lo, y_hi;
};

__global__ __launch_bounds__(256) void rpn_decode_clip_kernel(const float* __restrict__ rpn, int ld,
                                                             const float* __restrict__ probs_in,
                                                             const float* __restrict__ deltas_in,
                                                             const float* __restrict__ anchors, ClipInfo clip,
                                                             int total, int A, float* __restrict__ scores,
                                                             float* __restrict__ proposals) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int a = i % A;
    const int pix = i / A;
    float score;
    if (probs_in) {
      score = probs_in[i];
    } else {
      // 2-way softmax over (bg, fg) exactly as torch.softmax: subtract max, exp, divide by the sum
      const float bg = rpn[(size_t)pix * ld + a], fg = rpn[(size_t)pix * ld + A + a];
      const float m = fmaxf(bg, fg);
      const float eb = exp_f32(bg - m), ef = exp_f32(fg - m);
      score = ef / (eb + ef);
    }
    float4 d;
    if (deltas_in) d = reinterpret_cast<const float4*>(deltas_in)[i];
    else {
      const float* p = rpn + (size_t)pix * ld + 2 * A + a * 4;
      d = make_float4(p[0], p[1], p[2], p[3]);
    }
    const float4 an = reinterpret_cast<const float4*>(anchors)[i];
    float o[4];
    decode_box(an.x, an.y, an.z, an.w, d.x, d.y, d.z, d.w, o);
    float4 r;
    r.x = clampf(o[0], clip.x_lo, clip.x_hi);
    r.y = clampf(o[1], clip.y_lo, clip.y_hi);
    r.z = clampf(o[2], clip.x_lo, clip.x_hi);
    r.w = clampf(o[3], clip.y_lo, clip.y_hi);
    scores[i] = score;
    reinterpret_cast<float4*>(proposals)[i] = r;
  }
}

// ------------------------------------------------------------------------------------------------
// Top-k of n scores in the total order (score desc, index asc), one 1024-thread workgroup:
//   1. 4-pass 8-bit radix select of the top_n-th key (LDS histogram),
//   2. compaction of keys < kth (any order) and of the lowest-index ties == kth (ordered scan),
//   3. bitonic sort of the <= 16384 u64 (key<<32 | index) candidates in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int SORT_THREADS = 1024;

__global__ __launch_bounds__(SORT_THREADS) void sort_topk_desc_kernel(const float* __restrict__ scores, int n,
                                                                      int top_n, int npad,
                                                                      int64_t* __restrict__ order_out,
                                                                      float* __restrict__ scores_out,
                                                                      int* __restrict__ count_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sort_smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(sort_smem);             // [npad]
  uint32_t* hist = reinterpret_cast<uint32_t*>(keys + npad);           // [256]
  uint32_t* scan = hist + 256;                                         // [SORT_THREADS]
  __shared__ uint32_t s_prefix, s_need, s_fill;
  const int t = threadIdx.x;
  const int take = min(n, top_n);

  for (int i = t; i < npad; i += SORT_THREADS) keys[i] = ~0ull;
  if (t == 0) { s_prefix = 0; s_need = (uint32_t)take; s_fill = 0; }
  __syncthreads();

  uint32_t kth = 0xFFFFFFFFu;  // every key <= kth is a candidate when n <= top_n
  uint32_t need_eq = 0;        // how many keys == kth to take (lowest indices first)
  if (n > top_n) {
    // radix select: after the loop s_prefix = kth key, s_need = rank of kth among keys with that value
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      for (int i = t; i < 256; i += SORT_THREADS) hist[i] = 0;
      __syncthreads();
      const uint32_t prefix = s_prefix;
      const uint32_t himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
      for (int i = t; i < n; i += SORT_THREADS) {
        const uint32_t k = desc_key(scores[i]);
        if ((k & himask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (t == 0) {
        uint32_t need = s_need, acc = 0;
        int b = 0;
        for (; b < 256; ++b) {
          if (acc + hist[b] >= need) break;
          acc += hist[b];
        }
        s_prefix = prefix | ((uint32_t)b << shift);
        s_need = need - acc;
      }
      __syncthreads();
    }
    kth = s_prefix;
    need_eq = s_need;
  }
  __syncthreads();

  // compaction.  Thread t owns the contiguous index range [lo, hi) so ties are taken in index order.
  const int per = (n + SORT_THREADS - 1) / SORT_THREADS;
  const int lo = min(t * per, n), hi = min(lo + per, n);
  uint32_t my_eq = 0;
  for (int i = lo; i < hi; ++i) {
    const uint32_t k = desc_key(scores[i]);
    if (k < kth || (n <= top_n)) {
      const uint32_t pos = atomicAdd(&s_fill, 1u);
      keys[pos] = ((uint64_t)k << 32) | (uint32_t)i;
    } else if (k == kth) {
      ++my_eq;
    }
  }
  scan[t] = my_eq;
  __syncthreads();
  if (n > top_n) {
    // exclusive scan of tie counts (Hillis-Steele in LDS)
    for (int off = 1; off < SORT_THREADS; off <<= 1) {
      const uint32_t v = t >= off ? scan[t - off] : 0u;
      __syncthreads();
      scan[t] += v;
      __syncthreads();
    }
    uint32_t rank = scan[t] - my_eq;  // ties before this thread's range
    const uint32_t base = (uint32_t)take - need_eq;  // number of keys < kth
    for (int i = lo; i < hi && rank < need_eq; ++i) {
      const uint32_t k = desc_key(scores[i]);
      if (k == kth) {
        keys[base + rank] = ((uint64_t)k << 32) | (uint32_t)i;
        ++rank;
      }
    }
  }
  __syncthreads();

  block_bitonic_sort(keys, npad);

  for (int i = t; i < take; i += SORT_THREADS) {
    const uint32_t idx = (uint32_t)(keys[i] & 0xFFFFFFFFu);
    order_out[i] = (int64_t)idx;
    scores_out[i] = scores[idx];
  }
  if (t == 0) count_out[0] = take;
}

// ------------------------------------------------------------------------------------------------
// Multi-workgroup form of the same top-k for large n (10^5 .. 10^6 scores: FPN / training RPN).  Same total
// order and the same result as the single-workgroup kernel:
//   3 x { topk_hist_kernel (all CUs: 11/11/10-bit digit histogram of the keys that still match the prefix)
//         topk_pick_kernel (one wave: bucket holding the top_n-th key -> extends the prefix) }
//   topk_count_eq_kernel  per-workgroup count of keys == kth over a contiguous index range
//   topk_compact_kernel   keys < kth (any slot) and the first need_eq ties (ordered slots: ties are taken lowest index
//                         first, each workgroup adds up the counts of the workgroups before it) -> u64 candidates
//   topk_rank_sort_kernel   every CU ranks 32 candidates against all <= 16384 of them; outputs in the canonical order
// ------------------------------------------------------------------------------------------------
constexpr int TOPK_BINS = 2048;
constexpr int TOPK_BLOCK_ITEMS = 4096;  // scores per workgroup (256 threads x 16)

struct TopkState {   // lives at the start of the workspace
  uint32_t prefix;   // digits of the kth key decided so far (high bits)
  uint32_t need;     // rank still to resolve inside the current prefix
  uint32_t fill;     // slots handed out to keys < kth
  uint32_t pad;
  uint32_t hist[TOPK_BINS];
};

__device__ __forceinline__ void topk_pass_bits(int pass, int& shift, int& bits) {
  shift = pass == 0 ? 21 : pass == 1 ? 10 : 0;
  bits = pass == 2 ? 10 : 11;
}

__global__ __launch_bounds__(256) void topk_init_kernel(TopkState* st, int take) {
  for (int i = threadIdx.x; i < TOPK_BINS; i += 256) st->hist[i] = 0;
  if (threadIdx.x == 0) { st->prefix = 0; st->need = (uint32_t)take; st->fill = 0; st->pad = 0; }
}

__global__ __launch_bounds__(256) void topk_hist_kernel(const float* __restrict__ scores, int n, int pass,
                                                       TopkState* __restrict__ st) {
  __shared__ uint32_t h[TOPK_BINS];
  for (int i = threadIdx.x; i < TOPK_BINS; i += 256) h[i] = 0;
  __syncthreads();
  int shift, bits;
  topk_pass_bits(pass, shift, bits);
  const uint32_t prefix = st->prefix;
  const uint32_t himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + bits));
  const int lo = blockIdx.x * TOPK_BLOCK_ITEMS, hi = min(lo + TOPK_BLOCK_ITEMS, n);
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const uint32_t k = desc_key(scores[i]);
    if ((k & himask) == prefix) atomicAdd(&h[(k >> shift) & ((1u << bits) - 1u)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TOPK_BINS; i += 256)
    if (h[i]) atomicAdd(&st->hist[i], h[i]);
}

__global__ __launch_bounds__(64) void topk_pick_kernel(TopkState* __restrict__ st, int pass) {
  // one wave: lane l owns bins [32l, 32l+32); exclusive prefix over the lanes via shuffles, then the lane that
  // contains the target rank walks its 32 bins
  int shift, bits;
  topk_pass_bits(pass, shift, bits);
  const int lane = threadIdx.x;
  uint32_t mine = 0;
  for (int b = 0; b < 32; ++b) mine += st->hist[lane * 32 + b];
  uint32_t incl = mine;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  const uint32_t excl = incl - mine, need = st->need;
  uint32_t new_prefix = 0, new_need = 0;
  const bool owner = need > excl && need <= incl;
  if (owner) {
    uint32_t acc = excl;
    int b = 0;
    for (; b < 32; ++b) {
      const uint32_t c = st->hist[lane * 32 + b];
      if (acc + c >= need) break;
      acc += c;
    }
    new_prefix = st->prefix | ((uint32_t)(lane * 32 + b) << shift);
    new_need = need - acc;
  }
  __syncthreads();   // single wave: orders the reads of hist above against the clears below
  for (int b = 0; b < 32; ++b) st->hist[lane * 32 + b] = 0;
  if (owner) { st->prefix = new_prefix; st->need = new_need; }
}

__global__ __launch_bounds__(256) void topk_count_eq_kernel(const float* __restrict__ scores, int n,
                                                           const TopkState* __restrict__ st,
                                                           uint32_t* __restrict__ block_eq) {
  __shared__ uint32_t s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const uint32_t kth = st->prefix;
  const int lo = blockIdx.x * TOPK_BLOCK_ITEMS, hi = min(lo + TOPK_BLOCK_ITEMS, n);
  uint32_t c = 0;
  for (int i = lo + threadIdx.x; i < hi; i += 256) c += desc_key(scores[i]) == kth ? 1u : 0u;
  if (c) atomicAdd(&s_cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) block_eq[blockIdx.x] = s_cnt;
}

__global__ __launch_bounds__(256) void topk_compact_kernel(const float* __restrict__ scores, int n, int take,
                                                          TopkState* __restrict__ st,
                                                          const uint32_t* __restrict__ block_eq,
                                                          uint64_t* __restrict__ cand) {
  // thread t owns the contiguous range [lo + 16t, lo + 16t + 16): ties stay in index order.  Slots for the keys < kth
  // are handed out per WORKGROUP (one global atomic each, after a block scan of the per-thread counts) - one atomic per
  // key on a single address was 10 of this kernel's 16 us.
  __shared__ uint32_t scan_lt[256], scan_eq[256];
  __shared__ uint32_t s_before, s_base;
  const uint32_t kth = st->prefix, need_eq = st->need;
  const uint32_t base_eq = (uint32_t)take - need_eq;   // number of keys < kth
  const int lo = blockIdx.x * TOPK_BLOCK_ITEMS + threadIdx.x * 16;
  const int hi = min(lo + 16, min((blockIdx.x + 1) * TOPK_BLOCK_ITEMS, n));
  uint32_t key[16];
  uint32_t my_lt = 0, my_eq = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = lo + q;
    key[q] = i < hi ? desc_key(scores[i]) : 0xFFFFFFFFu;
    if (i < hi) {
      my_lt += key[q] < kth ? 1u : 0u;
      my_eq += key[q] == kth ? 1u : 0u;
    }
  }
  scan_lt[threadIdx.x] = my_lt;
  scan_eq[threadIdx.x] = my_eq;
  if (threadIdx.x == 0) s_before = 0;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const uint32_t v = threadIdx.x >= off ? scan_lt[threadIdx.x - off] : 0u;
    const uint32_t w = threadIdx.x >= off ? scan_eq[threadIdx.x - off] : 0u;
    __syncthreads();
    scan_lt[threadIdx.x] += v;
    scan_eq[threadIdx.x] += w;
    __syncthreads();
  }
  // ties before this workgroup's range: sum of the earlier workgroups' counts (<= 256 of them; a separate one-workgroup
  // scan kernel did this before, 4.4 us of launch for a handful of adds)
  {
    uint32_t part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) part += block_eq[b];
    if (part) atomicAdd(&s_before, part);
  }
  if (threadIdx.x == 255) s_base = scan_lt[255] ? atomicAdd(&st->fill, scan_lt[255]) : 0u;
  __syncthreads();
  uint32_t pos = s_base + scan_lt[threadIdx.x] - my_lt;
  uint32_t rank = s_before + scan_eq[threadIdx.x] - my_eq;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = lo + q;
    if (i >= hi) break;
    if (key[q] < kth) {
      cand[pos++] = ((uint64_t)key[q] << 32) | (uint32_t)i;
    } else if (key[q] == kth && rank < need_eq) {
      cand[base_eq + rank] = ((uint64_t)key[q] << 32) | (uint32_t)i;
      ++rank;
    }
  }
}

// Final ordering of the <= 16384 selected candidates by RANK: keys are unique (the index is in the low word), so the
// position of a key in the sorted order is the number of candidates smaller than it.  One workgroup ranks 32 keys:
// thread (key k, partition p) counts the keys of partition p below key k against an LDS copy of all candidates
// (32 lanes read the same LDS word: broadcast), the 32 partial counts of a key are summed, and the key is scattered to
// its rank.  6000 candidates: 188 workgroups of 1024 threads, one round on the chip, 188 compares per thread: 12.7 us
// (16 keys x 16 partitions: 375 workgroups in two rounds, 26 us; the round-1 bitonic network in ONE workgroup: 78 us).
constexpr int RANK_KEYS = 32, RANK_PARTS = 32;
__global__ __launch_bounds__(RANK_KEYS * RANK_PARTS) void topk_rank_sort_kernel(const float* __restrict__ scores,
                                                                              const uint64_t* __restrict__ cand, int take,
                                                                              int64_t* __restrict__ order_out,
                                                                              float* __restrict__ scores_out,
                                                                              int* __restrict__ count_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sort_smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(sort_smem);          // [take]
  __shared__ uint32_t part[RANK_PARTS][RANK_KEYS];
  for (int i = threadIdx.x; i < take; i += blockDim.x) keys[i] = cand[i];
  __syncthreads();
  const int k = threadIdx.x & (RANK_KEYS - 1), p = threadIdx.x / RANK_KEYS;
  const int ki = blockIdx.x * RANK_KEYS + k;
  const uint64_t mine = ki < take ? keys[ki] : 0ull;
  const int per = (take + RANK_PARTS - 1) / RANK_PARTS;
  const int lo = min(p * per, take), hi = min(lo + per, take);
  uint32_t below = 0;
  for (int j = lo; j < hi; ++j) below += keys[j] < mine ? 1u : 0u;
  part[p][k] = below;
  __syncthreads();
  if (p == 0 && ki < take) {
    uint32_t rank = 0;
#pragma unroll
    for (int q = 0; q < RANK_PARTS; ++q) rank += part[q][k];
    const uint32_t idx = (uint32_t)(mine & 0xFFFFFFFFu);
    order_out[rank] = (int64_t)idx;
    scores_out[rank] = scores[idx];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) count_out[0] = take;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ rows,
                                                         const int64_t* __restrict__ order,
                                                         const int* __restrict__ count, int max_count, int width,
                                                         float* __restrict__ out) {
  const int cnt = count ? min(*count, max_count) : max_count;
  const int total = max_count * width;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / width, c = i - r * width;
    out[i] = r < cnt ? rows[order[r] * width + c] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
// NMS (torchvision.ops.nms semantics) on score-ordered boxes: bit-matrix + single-wave greedy scan.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ n_dev,
                                                     int n_max, int nb, float thresh, uint64_t* __restrict__ mask,
                                                     uint64_t* __restrict__ diag_t) {
  // 1-D grid over the blocks on / right of the diagonal only (row bi holds nb - bi of them): the square grid spent its
  // time dispatching the empty half
  int bi, bj;
  {
    const int t = blockIdx.x;
    const float a = 2.0f * nb + 1.0f;
    bi = (int)((a - sqrtf(a * a - 8.0f * t)) * 0.5f);
    bi = max(0, min(bi, nb - 1));
    while (bi > 0 && bi * (2 * nb - bi + 1) / 2 > t) --bi;
    while ((bi + 1) * (2 * nb - bi) / 2 <= t) ++bi;
    bj = bi + (t - bi * (2 * nb - bi + 1) / 2);
  }
  const int n = n_dev ? min(*n_dev, n_max) : n_max;
  if (bi * 64 >= n) return;
  __shared__ float cols[64 * 4];
  const int lane = threadIdx.x;
  const int j0 = bj * 64;
  if (j0 + lane < n) {
    const float4 b = reinterpret_cast<const float4*>(boxes)[j0 + lane];
    cols[lane * 4 + 0] = b.x; cols[lane * 4 + 1] = b.y; cols[lane * 4 + 2] = b.z; cols[lane * 4 + 3] = b.w;
  }
  __syncthreads();
  const int i = bi * 64 + lane;
  if (i >= n) return;
  const float4 bq = reinterpret_cast<const float4*>(boxes)[i];
  const float me[4] = {bq.x, bq.y, bq.z, bq.w};
  uint64_t bits = 0, pre = 0;
  const int jn = min(64, n - j0);
  if (bi == bj && diag_t) {
    // diagonal block: the IoU test is symmetric, so the same pass also yields the PREDECESSOR word of box i (the boxes
    // of its own chunk ranked above it that overlap it), which is what the scan's parallel resolve reads
    for (int b = 0; b < jn; ++b) {
      if (j0 + b == i) continue;
      const bool hit = iou_gt(me, &cols[b * 4], thresh);
      if (hit) {
        if (j0 + b > i) bits |= 1ull << b;
        else pre |= 1ull << b;
      }
    }
    diag_t[i] = pre;
  } else {
    for (int b = 0; b < jn; ++b) {
      if (j0 + b > i && iou_gt(me, &cols[b * 4], thresh)) bits |= 1ull << b;
    }
  }
  mask[(size_t)i * nb + bj] = bits;
}

// ------------------------------------------------------------------------------------------------
// Greedy scan of the RPN NMS (torchvision.ops.nms at proposal_layer.py:46) over the bit-matrix, one workgroup:
// wave 0 walks the 64-box chunks, waves 1..15 only help fetching rows.  Per chunk:
//  * resolve - which boxes of the chunk survive - is the fixed point of  keep(j) <=> no kept predecessor overlaps j
//    over the chunk's 64 boxes, found in a few ballot rounds from the transposed diagonal word nms_mask_kernel wrote
//    (lane j: the boxes of the chunk ranked above j that overlap it).  The serial form (one step of ~170 cycles per
//    SURVIVOR) cost 51K cycles when an untrained RPN keeps 60 boxes of every chunk;
//  * the diagonal words and the two following words (c+1, c+2) of every row are requested FOUR chunks ahead - nothing
//    the scan computes feeds those addresses;
//  * a chunk with few survivors (<= 8, the trained-RPN regime: thousands of boxes, almost all suppressed) stays inside
//    wave 0: what its survivors remove from chunks c+1 / c+2 is an OR over registers (carry_a / carry_b), their words
//    >= c+3 are requested now and merged into the removed set two chunks later (ring of two pending batches);
//  * a chunk with many survivors wakes the helper waves: wave w fetches the rows of the survivors at bit positions
//    w, w+8, ..., w+56 and ORs them into its LDS slice, wave 0 combines the 8 slices - two memory latencies per chunk
//    instead of one per four rows (86K of the scan's 142K cycles before).
// Same survivors in the same order as wave_nms_scan (box_math.h, still used by the general per-class filter).
// The chunk body is stamped out four times by a macro so that every ring slot is a fixed set of registers.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_WAVES = 8;       // 512 threads: wave 0 may use 256 VGPRs (its rings), no scratch
constexpr int SCAN_DENSE = 8;      // more survivors than this in a chunk: cooperative row fetch

// rows of the survivors this wave owns (bit positions wave, wave + 8, ...: two batches of four rows), words > c,
// OR-ed per word into part[w]
__device__ __forceinline__ void scan_coop_fetch(const uint64_t* __restrict__ mask, int nb, int c, uint64_t kept, int wave,
                                                int lane, uint64_t* part) {
  uint64_t acc[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (((kept >> (wave + 32 * half)) & 0x01010101ull) == 0ull) continue;   // none of this batch's four rows survives
    uint64_t v[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int b = wave + SCAN_WAVES * (4 * half + q);
      const bool on = (kept >> b) & 1ull;                      // wave-uniform
      const uint64_t* row = mask + (size_t)(c * 64 + (on ? b : wave)) * nb;   // any valid row when off (masked below)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        if (s4 * 64 < nb) {
          const int w = lane + 64 * s4;
          const uint64_t val = row[min(w, nb - 1)];
          v[q][s4] = (on && w > c && w < nb) ? val : 0ull;
        } else {
          v[q][s4] = 0ull;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc[s4] |= v[q][s4];
  }
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4)
    if (s4 * 64 < nb) part[lane + 64 * s4] = acc[s4];
}

#define FRCNN_SCAN_FETCH4(V, FIRST_WORD)                                               \
  _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) {                                   \
    if (s4 * 64 < nb) {                                                                \
      const int w_ = lane + 64 * s4;                                                   \
      const int wc_ = min(w_, nb - 1);                                                 \
      const bool ok_ = w_ >= (FIRST_WORD) && w_ < nb;                                  \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) V[q][s4] = rows[q][wc_];           \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) V[q][s4] = ok_ ? V[q][s4] : 0ull;  \
    } else {                                                                           \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) V[q][s4] = 0ull;                   \
    }                                                                                  \
  }
#define FRCNN_SCAN_MERGE4(V)                      \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) { \
    rm0 |= V[q][0];                               \
    rm1 |= V[q][1];                               \
    rm2 |= V[q][2];                               \
    rm3 |= V[q][3];                               \
  }
#define FRCNN_SCAN_ROWS4()                                                      \
  const uint64_t* rows[4];                                                      \
  {                                                                             \
    int nr_ = 0;                                                                \
    for (; nr_ < 4 && todo != 0; ++nr_) {                                       \
      rows[nr_] = mask + (size_t)(c * 64 + __builtin_ctzll(todo)) * nb;         \
      todo &= todo - 1ull;                                                      \
    }                                                                           \
    for (int q = nr_; q < 4; ++q) rows[q] = rows[0]; /* OR is idempotent */     \
  }
#define FRCNN_SCAN_CHUNK(C, PEND, K)                                                                                \
  {                                                                                                                 \
    const int c = (C);                                                                                              \
    if (!(c < nb && c * 64 < n && count < max_keep)) break;                                                         \
    const int i = c * 64 + lane;                                                                                    \
    const uint64_t pre = pq[K], n1 = n1q[K], n2 = n2q[K];                                                           \
    {                                                                                                               \
      const int in = i + 64 * 4, cn = c + 4;                                                                        \
      pq[K] = in < n ? diag_t[in] : 0ull;                                                                           \
      n1q[K] = (in < n && cn + 1 < nb) ? mask[(size_t)in * nb + (cn + 1)] : 0ull;                                   \
      n2q[K] = (in < n && cn + 2 < nb) ? mask[(size_t)in * nb + (cn + 2)] : 0ull;                                   \
    }                                                                                                               \
    const int slot = c >> 6;                                                                                        \
    const uint64_t mine = slot == 0 ? rm0 : slot == 1 ? rm1 : slot == 2 ? rm2 : rm3;                                \
    const uint64_t rw = readlane_u64(mine, c & 63) | carry_a;                                                       \
    const int live = n - c * 64;                                                                                    \
    const uint64_t valid = live >= 64 ? ~0ull : ((1ull << live) - 1ull);                                            \
    const uint64_t alive = ~rw & valid;                                                                             \
    /* fixed point of the greedy rule over the chunk (boxes removed by earlier chunks count as removed) */           \
    uint64_t kept = 0, rem = ~alive;                                                                                \
    bool und = (alive >> lane) & 1ull;                                                                              \
    for (;;) {                                                                                                      \
      const bool k_ = und && (pre & ~rem) == 0ull;                                                                  \
      const bool r_ = und && (pre & kept) != 0ull;                                                                  \
      const uint64_t kb = __ballot(k_), rb = __ballot(r_);                                                          \
      if ((kb | rb) == 0ull) break;                                                                                 \
      kept |= kb;                                                                                                   \
      rem |= rb;                                                                                                    \
      und = und && !k_ && !r_;                                                                                      \
    }                                                                                                               \
    int nk = __builtin_popcountll(kept);                                                                            \
    while (count + nk > max_keep) { /* the cap cuts the lowest-ranked survivors of the last chunk */                \
      kept &= ~(1ull << (63 - __builtin_clzll(kept)));                                                              \
      --nk;                                                                                                         \
    }                                                                                                               \
    if ((kept >> lane) & 1ull) {                                                                                    \
      const int pos = count + __builtin_popcountll(kept & ((1ull << lane) - 1ull));                                 \
      keep_idx[pos] = i;                                                                                            \
      if (keep_mask) keep_mask[i] = 1;                                                                              \
    }                                                                                                               \
    count += nk;                                                                                                    \
    if (count >= max_keep) break;                                                                                   \
    /* the batch requested two chunks ago (words >= c+1 of chunk c-2's survivors) lands in the removed set now */   \
    FRCNN_SCAN_MERGE4(PEND)                                                                                         \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) PEND[q][s4] = 0ull; \
    if (nk > SCAN_DENSE) {                                                                                          \
      if (lane == 0) {                                                                                              \
        s_kept = kept;                                                                                              \
        s_cmd = c;                                                                                                  \
      }                                                                                                             \
      __syncthreads();                                                                                              \
      scan_coop_fetch(mask, nb, c, kept, 0, lane, s_part[0]);                                                       \
      __syncthreads();                                                                                              \
      _Pragma("unroll") for (int s4 = 0; s4 < 4; ++s4) {                                                            \
        if (s4 * 64 < nb) {                                                                                         \
          uint64_t a_ = 0ull;                                                                                       \
          _Pragma("unroll") for (int w_ = 0; w_ < SCAN_WAVES; ++w_) a_ |= s_part[w_][lane + 64 * s4];               \
          if (s4 == 0) rm0 |= a_;                                                                                   \
          else if (s4 == 1) rm1 |= a_;                                                                              \
          else if (s4 == 2) rm2 |= a_;                                                                              \
          else rm3 |= a_;                                                                                           \
        }                                                                                                           \
      }                                                                                                             \
      carry_a = carry_b; /* this chunk's words c+1, c+2 are in rm already */                                        \
      carry_b = 0ull;                                                                                               \
    } else {                                                                                                        \
      uint64_t k1 = 0ull, k2 = 0ull;                                                                                \
      for (uint64_t t_ = kept; t_ != 0ull; t_ &= t_ - 1ull) {                                                       \
        const int b = __builtin_ctzll(t_);                                                                          \
        k1 |= readlane_u64(n1, b);                                                                                  \
        k2 |= readlane_u64(n2, b);                                                                                  \
      }                                                                                                             \
      carry_a = carry_b | k1; /* word c+1: chunk c-1's second look-ahead word and this chunk's first */             \
      carry_b = k2;           /* word c+2 */                                                                        \
      uint64_t todo = kept;                                                                                         \
      if (todo != 0ull) {                                                                                           \
        FRCNN_SCAN_ROWS4()                                                                                          \
        FRCNN_SCAN_FETCH4(PEND, c + 3) /* merged at the end of chunk c+2 */                                         \
      }                                                                                                             \
      if (todo != 0ull) { /* survivors five to eight: waited for */                                                 \
        FRCNN_SCAN_ROWS4()                                                                                          \
        uint64_t v[4][4];                                                                                           \
        FRCNN_SCAN_FETCH4(v, c + 3)                                                                                 \
        FRCNN_SCAN_MERGE4(v)                                                                                        \
      }                                                                                                             \
    }                                                                                                               \
  }

__global__ __launch_bounds__(64 * SCAN_WAVES) void nms_scan_kernel(const uint64_t* __restrict__ mask,
                                                                  const uint64_t* __restrict__ diag_t,
                                                                  const int* __restrict__ n_dev, int n_max, int nb,
                                                                  int max_keep, int64_t* __restrict__ keep_idx,
                                                                  uint8_t* __restrict__ keep_mask,
                                                                  int* __restrict__ keep_count) {
  __shared__ uint64_t s_part[SCAN_WAVES][256];
  __shared__ uint64_t s_kept;
  __shared__ int s_cmd;
  const int n = n_dev ? min(*n_dev, n_max) : n_max;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave != 0) {
    // helpers sleep at the barrier until wave 0 posts a chunk (s_cmd >= 0) or the end of the scan (s_cmd < 0)
    for (;;) {
      __syncthreads();
      const int c = s_cmd;
      if (c < 0) return;
      scan_coop_fetch(mask, nb, c, s_kept, wave, lane, s_part[wave]);
      __syncthreads();
    }
  }
  uint64_t rm0 = 0, rm1 = 0, rm2 = 0, rm3 = 0;   // removed bits of words lane, lane+64, lane+128, lane+192
  uint64_t carry_a = 0, carry_b = 0;             // removed bits of words c and c+1 not yet merged into rm
  uint64_t pend_a[4][4], pend_b[4][4];           // ring of two pending batches [row][word slot]: even / odd chunks
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) pend_a[q][s4] = pend_b[q][s4] = 0ull;
  int count = 0;
  uint64_t pq[4], n1q[4], n2q[4];                // predecessor word + two look-ahead words of the next four chunks' rows
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = k * 64 + lane;
    pq[k] = i < n ? diag_t[i] : 0ull;
    n1q[k] = (i < n && k + 1 < nb) ? mask[(size_t)i * nb + (k + 1)] : 0ull;
    n2q[k] = (i < n && k + 2 < nb) ? mask[(size_t)i * nb + (k + 2)] : 0ull;
  }
  for (int c4 = 0;; c4 += 4) {
    FRCNN_SCAN_CHUNK(c4, pend_a, 0)
    FRCNN_SCAN_CHUNK(c4 + 1, pend_b, 1)
    FRCNN_SCAN_CHUNK(c4 + 2, pend_a, 2)
    FRCNN_SCAN_CHUNK(c4 + 3, pend_b, 3)
  }
  for (int i = count + lane; i < max_keep; i += 64) keep_idx[i] = 0;   // the unused tail is defined (callers pass empty buffers)
  if (lane == 0) {
    s_cmd = -1;
    keep_count[0] = count;
  }
  __syncthreads();   // releases the helpers
}
#undef FRCNN_SCAN_CHUNK
#undef FRCNN_SCAN_ROWS4
#undef FRCNN_SCAN_MERGE4
#undef FRCNN_SCAN_FETCH4

__global__ __launch_bounds__(256) void make_rois_kernel(const float* __restrict__ boxes, const float* __restrict__ scs,
                                                       const int64_t* __restrict__ keep_idx,
                                                       const int* __restrict__ keep_count, int max_keep,
                                                       float* __restrict__ rois, float* __restrict__ roi_scores) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= max_keep) return;
  const int cnt = min(*keep_count, max_keep);
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  float s = 0.f;
  if (i < cnt) {
    const int64_t k = keep_idx[i];
    b = reinterpret_cast<const float4*>(boxes)[k];
    s = scs[k];
  }
  float* r = rois + (size_t)i * 5;
  r[0] = 0.f; r[1] = b.x; r[2] = b.y; r[3] = b.z; r[4] = b.w;
  if (roi_scores) roi_scores[i] = s;
}

// ------------------------------------------------------------------------------------------------
// filter_and_draw_prep — lib/utils/filter_predictions.py:75-130 (+ test.py:210-221 max_dets cut).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void clamp_pred_boxes_kernel(float* __restrict__ boxes, int total_boxes, float x_hi,
                                                              float y_hi) {
  // filter_predictions.py:85-91: x1,y1 = clamp_min(0); x2 = clamp_max(frame_w/scale - 1); y2 likewise
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total_boxes; i += gridDim.x * blockDim.x) {
    float4 b = reinterpret_cast<float4*>(boxes)[i];
    b.x = clamp_minf(b.x, 0.f);
    b.y = clamp_minf(b.y, 0.f);
    b.z = clamp_maxf(b.z, x_hi);
    b.w = clamp_maxf(b.w, y_hi);
    reinterpret_cast<float4*>(boxes)[i] = b;
  }
}

constexpr int FILTER_THREADS = 256;

// One workgroup per foreground class.  ws per class: sorted boxes [R][4] floats, mask [R][nb] u64,
// keep_idx [R] int64.  E = 4: image boxes; E = 7: LiDAR boxes [xc,yc,zc,l,w,h,ry], suppressed on the
// yaw-less BEV rectangle xc -+ l/2, yc -+ w/2 (filter_predictions.py:55-62,67); rows of dets are E+1 wide.
// Class 0 (background) never has detections (filter_predictions.py:45 loops over classes 1..K-1): block 0 of the
// per-class filter kernels writes that class's slice of every output, so the callers need no zero-filled buffers.
template <int E>
__device__ __forceinline__ void filter_fill_background(int t, int threads, int max_out, float* __restrict__ dets,
                                                       int* __restrict__ det_count, int* __restrict__ det_roi) {
  for (int e = t; e < max_out * (E + 1); e += threads) dets[e] = 0.f;
  if (det_roi)
    for (int e = t; e < max_out; e += threads) det_roi[e] = -1;
  if (t == 0) det_count[0] = 0;
}

template <int E>
__global__ __launch_bounds__(FILTER_THREADS) void filter_class_kernel(
    const float* __restrict__ pred_boxes, const float* __restrict__ cls_prob, const int* __restrict__ roi_count,
    int num_rois, int num_classes, float thresh, float nms_thresh, int max_dets, int max_out, int npad, int nb,
    float* __restrict__ dets, int* __restrict__ det_count, int* __restrict__ det_roi, unsigned char* __restrict__ ws,
    size_t ws_per_class) {
  extern __shared__ __attribute__((aligned(16))) unsigned char filt_smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(filt_smem);  // [npad]
  __shared__ int s_n, s_keep;
  const int cls = blockIdx.x;
  const int t = threadIdx.x;
  if (cls == 0) {   // background rows of the outputs (see filter_fill_background)
    filter_fill_background<E>(t, FILTER_THREADS, max_out, dets, det_count, det_roi);
    return;
  }
  const int R = roi_count ? min(*roi_count, num_rois) : num_rois;

  unsigned char* my = ws + (size_t)(blockIdx.x - 1) * ws_per_class;
  float* sboxes = reinterpret_cast<float*>(my);
  uint64_t* mask = reinterpret_cast<uint64_t*>(my + align_up((size_t)num_rois * 16, 16));
  int64_t* keep_idx = reinterpret_cast<int64_t*>(reinterpret_cast<unsigned char*>(mask) + (size_t)num_rois * nb * 8);

  if (t == 0) s_n = 0;
  for (int i = t; i < npad; i += FILTER_THREADS) keys[i] = ~0ull;
  __syncthreads();
  // inds = scores[:, c] > thresh  (filter_predictions.py:46)
  for (int r = t; r < R; r += FILTER_THREADS) {
    const float s = cls_prob[(size_t)r * num_classes + cls];
    if (s > thresh) {
      const int pos = atomicAdd(&s_n, 1);
      keys[pos] = ((uint64_t)desc_key(s) << 32) | (uint32_t)r;
    }
  }
  __syncthreads();
  const int n = s_n;
  block_bitonic_sort(keys, npad);  // (score desc, roi index asc)
  for (int i = t; i < n; i += FILTER_THREADS) {
    const uint32_t r = (uint32_t)(keys[i] & 0xFFFFFFFFu);
    const float* pb = pred_boxes + ((size_t)r * num_classes + cls) * E;
    if (E == 4) {
      reinterpret_cast<float4*>(sboxes)[i] = make_float4(pb[0], pb[1], pb[2], pb[3]);
    } else {
      reinterpret_cast<float4*>(sboxes)[i] =
          make_float4(pb[0] - pb[3] / 2.0f, pb[1] - pb[4] / 2.0f, pb[0] + pb[3] / 2.0f, pb[1] + pb[4] / 2.0f);
    }
  }
  __syncthreads();
  // suppression bit-matrix (words on/right of the diagonal)
  const int nbl = (n + 63) / 64;
  for (int wi = t; wi < n * nbl; wi += FILTER_THREADS) {
    const int i = wi / nbl, w = wi - i * nbl;
    if (w < (i >> 6)) continue;
    uint64_t bits = 0;
    const int jn = min(64, n - w * 64);
    for (int b = 0; b < jn; ++b) {
      const int j = w * 64 + b;
      if (j > i && iou_gt(&sboxes[i * 4], &sboxes[j * 4], nms_thresh)) bits |= 1ull << b;
    }
    mask[(size_t)i * nbl + w] = bits;
  }
  __syncthreads();
  if (t < 64) {
    const int cnt = wave_nms_scan(mask, nbl, n, n, keep_idx, nullptr);
    if (t == 0) s_keep = cnt;
  }
  __syncthreads();
  int kept = s_keep;
  // test.py:213-221: if more than max_dets survive keep score >= the max_dets-th best (ties stay)
  if (max_dets > 0 && kept > max_dets) {
    const float cut = cls_prob[(size_t)(uint32_t)(keys[keep_idx[max_dets - 1]] & 0xFFFFFFFFu) * num_classes + cls];
    int m = max_dets;
    while (m < kept &&
           cls_prob[(size_t)(uint32_t)(keys[keep_idx[m]] & 0xFFFFFFFFu) * num_classes + cls] >= cut)
      ++m;
    kept = m;
  }
  kept = min(kept, max_out);
  float* out = dets + (size_t)cls * max_out * (E + 1);
  for (int i = t; i < max_out; i += FILTER_THREADS) {
    float v[E + 1];
    for (int q = 0; q <= E; ++q) v[q] = 0.f;
    int roi = -1;
    if (i < kept) {
      const int64_t k = keep_idx[i];
      const uint32_t r = (uint32_t)(keys[k] & 0xFFFFFFFFu);
      const float* pb = pred_boxes + ((size_t)r * num_classes + cls) * E;
      for (int q = 0; q < E; ++q) v[q] = pb[q];
      v[E] = cls_prob[(size_t)r * num_classes + cls];
      roi = (int)r;
    }
    for (int q = 0; q <= E; ++q) out[i * (E + 1) + q] = v[q];
    if (det_roi) det_roi[(size_t)cls * max_out + i] = roi;
  }
  if (t == 0) det_count[cls] = kept;
}

// Same per-class filter for num_rois <= 1024 (every detector of the reference: 300 test-time RoIs), entirely in LDS
// with 16 waves.  The general kernel above spent 161 us per frame on ONE workgroup of 4 waves: a 45-stage bitonic
// sort of 512 keys, a suppression matrix with one thread looping over the 64 IoUs of a word, and a greedy scan whose
// row fetches each paid a global-memory round trip.  Here: keys are ranked (unique keys: position = number of smaller
// keys, n broadcast LDS reads per thread, no stage barriers), one WAVE builds a matrix word with one IoU per lane and a
// ballot, and the matrix (n x ceil(n/64) words) stays in LDS for the scan.  Same order, same IoU test, same outputs.
constexpr int FILTER_SMALL_THREADS = 1024, FILTER_SMALL_MAX = 1024;

template <int E>
__global__ __launch_bounds__(FILTER_SMALL_THREADS) void filter_class_small_kernel(
    const float* __restrict__ pred_boxes, const float* __restrict__ cls_prob, const int* __restrict__ roi_count,
    int num_rois, int num_classes, float thresh, float nms_thresh, int max_dets, int max_out, float* __restrict__ dets,
    int* __restrict__ det_count, int* __restrict__ det_roi) {
  extern __shared__ __attribute__((aligned(16))) unsigned char filt_smem[];
  // LDS: raw keys [num_rois] | sorted keys [num_rois] | boxes [num_rois] float4 | keep_idx [num_rois] i64 | mask [n][nbl] u64
  uint64_t* raw = reinterpret_cast<uint64_t*>(filt_smem);
  uint64_t* keys = raw + num_rois;
  float4* sboxes = reinterpret_cast<float4*>(keys + num_rois);
  int64_t* keep_idx = reinterpret_cast<int64_t*>(sboxes + num_rois);
  uint64_t* mask = reinterpret_cast<uint64_t*>(keep_idx + num_rois);
  __shared__ int s_n;
  const int cls = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (cls == 0) {   // the background class has no detections: its rows of the outputs are written here, not by the caller
    filter_fill_background<E>(t, FILTER_SMALL_THREADS, max_out, dets, det_count, det_roi);
    return;
  }
  const int R = roi_count ? min(*roi_count, num_rois) : num_rois;
  if (t == 0) s_n = 0;
  __syncthreads();
  // inds = scores[:, c] > thresh  (filter_predictions.py:46)
  for (int r = t; r < R; r += FILTER_SMALL_THREADS) {
    const float sc = cls_prob[(size_t)r * num_classes + cls];
    if (sc > thresh) raw[atomicAdd(&s_n, 1)] = ((uint64_t)desc_key(sc) << 32) | (uint32_t)r;
  }
  __syncthreads();
  const int n = s_n;
  // (score desc, roi index asc): rank = number of smaller keys
  for (int i = t; i < n; i += FILTER_SMALL_THREADS) {
    const uint64_t mine = raw[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += raw[j] < mine ? 1 : 0;
    keys[rank] = mine;
    const uint32_t r = (uint32_t)(mine & 0xFFFFFFFFu);
    const float* pb = pred_boxes + ((size_t)r * num_classes + cls) * E;
    sboxes[rank] = E == 4 ? make_float4(pb[0], pb[1], pb[2], pb[3])
                          : make_float4(pb[0] - pb[3] / 2.0f, pb[1] - pb[4] / 2.0f, pb[0] + pb[3] / 2.0f,
                                        pb[1] + pb[4] / 2.0f);
  }
  __syncthreads();
  // PREDECESSOR matrix P[j] = bits i < j with IoU(i, j) > nms_thresh (boxes in descending-score order): one wave per
  // 64-box word of the upper triangle, one IoU per lane, the hits scattered to the transposed position by LDS atomics
  // (sparse).  Greedy NMS is then the unique fixed point of  keep(j) <=> no kept predecessor in P[j] , found by rounds
  // in which every undecided box looks at the decided sets: a kept predecessor removes it, all predecessors removed keeps
  // it.  Every round decides at least the first undecided box; a frame's boxes take a handful of rounds, against one
  // serial step per kept box in a scan (which cost 40 of this kernel's 75 us).
  const int nbl = (n + 63) / 64;
  uint64_t* const P = mask;
  __shared__ unsigned long long s_sets[2 * (FILTER_SMALL_MAX / 64)];
  unsigned long long* const kept_set = s_sets;                          // [nbl]
  unsigned long long* const rem_set = s_sets + FILTER_SMALL_MAX / 64;   // [nbl]
  for (int e = t; e < n * nbl; e += FILTER_SMALL_THREADS) P[e] = 0ull;
  __syncthreads();
  // work unit = (word w of boxes j, chunk c <= w of predecessors i, quarter of the chunk): lane j keeps its box in
  // registers, the 16 predecessor boxes are LDS broadcasts, the hits are collected in a register word - one LDS atomic
  // per lane and unit
  constexpr int PARTS = 4, ROWS = 64 / PARTS;
  const int units = nbl * (nbl + 1) / 2 * PARTS;
  for (int u = wave; u < units; u += FILTER_SMALL_THREADS / 64) {
    int pair = u / PARTS, w = 0;
    const int part = u - pair * PARTS;
    while (pair > w) { pair -= w + 1; ++w; }
    const int c = pair;
    const int j = w * 64 + lane;
    const float4 bj = sboxes[min(j, n - 1)];
    const float b[4] = {bj.x, bj.y, bj.z, bj.w};
    const int i0 = c * 64 + part * ROWS;
    uint64_t word = 0ull;
#pragma unroll 4
    for (int ii = 0; ii < ROWS; ++ii) {
      const int i = i0 + ii;
      if (i >= n) break;
      const float4 bi = sboxes[i];
      const float a[4] = {bi.x, bi.y, bi.z, bi.w};
      if (j < n && j > i && iou_gt_lazy_div(a, b, nms_thresh)) word |= 1ull << (i & 63);
    }
    if (word != 0ull) atomicOr(reinterpret_cast<unsigned long long*>(&P[(size_t)j * nbl + c]), word);
  }
  __syncthreads();
  if (t < 2 * (FILTER_SMALL_MAX / 64)) s_sets[t] = 0ull;
  bool undecided = t < n;
  const int my_words = (t >> 6) + 1;            // predecessors of box t live in words 0 .. t / 64
  for (;;) {
    __syncthreads();
    if (undecided) {
      bool any_kept = false, all_removed = true;
      for (int w = 0; w < my_words; ++w) {
        const uint64_t pre = P[(size_t)t * nbl + w];
        any_kept |= (pre & kept_set[w]) != 0ull;
        all_removed &= (pre & ~rem_set[w]) == 0ull;
      }
      if (any_kept) {
        atomicOr(&rem_set[t >> 6], 1ull << (t & 63));
        undecided = false;
      } else if (all_removed) {
        atomicOr(&kept_set[t >> 6], 1ull << (t & 63));
        undecided = false;
      }
    }
    if (__syncthreads_count(undecided ? 1 : 0) == 0) break;
  }
  // positions: survivors in ascending box order
  uint64_t my_kept_word = 0;
  int before = 0, total = 0;
  for (int w = 0; w < nbl; ++w) {
    const uint64_t kw = kept_set[w];
    const int pc = __builtin_popcountll(kw);
    if (w < (t >> 6)) before += pc;
    if (w == (t >> 6)) my_kept_word = kw;
    total += pc;
  }
  if (t < n && ((my_kept_word >> (t & 63)) & 1ull))
    keep_idx[before + __builtin_popcountll(my_kept_word & ((1ull << (t & 63)) - 1ull))] = t;
  __syncthreads();
  int kept = total;
  // test.py:213-221: if more than max_dets survive keep score >= the max_dets-th best (ties stay).  Keys ascend along
  // keep_idx, so the survivors of the cut are a prefix: count the ties in parallel
  if (max_dets > 0 && kept > max_dets) {
    const uint32_t cut = (uint32_t)(keys[keep_idx[max_dets - 1]] >> 32);     // ascending key = descending score
    int ties = 0;
    for (int base = max_dets; base < kept; base += FILTER_SMALL_THREADS) {
      const int m = base + t;
      ties += __syncthreads_count(m < kept && (uint32_t)(keys[keep_idx[m]] >> 32) <= cut);
    }
    kept = max_dets + ties;
  }
  kept = min(kept, max_out);
  float* out = dets + (size_t)cls * max_out * (E + 1);
  for (int e = t; e < max_out * (E + 1); e += FILTER_SMALL_THREADS) {     // one output element per thread: coalesced
    const int i = e / (E + 1), q = e - i * (E + 1);
    float v = 0.f;
    int roi = -1;
    if (i < kept) {
      const uint32_t r = (uint32_t)(keys[keep_idx[i]] & 0xFFFFFFFFu);
      v = q < E ? pred_boxes[((size_t)r * num_classes + cls) * E + q] : cls_prob[(size_t)r * num_classes + cls];
      roi = (int)r;
    }
    out[e] = v;
    if (det_roi && q == 0) det_roi[(size_t)cls * max_out + i] = roi;
  }
  if (t == 0) det_count[cls] = kept;
}

static size_t filter_small_lds(int num_rois) {
  const size_t nb = (size_t)(num_rois + 63) / 64;
  return (size_t)num_rois * (8 + 8 + 16 + 8) + (size_t)num_rois * nb * 8;
}

// bbox_transform_inv for (N boxes) x (Kc classes) — lib/model/bbox_transform.py:75-105 — and clip_boxes
// (:235-257) as stand-alone entry points (the proposal / head kernels fuse the same arithmetic).
__global__ __launch_bounds__(256) void bbox_transform_inv_kernel(const float* __restrict__ boxes, int box_ld,
                                                                const float* __restrict__ deltas, int n, int kc,
                                                                float scale, int use_scale,
                                                                float* __restrict__ out) {
  const int total = n * kc;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / kc;
    const float* b = boxes + (size_t)r * box_ld;
    float x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
    if (use_scale) { x1 = x1 / scale; y1 = y1 / scale; x2 = x2 / scale; y2 = y2 / scale; }
    const float4 d = reinterpret_cast<const float4*>(deltas)[i];
    float o[4];
    decode_box(x1, y1, x2, y2, d.x, d.y, d.z, d.w, o);
    reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// lidar_3d_bbox_transform_inv for (N rois) x (Kc classes) — lib/model/bbox_transform.py:174-233.
__global__ __launch_bounds__(256) void lidar_bbox_transform_inv_kernel(const float* __restrict__ rois, int roi_ld,
                                                                      const float* __restrict__ anchors3d,
                                                                      const float* __restrict__ deltas, int n, int kc,
                                                                      float scale, int use_scale,
                                                                      float* __restrict__ out) {
  const int total = n * kc;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / kc;
    const float* b = rois + (size_t)r * roi_ld;
    float x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3];
    if (use_scale) { x1 = x1 / scale; y1 = y1 / scale; x2 = x2 / scale; y2 = y2 / scale; }
    float d[7], o[7];
    for (int q = 0; q < 7; ++q) d[q] = deltas[(size_t)i * 7 + q];
    decode_box_lidar(x1, y1, x2, y2, anchors3d + (size_t)r * 7, d, o);
    for (int q = 0; q < 7; ++q) out[(size_t)i * 7 + q] = o[q];
  }
}

__global__ __launch_bounds__(256) void clip_boxes_kernel(const float* __restrict__ in, int total, ClipInfo clip,
                                                        float* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float4 b = reinterpret_cast<const float4*>(in)[i];
    b.x = clampf(b.x, clip.x_lo, clip.x_hi);
    b.y = clampf(b.y, clip.y_lo, clip.y_hi);
    b.z = clampf(b.z, clip.x_lo, clip.x_hi);
    b.w = clampf(b.w, clip.y_lo, clip.y_hi);
    reinterpret_cast<float4*>(out)[i] = b;
  }
}

// LevelMapper.__call__ (lib/utils/torchpoolers.py:39-51): FPN level of each RoI from its area (no +1):
// floor(lvl0 + log2(sqrt(area) / s0) + eps) clamped to [k_min, k_max], returned relative to k_min.
__global__ __launch_bounds__(256) void fpn_level_map_kernel(const float* __restrict__ rois, int n, int k_min, int k_max,
                                                           float s0, float lvl0, float eps, int* __restrict__ levels) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* b = rois + (size_t)i * 5 + 1;
    const float area = (b[2] - b[0]) * (b[3] - b[1]);
    const float s = sqrtf(area);
    float lvl = floorf(lvl0 + (float)log2((double)(s / s0)) + eps);
    lvl = fminf(fmaxf(lvl, (float)k_min), (float)k_max);   // NaN (negative area) falls to k_min like clamp(min=...)
    if (!(lvl >= (float)k_min)) lvl = (float)k_min;
    levels[i] = (int)lvl - k_min;
  }
}

int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

extern "C" int frcnn_generate_anchors(const double* base, int num_base, int height, int width, int feat_stride,
                                      float* anchors, void* stream_) {
  FRCNN_REQUIRE(base && anchors && num_base > 0 && height > 0 && width > 0 && feat_stride > 0,
                "generate_anchors: bad arguments");
  const int total = height * width * num_base;
  hipLaunchKernelGGL(anchors_kernel, dim3(std::min((total + 255) / 256, 2048)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), base, num_base, height, width, feat_stride, anchors);
  return check_launch("anchors_kernel");
}

extern "C" int frcnn_rpn_decode_clip(const float* rpn, int ld, const float* probs_in, const float* deltas_in,
                                     const float* anchors, const float* info_host, int hw, int num_anchors,
                                     float* scores, float* proposals, void* stream_) {
  FRCNN_REQUIRE(anchors && info_host && scores && proposals && hw > 0 && num_anchors > 0,
                "rpn_decode_clip: bad arguments");
  FRCNN_REQUIRE((rpn && ld >= 6 * num_anchors) || (probs_in && deltas_in),
                "rpn_decode_clip: need the fused rpn tensor (ld >= 6A) or probs_in + deltas_in");
  FRCNN_REQUIRE(rpn || (probs_in && deltas_in), "rpn_decode_clip: probs_in/deltas_in need each other without rpn");
  // clip_boxes (bbox_transform.py:252-255): x in [info[0], info[1]-1], y in [info[2], info[3]-1], fp32 maths
  ClipInfo clip{info_host[0], info_host[1] - 1.0f, info_host[2], info_host[3] - 1.0f};
  const int total = hw * num_anchors;
  hipLaunchKernelGGL(rpn_decode_clip_kernel, dim3(std::min((total + 255) / 256, 2048)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), rpn, ld, probs_in, deltas_in, anchors, clip, total, num_anchors,
                     scores, proposals);
  return check_launch("rpn_decode_clip_kernel");
}

constexpr int TOPK_MULTI_MIN_N = 16384;   // below this one workgroup does everything in LDS

static size_t topk_multi_layout(int n, size_t* off_blocks, size_t* off_cand) {
  const size_t nblocks = ((size_t)n + TOPK_BLOCK_ITEMS - 1) / TOPK_BLOCK_ITEMS;
  size_t o = align_up(sizeof(TopkState), 256);
  *off_blocks = o;
  o = align_up(o + nblocks * sizeof(uint32_t), 256);
  *off_cand = o;
  return o + (size_t)16384 * sizeof(uint64_t);
}

extern "C" size_t frcnn_sort_topk_desc_ws_bytes(int n, int top_n) {
  if (n <= TOPK_MULTI_MIN_N || n <= top_n) return 0;   // everything lives in LDS
  size_t a, b;
  return topk_multi_layout(n, &a, &b);
}

extern "C" int frcnn_sort_topk_desc(const float* scores, int n, int top_n, int64_t* order_out, float* scores_out,
                                    int* count_out, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FRCNN_REQUIRE(scores && order_out && scores_out && count_out && n > 0 && top_n > 0, "sort_topk_desc: bad arguments");
  FRCNN_REQUIRE(top_n <= 16384, "sort_topk_desc: top_n %d > 16384", top_n);
  const int take = std::min(n, top_n);
  const int npad = next_pow2(std::max(take, 2));
  if (n > TOPK_MULTI_MIN_N && n > top_n) {
    size_t off_blocks, off_cand;
    const size_t need = topk_multi_layout(n, &off_blocks, &off_cand);
    if (!ws || ws_bytes < need) return fail(FRCNN_ERR_WS, "sort_topk_desc: workspace %zu < %zu bytes", ws_bytes, need);
    char* base = static_cast<char*>(ws);
    TopkState* st = reinterpret_cast<TopkState*>(base);
    uint32_t* block_eq = reinterpret_cast<uint32_t*>(base + off_blocks);
    uint64_t* cand = reinterpret_cast<uint64_t*>(base + off_cand);
    const int nblocks = (n + TOPK_BLOCK_ITEMS - 1) / TOPK_BLOCK_ITEMS;
    hipLaunchKernelGGL(topk_init_kernel, dim3(1), dim3(256), 0, stream, st, take);
    for (int pass = 0; pass < 3; ++pass) {
      hipLaunchKernelGGL(topk_hist_kernel, dim3(nblocks), dim3(256), 0, stream, scores, n, pass, st);
      hipLaunchKernelGGL(topk_pick_kernel, dim3(1), dim3(64), 0, stream, st, pass);
    }
    hipLaunchKernelGGL(topk_count_eq_kernel, dim3(nblocks), dim3(256), 0, stream, scores, n, st, block_eq);
    hipLaunchKernelGGL(topk_compact_kernel, dim3(nblocks), dim3(256), 0, stream, scores, n, take, st, block_eq, cand);
    int rc = check_launch("topk multi-workgroup select");
    if (rc != FRCNN_OK) return rc;
    const size_t lds = (size_t)take * 8;
    static std::atomic<size_t> configured_f{0};
    if (lds > configured_f.load()) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&topk_rank_sort_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return fail(FRCNN_ERR_LAUNCH, "sort_topk_desc: set LDS size: %s", hipGetErrorString(e));
      configured_f.store(lds);
    }
    hipLaunchKernelGGL(topk_rank_sort_kernel, dim3((take + RANK_KEYS - 1) / RANK_KEYS), dim3(RANK_KEYS * RANK_PARTS), lds,
                       stream, scores, cand, take, order_out, scores_out, count_out);
    return check_launch("topk_rank_sort_kernel");
  }
  const size_t lds = (size_t)npad * 8 + 256 * 4 + SORT_THREADS * 4;
  static std::atomic<size_t> configured{0};
  if (lds > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sort_topk_desc_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(FRCNN_ERR_LAUNCH, "sort_topk_desc: set LDS size: %s", hipGetErrorString(e));
    configured = lds;
  }
  hipLaunchKernelGGL(sort_topk_desc_kernel, dim3(1), dim3(SORT_THREADS), lds, stream, scores, n, top_n, npad, order_out,
                     scores_out, count_out);
  return check_launch("sort_topk_desc_kernel");
}

extern "C" int frcnn_gather_rows(const float* rows, const int64_t* order, const int* count, int max_count, int width,
                                 float* rows_out, void* stream_) {
  FRCNN_REQUIRE(rows && order && rows_out && max_count > 0 && width > 0, "gather_rows: bad arguments");
  const int total = max_count * width;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(std::min((total + 255) / 256, 2048)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), rows, order, count, max_count, width, rows_out);
  return check_launch("gather_rows_kernel");
}

// IoU == threshold: 1 = the box is suppressed (torchvision 0.4.0's CPU kernel, `>=`; default), 0 = it survives (its CUDA
// kernel, `>`).  Read at launch time; a captured hipGraph keeps the setting it was captured under.
static std::atomic<int> g_nms_at_equal{1};
extern "C" int frcnn_nms_set_suppress_at_equal(int on) {
  g_nms_at_equal.store(on ? 1 : 0);
  return FRCNN_OK;
}
extern "C" int frcnn_nms_get_suppress_at_equal(void) { return g_nms_at_equal.load(); }
static inline float nms_kernel_thresh(float thresh) {
  return g_nms_at_equal.load() ? nextafterf(thresh, -INFINITY) : thresh;   // `iou > t` then means `iou >= thresh`
}

extern "C" size_t frcnn_nms_ws_bytes(int n_max) {
  if (n_max <= 0) return 0;
  const size_t nb = (size_t)(n_max + 63) / 64;
  return (size_t)n_max * nb * sizeof(uint64_t) + (size_t)n_max * sizeof(uint64_t);   // bit-matrix + predecessor words
}

extern "C" int frcnn_nms(const float* boxes, const int* n_dev, int n_max, float thresh, int max_keep,
                         int64_t* keep_idx, uint8_t* keep_mask, int* keep_count, void* ws, size_t ws_bytes,
                         void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FRCNN_REQUIRE(boxes && keep_idx && keep_count && n_max > 0 && max_keep > 0, "nms: bad arguments");
  FRCNN_REQUIRE(n_max <= 16384, "nms: n_max %d > 16384", n_max);
  const int nb = (n_max + 63) / 64;
  const size_t need = frcnn_nms_ws_bytes(n_max);
  if (!ws || ws_bytes < need) return fail(FRCNN_ERR_WS, "nms: workspace %zu < %zu bytes", ws_bytes, need);
  uint64_t* mask = static_cast<uint64_t*>(ws);
  uint64_t* diag_t = mask + (size_t)n_max * nb;
  if (keep_mask) {
    hipError_t e = frcnn::fill_bytes(keep_mask, 0, (size_t)n_max, stream);
    if (e != hipSuccess) return fail(FRCNN_ERR_LAUNCH, "nms: memset: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nb * (nb + 1) / 2), dim3(64), 0, stream, boxes, n_dev, n_max, nb,
                     nms_kernel_thresh(thresh), mask, diag_t);
  int rc = check_launch("nms_mask_kernel");
  if (rc != FRCNN_OK) return rc;
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64 * SCAN_WAVES), 0, stream, (const uint64_t*)mask,
                     (const uint64_t*)diag_t, n_dev, n_max, nb, std::min(max_keep, n_max), keep_idx, keep_mask, keep_count);
  return check_launch("nms_scan_kernel");
}

extern "C" int frcnn_make_rois(const float* sorted_boxes, const float* sorted_scores, const int64_t* keep_idx,
                               const int* keep_count, int max_keep, float* rois, float* roi_scores, void* stream_) {
  FRCNN_REQUIRE(sorted_boxes && sorted_scores && keep_idx && keep_count && rois && max_keep > 0,
                "make_rois: bad arguments");
  hipLaunchKernelGGL(make_rois_kernel, dim3((max_keep + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     sorted_boxes, sorted_scores, keep_idx, keep_count, max_keep, rois, roi_scores);
  return check_launch("make_rois_kernel");
}

// test hook: 0 = automatic (LDS kernel for num_rois <= 1024), 1 = always the general kernel
static int g_filter_variant = 0;
extern "C" int frcnn_filter_set_variant(int v) {
  g_filter_variant = v;
  return FRCNN_OK;
}
unsigned long long frcnn::boxes_settings_word() {
  return (unsigned long long)(unsigned)g_filter_variant | ((unsigned long long)(unsigned)g_nms_at_equal.load() << 32);
}

static size_t filter_ws_per_class(int num_rois) {
  const size_t nb = (size_t)(num_rois + 63) / 64;
  return align_up(align_up((size_t)num_rois * 16, 16) + (size_t)num_rois * nb * 8 + (size_t)num_rois * 8, 16);
}

extern "C" size_t frcnn_filter_per_class_ws_bytes(int num_rois, int num_classes) {
  if (num_rois <= 0 || num_classes <= 1) return 0;
  return filter_ws_per_class(num_rois) * (size_t)(num_classes - 1);
}

template <int E>
static int launch_filter(float* pred_boxes, const float* cls_prob, const int* roi_count, int num_rois, int num_classes,
                         float frame_w, float frame_h, float scale, float thresh, float nms_thresh, int max_dets,
                         int max_out, float* dets, int* det_count, int* det_roi, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FRCNN_REQUIRE(pred_boxes && cls_prob && dets && det_count && num_rois > 0 && num_classes > 1 && max_out > 0,
                "filter_per_class: bad arguments");
  FRCNN_REQUIRE(num_rois <= 8192, "filter_per_class: num_rois %d > 8192", num_rois);
  const size_t need = frcnn_filter_per_class_ws_bytes(num_rois, num_classes);
  if (!ws || ws_bytes < need) return fail(FRCNN_ERR_WS, "filter_per_class: workspace %zu < %zu bytes", ws_bytes, need);
  nms_thresh = nms_kernel_thresh(nms_thresh);
  if (E == 4) {
    // frame_width/scale - 1 evaluated in fp32 like the numpy float32 scalars of filter_predictions.py:77-91
    const float x_hi = frame_w / scale - 1.0f, y_hi = frame_h / scale - 1.0f;
    const int total_boxes = num_rois * num_classes;
    hipLaunchKernelGGL(clamp_pred_boxes_kernel, dim3(std::min((total_boxes + 255) / 256, 1024)), dim3(256), 0, stream,
                       pred_boxes, total_boxes, x_hi, y_hi);
    int rc = check_launch("clamp_pred_boxes_kernel");
    if (rc != FRCNN_OK) return rc;
  }  // LiDAR boxes are not clamped (filter_predictions.py:92-93)
  if (num_rois <= FILTER_SMALL_MAX && filter_small_lds(num_rois) <= (size_t)150 * 1024 && g_filter_variant != 1) {
    const size_t lds = filter_small_lds(num_rois);
    static std::atomic<size_t> configured_s{0};
    if (lds > configured_s.load()) {
      hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&filter_class_small_kernel<E>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e2 != hipSuccess) return fail(FRCNN_ERR_LAUNCH, "filter_per_class: set LDS size: %s", hipGetErrorString(e2));
      configured_s.store(lds);
    }
    hipLaunchKernelGGL(filter_class_small_kernel<E>, dim3(num_classes), dim3(FILTER_SMALL_THREADS), lds, stream,
                       pred_boxes, cls_prob, roi_count, num_rois, num_classes, thresh, nms_thresh, max_dets, max_out, dets,
                       det_count, det_roi);
    return check_launch("filter_class_small_kernel");
  }
  const int npad = next_pow2(std::max(num_rois, 2));
  const int nb = (num_rois + 63) / 64;
  const size_t lds = (size_t)npad * 8;
  static std::atomic<size_t> configured{0};
  if (lds > configured.load()) {
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&filter_class_kernel<E>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e2 != hipSuccess) return fail(FRCNN_ERR_LAUNCH, "filter_per_class: set LDS size: %s", hipGetErrorString(e2));
    configured.store(lds);
  }
  hipLaunchKernelGGL(filter_class_kernel<E>, dim3(num_classes), dim3(FILTER_THREADS), lds, stream, pred_boxes,
                     cls_prob, roi_count, num_rois, num_classes, thresh, nms_thresh, max_dets, max_out, npad, nb, dets,
                     det_count, det_roi, static_cast<unsigned char*>(ws), filter_ws_per_class(num_rois));
  return check_launch("filter_class_kernel");
}

extern "C" int frcnn_filter_per_class(float* pred_boxes, const float* cls_prob, const int* roi_count, int num_rois,
                                      int num_classes, float frame_w, float frame_h, float scale, float thresh,
                                      float nms_thresh, int max_dets, int max_out, float* dets, int* det_count,
                                      int* det_roi, void* ws, size_t ws_bytes, void* stream_) {
  return launch_filter<4>(pred_boxes, cls_prob, roi_count, num_rois, num_classes, frame_w, frame_h, scale, thresh,
                          nms_thresh, max_dets, max_out, dets, det_count, det_roi, ws, ws_bytes, stream_);
}

extern "C" int frcnn_filter_per_class_lidar(const float* pred_boxes, const float* cls_prob, const int* roi_count,
                                            int num_rois, int num_classes, float thresh, float nms_thresh,
                                            int max_dets, int max_out, float* dets, int* det_count, int* det_roi,
                                            void* ws, size_t ws_bytes, void* stream_) {
  return launch_filter<7>(const_cast<float*>(pred_boxes), cls_prob, roi_count, num_rois, num_classes, 0.f, 0.f, 1.f,
                          thresh, nms_thresh, max_dets, max_out, dets, det_count, det_roi, ws, ws_bytes, stream_);
}

extern "C" int frcnn_generate_anchors_3d(const float* base, int num_types, int height, int width, int feat_stride,
                                         float* anchors_3d, float* anchors_2d, void* stream_) {
  FRCNN_REQUIRE(base && anchors_3d && anchors_2d && num_types > 0 && height > 0 && width > 0 && feat_stride > 0,
                "generate_anchors_3d: bad arguments");
  const int total = height * width * num_types;
  hipLaunchKernelGGL(anchors3d_kernel, dim3(std::min((total + 255) / 256, 2048)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), base, num_types, height, width, feat_stride, anchors_3d,
                     anchors_2d);
  return check_launch("anchors3d_kernel");
}

extern "C" int frcnn_bbox_transform_inv(const float* boxes, int box_ld, const float* deltas, int n, int num_classes,
                                        float scale, float* out, void* stream_) {
  FRCNN_REQUIRE(boxes && deltas && out && n > 0 && num_classes > 0 && box_ld >= 4, "bbox_transform_inv: bad arguments");
  const int total = n * num_classes;
  hipLaunchKernelGGL(bbox_transform_inv_kernel, dim3(std::min((total + 255) / 256, 2048)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), boxes, box_ld, deltas, n, num_classes, scale, scale > 0.f ? 1 : 0,
                     out);
  return check_launch("bbox_transform_inv_kernel");
}

namespace {
// uncertainty_transform_inv / lidar_3d_uncertainty_transform_inv (lib/model/bbox_transform.py:107-130,132-169): a per-element
// uncertainty of the 7-element deltas [x,y,z,l,w,h,ry] mapped to box space and squared.  lidar == 0: the 4 BEV terms
// [x,y,l,w] -> out (n, 4K); lidar == 1: all 7 -> out (n, 7K).  Centre terms scale with the RoI's +1 sizes (z with the 3-D
// anchor's height), size terms are exp(u) - 1, the yaw term passes through.  The image form as written upstream omits the
// unsqueeze that makes the size factors per-box (see tests/golden/make_golden_uc_inv.py); per-box scaling is implemented.
__global__ __launch_bounds__(256) void uc_transform_inv_kernel(const float* __restrict__ rois, int roi_ld,
                                                              const float* __restrict__ anchors3d,
                                                              const float* __restrict__ uc, int n, int k, float scale,
                                                              int use_scale, int lidar, int is_var,
                                                              float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * k) return;
  const int i = t / k, c = t - i * k;
  const float* r = rois + (size_t)i * roi_ld;
  float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
  if (use_scale) { x1 = x1 / scale; y1 = y1 / scale; x2 = x2 / scale; y2 = y2 / scale; }
  const float len = x2 - x1 + 1.f, wid = y2 - y1 + 1.f;
  float u[7];
#pragma unroll
  for (int e = 0; e < 7; ++e) {
    const float v = uc[((size_t)i * k + c) * 7 + e];
    u[e] = is_var ? sqrtf(v) : v;          // a variance is turned into the standard deviation the formulas take
  }
  const float ux = u[0] * len, uy = u[1] * wid;
  const float ul = exp_f32(u[3]) - 1.f, uw = exp_f32(u[4]) - 1.f;
  if (!lidar) {
    float* o = out + ((size_t)i * k + c) * 4;
    o[0] = ux * ux; o[1] = uy * uy; o[2] = ul * ul; o[3] = uw * uw;
    return;
  }
  const float ht = anchors3d[(size_t)i * 7 + 5];
  const float uz = u[2] * ht, uh = exp_f32(u[5]) - 1.f, ur = u[6];
  float* o = out + ((size_t)i * k + c) * 7;
  o[0] = ux * ux; o[1] = uy * uy; o[2] = uz * uz; o[3] = ul * ul; o[4] = uw * uw; o[5] = uh * uh; o[6] = ur * ur;
}
}  // namespace

extern "C" int frcnn_uncertainty_transform_inv(const float* rois, int roi_ld, const float* anchors_3d, const float* uncertainty,
                                               int n, int num_classes, float scale, int lidar, int input_is_variance, float* out,
                                               void* stream_) {
  FRCNN_REQUIRE(rois && uncertainty && out && n > 0 && num_classes > 0 && roi_ld >= 4 && (!lidar || anchors_3d),
                "uncertainty_transform_inv: bad arguments");
  const int total = n * num_classes;
  hipLaunchKernelGGL(uc_transform_inv_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_), rois,
                     roi_ld, anchors_3d, uncertainty, n, num_classes, scale, scale > 0.f ? 1 : 0, lidar ? 1 : 0,
                     input_is_variance ? 1 : 0, out);
  return check_launch("uc_transform_inv_kernel");
}

extern "C" int frcnn_lidar_bbox_transform_inv(const float* rois, int roi_ld, const float* anchors_3d,
                                              const float* deltas, int n, int num_classes, float scale, float* out,
                                              void* stream_) {
  FRCNN_REQUIRE(rois && anchors_3d && deltas && out && n > 0 && num_classes > 0 && roi_ld >= 4,
                "lidar_bbox_transform_inv: bad arguments");
  const int total = n * num_classes;
  hipLaunchKernelGGL(lidar_bbox_transform_inv_kernel, dim3(std::min((total + 255) / 256, 2048)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), rois, roi_ld, anchors_3d, deltas, n, num_classes, scale,
                     scale > 0.f ? 1 : 0, out);
  return check_launch("lidar_bbox_transform_inv_kernel");
}

extern "C" int frcnn_fpn_level_map(const float* rois, int num_rois, int k_min, int k_max, float canonical_scale,
                                   float canonical_level, float eps, int* levels, void* stream_) {
  FRCNN_REQUIRE(rois && levels && num_rois > 0 && k_min <= k_max && canonical_scale > 0.f, "fpn_level_map: bad arguments");
  hipLaunchKernelGGL(fpn_level_map_kernel, dim3(std::min((num_rois + 255) / 256, 1024)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), rois, num_rois, k_min, k_max, canonical_scale, canonical_level, eps,
                     levels);
  return check_launch("fpn_level_map_kernel");
}

extern "C" int frcnn_clip_boxes(const float* boxes, int num_boxes, const float* info_host, float* out, void* stream_) {
  FRCNN_REQUIRE(boxes && out && info_host && num_boxes > 0, "clip_boxes: bad arguments");
  ClipInfo clip{info_host[0], info_host[1] - 1.0f, info_host[2], info_host[3] - 1.0f};
  hipLaunchKernelGGL(clip_boxes_kernel, dim3(std::min((num_boxes + 255) / 256, 2048)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), boxes, num_boxes, clip, out);
  return check_launch("clip_boxes_kernel");
}

