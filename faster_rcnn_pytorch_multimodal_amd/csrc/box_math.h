// Box arithmetic shared by the proposal / detection-tail kernels.  Every translation unit that includes
// this header is compiled with -ffp-contract=off: the reference evaluates these expressions as separate
// torch elementwise ops (one rounding per op), so no mul+add may fuse into an fma here.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace frcnn {

// exp() rounded once from a double evaluation: within 1 ulp of any faithful fp32 expf (the CPU's
// vectorised expf is not bit-reproducible across ISAs, so this is the best-defined target).
__device__ __forceinline__ float exp_f32(float v) { return (float)exp((double)v); }

// bbox_transform_inv for one (box, delta) pair — lib/model/bbox_transform.py:82-103.
// The codec is non-standard: centre deltas are scaled by the box DIAGONAL sqrt(w^2+h^2) (:84,:94-95),
// widths use the +1 convention (:82-83) and x2 = cx + 0.5*w' (no -1) (:99-103).
__device__ __forceinline__ void decode_box(float x1, float y1, float x2, float y2, float dx, float dy, float dw,
                                           float dh, float out[4]) {
  const float w = x2 - x1 + 1.0f;
  const float h = y2 - y1 + 1.0f;
  const float diag = sqrtf(w * w + h * h);
  const float cx = x1 + 0.5f * w;
  const float cy = y1 + 0.5f * h;
  const float pcx = dx * diag + cx;
  const float pcy = dy * diag + cy;
  const float pw = exp_f32(dw) * w;
  const float ph = exp_f32(dh) * h;
  out[0] = pcx - 0.5f * pw;
  out[1] = pcy - 0.5f * ph;
  out[2] = pcx + 0.5f * pw;
  out[3] = pcy + 0.5f * ph;
}

// lidar_3d_bbox_transform_inv for one (roi, 3-D anchor, delta) triple — lib/model/bbox_transform.py:185-222:
// length/width/centre from the axis-aligned RoI (+1 convention, centre = x1 + len/2), height and z from the
// 3-D anchor, centre deltas scaled by the RoI diagonal, heading = raw delta.
__device__ __forceinline__ void decode_box_lidar(float x1, float y1, float x2, float y2, const float* anchor3d,
                                                 const float d[7], float out[7]) {
  const float ln = x2 - x1 + 1.0f, wd = y2 - y1 + 1.0f, ht = anchor3d[5];
  const float cx = x1 + ln / 2.0f, cy = y1 + wd / 2.0f, cz = anchor3d[2];
  const float diag = sqrtf(ln * ln + wd * wd);
  out[0] = d[0] * diag + cx;
  out[1] = d[1] * diag + cy;
  out[2] = d[2] * ht + cz;
  out[3] = exp_f32(d[3]) * ln;
  out[4] = exp_f32(d[4]) * wd;
  out[5] = exp_f32(d[5]) * ht;
  out[6] = d[6];
}

// torch.clamp(v, lo, hi) = min(max(v, lo), hi), and like torch's it hands a NaN through (fminf / fmaxf would return the
// bound: a NaN coordinate of a diverged regression would silently become a frame edge).  clamp_min / clamp_max likewise.
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float clamp_minf(float v, float lo) { return v < lo ? lo : v; }
__device__ __forceinline__ float clamp_maxf(float v, float hi) { return v > hi ? hi : v; }

// What happens when IoU == threshold exactly is a RUN-TIME choice (frcnn_nms_set_suppress_at_equal, csrc/boxes.hip; DESIGN.md
// section 1): torchvision 0.4.0 (req.txt:283, not vendored) suppresses on `iou >= threshold` in its CPU kernel and on
// `iou > threshold` in its CUDA kernel.  The default follows the CPU kernel - the path the reference is compared against.
// The kernels below always test `iou > t`; for the inclusive form the entry points hand them t = nextafterf(threshold, -inf):
// no float lies strictly between the two, so `iou > t` IS `iou >= threshold`, bit for bit, NaN included (both false).

// IoU test of torchvision.ops.nms: areas (x2-x1)*(y2-y1) without +1, suppress when iou > thresh (see above).
__device__ __forceinline__ bool iou_gt(const float* a, const float* b, float thresh) {
  const float xx1 = fmaxf(a[0], b[0]), yy1 = fmaxf(a[1], b[1]);
  const float xx2 = fminf(a[2], b[2]), yy2 = fminf(a[3], b[3]);
  const float w = fmaxf(xx2 - xx1, 0.f), h = fmaxf(yy2 - yy1, 0.f);
  const float inter = w * h;
  const float sa = (a[2] - a[0]) * (a[3] - a[1]);
  const float sb = (b[2] - b[0]) * (b[3] - b[1]);
  const float iou = inter / (sa + sb - inter);
  return iou > thresh;
}

// The same predicate, bit for bit, with the IEEE division (~25 instructions) taken only near the threshold: inter against
// thresh * union decides every other pair with a margin (1e-6 relative, against 2^-24 rounding in each of the two
// products and half an ulp in the quotient); degenerate boxes (union <= 0, NaN) always take the division.  Pays where the
// IoU count per lane is small and instruction-bound (the per-class filter); the 64-deep serial loop of nms_mask_kernel
// is faster with the plain form.
__device__ __forceinline__ bool iou_gt_lazy_div(const float* a, const float* b, float thresh) {
  const float xx1 = fmaxf(a[0], b[0]), yy1 = fmaxf(a[1], b[1]);
  const float xx2 = fminf(a[2], b[2]), yy2 = fminf(a[3], b[3]);
  const float w = fmaxf(xx2 - xx1, 0.f), h = fmaxf(yy2 - yy1, 0.f);
  const float inter = w * h;
  const float sa = (a[2] - a[0]) * (a[3] - a[1]);
  const float sb = (b[2] - b[0]) * (b[3] - b[1]);
  const float uni = sa + sb - inter;
  const float tu = thresh * uni;
  const bool sure_yes = inter > tu * 1.000001f, sure_no = inter < tu * 0.999999f;
  bool hit = sure_yes;
  if (!(uni > 0.f) || !(sure_yes || sure_no)) {   // rarely taken (skipped when no lane of the wave needs it)
    const float iou = inter / uni;
    hit = iou > thresh;
  }
  return hit;
}

// fp32 -> u32 key whose ASCENDING order is DESCENDING score (-0 is folded into +0 first).  NaN - with either sign bit - gets
// the smallest key: torch.sort(descending=True) (lib/layer_utils/proposal_layer.py:39) ranks NaN above +Inf, ties by index.
__device__ __forceinline__ uint32_t desc_key(float s) {
  if (s != s) return 0u;
  if (s == 0.f) s = 0.f;
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~u;
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
  const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l);
  const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}

// In-LDS bitonic sort (ascending) of npad (power of two) u64 keys by the whole workgroup.
__device__ __forceinline__ void block_bitonic_sort(uint64_t* keys, int npad) {
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npad / 2; i += blockDim.x) {
        // pair (a, a+j) where a has bit j clear
        const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int b = a | j;
        const uint64_t ka = keys[a], kb = keys[b];
        const bool up = (a & k) == 0;
        if ((ka > kb) == up) {
          keys[a] = kb;
          keys[b] = ka;
        }
      }
      __syncthreads();
    }
  }
}

// Greedy scan over a precomputed suppression bit-matrix by ONE wave (lane = threadIdx.x & 63).
// mask[i*nb + w] bit b set  <=>  box (w*64+b) has IoU > thresh with box i and (w*64+b) > i.
// Boxes are in descending-score order.  Writes survivors' positions to keep_idx (first max_keep of them)
// and optional per-box bytes to keep_mask (pre-zeroed by the caller).  nb <= 256.
// MaskPtr: `const uint64_t*` (global) or an address_space(3) pointer when the matrix lives in LDS - a generic pointer
// would turn every row fetch into a flat load (40 of the 75 us of the per-class filter kernel were that).
template <typename MaskPtr, typename KeepPtr>
__device__ __forceinline__ int wave_nms_scan(MaskPtr mask, int nb, int n, int max_keep, KeepPtr keep_idx,
                                             uint8_t* keep_mask) {
  const int lane = threadIdx.x & 63;
  uint64_t rm0 = 0, rm1 = 0, rm2 = 0, rm3 = 0;  // removed bits of words lane, lane+64, lane+128, lane+192
  int count = 0;
  // The chain per 64-box chunk is: diagonal word of the 64 rows -> in-register greedy resolve -> rows of the kept boxes
  // OR-ed into the removed set -> next chunk.  Two memory latencies per chunk would sit on that chain; the diagonal
  // word of the NEXT chunk does not depend on anything computed here, so it is fetched one chunk ahead, and the rows
  // of up to four kept boxes are requested together before any of them is consumed.
  uint64_t d_next = (lane < n) ? mask[(size_t)lane * nb] : 0ull;
  for (int c = 0; c < nb && c * 64 < n && count < max_keep; ++c) {
    const int i = c * 64 + lane;
    const uint64_t d = d_next;
    {
      const int in = i + 64;
      d_next = (c + 1 < nb && in < n) ? mask[(size_t)in * nb + (c + 1)] : 0ull;
    }
    const int slot = c >> 6;
    const uint64_t mine = slot == 0 ? rm0 : slot == 1 ? rm1 : slot == 2 ? rm2 : rm3;
    const uint64_t rw = readlane_u64(mine, c & 63);
    const int live = n - c * 64;
    const uint64_t valid = live >= 64 ? ~0ull : ((1ull << live) - 1ull);
    uint64_t alive = ~rw & valid;
    uint64_t kept = 0;
    int nk = 0;
    while (alive != 0 && count + nk < max_keep) {
      const int b = __builtin_ctzll(alive);
      kept |= 1ull << b;
      ++nk;
      const uint64_t db = readlane_u64(d, b);
      alive &= ~(db | (1ull << b));
    }
    if ((kept >> lane) & 1ull) {
      const int pos = count + __builtin_popcountll(kept & ((1ull << lane) - 1ull));
      keep_idx[pos] = i;
      if (keep_mask) keep_mask[i] = 1;
    }
    count += nk;
    if (count >= max_keep) break;
    uint64_t todo = kept;
    while (todo != 0) {
      MaskPtr rows[4];
      int nr = 0;
      for (; nr < 4 && todo != 0; ++nr) {
        rows[nr] = mask + (size_t)(c * 64 + __builtin_ctzll(todo)) * nb;
        todo &= todo - 1ull;
      }
      for (int q = nr; q < 4; ++q) rows[q] = rows[0];      // duplicates: OR is idempotent
      // unconditional loads at a clamped word index (issued back to back), masked afterwards; slots beyond the row
      // length are skipped by a wave-uniform branch
      uint64_t v[4][4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        if (s4 * 64 < nb) {
          const int w = lane + 64 * s4;
          const int wc = min(w, nb - 1);
          const bool ok = w > c && w < nb;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q][s4] = rows[q][wc];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q][s4] = ok ? v[q][s4] : 0ull;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q][s4] = 0ull;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        rm0 |= v[q][0];
        rm1 |= v[q][1];
        rm2 |= v[q][2];
        rm3 |= v[q][3];
      }
    }
  }
  return count;
}

}  // namespace frcnn
