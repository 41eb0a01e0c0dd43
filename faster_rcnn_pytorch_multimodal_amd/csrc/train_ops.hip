// Training-path kernels other than the convolutions: activation backward, FPN bilinear upsample-add
// (forward/backward), RoIAlign backward, the RPN / detection losses with their gradients.
// HBM/latency-bound elementwise and gather work: 16-byte accesses along the channel dimension, wave-uniform
// sampling weights, fixed-order reductions (the only non-deterministic pieces are the float-atomic scatters: RoIAlign backward, patch gradients).
// Built with -ffp-contract=off: the reference evaluates these expressions as separate torch ops.
#include "common.h"
#include "box_math.h"
#include "rng.h"

using namespace frcnn;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// y = act(conv*scale + shift + res)  ->  dz = relu ? (y > 0 ? dy : 0) : dy ; d_res = dz ; d_conv = dz*scale
// (autograd of F.relu / eval-mode batch_norm / the residual add in lib/nets/resnet.py:98-127).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                     const float* __restrict__ scale, int relu, size_t n4, int K4,
                                                     float* __restrict__ d_conv, float* __restrict__ d_res) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
    if (relu) {
      const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
      for (int e = 0; e < 4; ++e) g[e] = v[e] > 0.f ? g[e] : 0.f;
    }
    if (d_res) reinterpret_cast<f32x4*>(d_res)[i] = g;
    if (scale) {
      const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[i % K4];
      for (int e = 0; e < 4; ++e) g[e] = g[e] * sc[e];
    }
    reinterpret_cast<f32x4*>(d_conv)[i] = g;
  }
}

// ------------------------------------------------------------------------------------------------
// F.interpolate(x, size=(H,W), mode='bilinear', align_corners=False) + y  — lib/nets/fpn.py:42-45.
// Source coordinate = (dst + 0.5) * (in/out) - 0.5, clamped at 0; the four taps are combined as
// l0y*(l0x*v00 + l1x*v01) + l1y*(l0x*v10 + l1x*v11) like ATen's upsample_bilinear2d.
// ------------------------------------------------------------------------------------------------
struct Tap {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Tap bilinear_tap(int dst, float ratio, int in_size) {
  float src = ((float)dst + 0.5f) * ratio - 0.5f;
  if (src < 0.f) src = 0.f;
  Tap t;
  t.i0 = (int)src;
  t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}

__global__ __launch_bounds__(256) void upsample_add_fwd_kernel(const float* __restrict__ x, const float* __restrict__ lat,
                                                              int N, int h, int w, int H, int W, int C4,
                                                              float* __restrict__ out) {
  const float ry = (float)h / (float)H, rx = (float)w / (float)W;
  const size_t total = (size_t)N * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t t = i / C4;
    const int X = (int)(t % W);
    t /= W;
    const int Y = (int)(t % H);
    const int n = (int)(t / H);
    const Tap ty = bilinear_tap(Y, ry, h), tx = bilinear_tap(X, rx, w);
    const f32x4* xb = reinterpret_cast<const f32x4*>(x) + (size_t)n * h * w * C4 + c4;
    const f32x4 v00 = xb[((size_t)ty.i0 * w + tx.i0) * C4], v01 = xb[((size_t)ty.i0 * w + tx.i1) * C4];
    const f32x4 v10 = xb[((size_t)ty.i1 * w + tx.i0) * C4], v11 = xb[((size_t)ty.i1 * w + tx.i1) * C4];
    const f32x4 l = reinterpret_cast<const f32x4*>(lat)[i];
    f32x4 o;
    for (int e = 0; e < 4; ++e)
      o[e] = (ty.l0 * (tx.l0 * v00[e] + tx.l1 * v01[e]) + ty.l1 * (tx.l0 * v10[e] + tx.l1 * v11[e])) + l[e];
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

// Gradient w.r.t. the low-resolution input: each SOURCE pixel gathers from the destination pixels whose taps
// touch it (bounded search window, exact tap test), in fixed (Y, X) order -> deterministic, no atomics.
// (The gradient w.r.t. the lateral input is dout itself.)
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dout, int N, int h, int w, int H,
                                                          int W, int C4, float* __restrict__ dx) {
  const float ry = (float)h / (float)H, rx = (float)w / (float)W;
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  const size_t total = (size_t)N * h * w * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t t = i / C4;
    const int xs = (int)(t % w);
    t /= w;
    const int ys = (int)(t % h);
    const int n = (int)(t / h);
    // destination rows whose source coordinate can fall in (ys-1, ys+1), widened by one row each side
    const int Y0 = max(0, (int)floorf(((float)ys - 1.f + 0.5f) * sy - 0.5f) - 1);
    const int Y1 = min(H - 1, (int)ceilf(((float)ys + 1.f + 0.5f) * sy - 0.5f) + 1);
    const int X0 = max(0, (int)floorf(((float)xs - 1.f + 0.5f) * sx - 0.5f) - 1);
    const int X1 = min(W - 1, (int)ceilf(((float)xs + 1.f + 0.5f) * sx - 0.5f) + 1);
    const f32x4* db = reinterpret_cast<const f32x4*>(dout) + (size_t)n * H * W * C4 + c4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int Y = Y0; Y <= Y1; ++Y) {
      const Tap ty = bilinear_tap(Y, ry, h);
      float wy = 0.f;
      if (ty.i0 == ys) wy += ty.l0;
      if (ty.i1 == ys) wy += ty.l1;
      if (wy == 0.f) continue;
      for (int X = X0; X <= X1; ++X) {
        const Tap tx = bilinear_tap(X, rx, w);
        float wx = 0.f;
        if (tx.i0 == xs) wx += tx.l0;
        if (tx.i1 == xs) wx += tx.l1;
        if (wx == 0.f) continue;
        const f32x4 g = db[((size_t)Y * W + X) * C4];
        const float wgt = wy * wx;
        for (int e = 0; e < 4; ++e) acc[e] += wgt * g[e];
      }
    }
    reinterpret_cast<f32x4*>(dx)[i] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// RoIAlign backward (torchvision roi_align autograd, aligned=False): every sample scatters
// dout/count * bilinear weight onto its four pixels with float atomics.  dfeat must be zero-filled.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void roi_align_bwd_nhwc(const float* __restrict__ dout, int H, int W, int C,
                                                         const float* __restrict__ rois,
                                                         const int* __restrict__ roi_count, int num_rois, int P,
                                                         float spatial_scale, int sampling_ratio,
                                                         const int* __restrict__ level_of_roi, int level,
                                                         float* __restrict__ dfeat) {
  const int live = roi_count ? min(*roi_count, num_rois) : num_rois;
  const size_t total = (size_t)num_rois * P * P * C;
  for (size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x; item < total;
       item += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(item % C);
    size_t bin = item / C;
    const int pw = (int)(bin % P);
    bin /= P;
    const int ph = (int)(bin % P);
    const int r = (int)(bin / P);
    if (r >= live) continue;
    if (level_of_roi && level_of_roi[r] != level) continue;
    const float* roi = rois + (size_t)r * 5;
    const int b = (int)roi[0];
    const float roi_start_w = roi[1] * spatial_scale, roi_start_h = roi[2] * spatial_scale;
    const float roi_end_w = roi[3] * spatial_scale, roi_end_h = roi[4] * spatial_scale;
    const float roi_width = fmaxf(roi_end_w - roi_start_w, 1.0f), roi_height = fmaxf(roi_end_h - roi_start_h, 1.0f);
    const float bin_size_h = roi_height / (float)P, bin_size_w = roi_width / (float)P;
    const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_height / (float)P);
    const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_width / (float)P);
    const float g = dout[item] / (float)(grid_h * grid_w);
    float* fb = dfeat + (size_t)b * H * W * C + c;
    for (int iy = 0; iy < grid_h; ++iy) {
      const float y0 = roi_start_h + ph * bin_size_h + ((float)iy + .5f) * bin_size_h / (float)grid_h;
      for (int ix = 0; ix < grid_w; ++ix) {
        float x = roi_start_w + pw * bin_size_w + ((float)ix + .5f) * bin_size_w / (float)grid_w;
        float y = y0;
        if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
        if (y <= 0.f) y = 0.f;
        if (x <= 0.f) x = 0.f;
        int y_low = (int)y, x_low = (int)x, y_high, x_high;
        if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
        if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
        const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
        atomicAdd(fb + ((size_t)y_low * W + x_low) * C, g * (hy * hx));
        atomicAdd(fb + ((size_t)y_low * W + x_high) * C, g * (hy * lx));
        atomicAdd(fb + ((size_t)y_high * W + x_low) * C, g * (ly * hx));
        atomicAdd(fb + ((size_t)y_high * W + x_high) * C, g * (ly * lx));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fc7 = x.mean(3).mean(2) of the layer4 output (_head_to_tail of the non-FPN detector) and its backward.
// x (R,P,P,C) NHWC: mean over W inside each row, then over the P row means, like the reference's two calls.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spatial_mean_fwd_kernel(const float* __restrict__ x, int R, int P, int C4,
                                                              float* __restrict__ out) {
  const size_t total = (size_t)R * C4;
  const float inv = (float)P;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const size_t r = i / C4;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x) + r * P * P * C4 + c4;
    f32x4 tot = {0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < P; ++h) {
      f32x4 row = {0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < P; ++w) {
        const f32x4 v = xr[(size_t)(h * P + w) * C4];
        for (int e = 0; e < 4; ++e) row[e] += v[e];
      }
      for (int e = 0; e < 4; ++e) tot[e] += row[e] / inv;
    }
    for (int e = 0; e < 4; ++e) tot[e] = tot[e] / inv;
    reinterpret_cast<f32x4*>(out)[i] = tot;
  }
}

__global__ __launch_bounds__(256) void spatial_mean_bwd_kernel(const float* __restrict__ dout, int R, int P, int C4,
                                                              float* __restrict__ dx) {
  const size_t total = (size_t)R * P * P * C4;
  const float inv = 1.0f / ((float)P * (float)P);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const size_t r = i / ((size_t)P * P * C4);
    f32x4 g = reinterpret_cast<const f32x4*>(dout)[r * C4 + c4];
    for (int e = 0; e < 4; ++e) g[e] = g[e] * inv;
    reinterpret_cast<f32x4*>(dx)[i] = g;
  }
}

// ------------------------------------------------------------------------------------------------
// Fixed-order block reduction helper: 256 threads -> lane 0 of wave 0 holds the sum.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* smem4) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) smem4[wave] = v;
  __syncthreads();
  const float s = ((smem4[0] + smem4[1]) + smem4[2]) + smem4[3];
  __syncthreads();
  return s;
}

// ------------------------------------------------------------------------------------------------
// RPN losses on the fused head output rpn (HW, ld) = [A bg logits | A fg logits | 4A deltas | pad]:
//   cross_entropy over anchors with label != -1 (mean)            — ancestor network.py _add_losses
//   smooth_l1_loss('RPN', pred, targets, inside, outside, dim=[1,2,3])  — lib/utils/loss_utils.py:39-101
// Pass 1 (LOSS_BLOCKS workgroups): partial sums of CE, labelled-anchor count and box loss.
// Pass 2 (one workgroup): totals in block order.  Pass 3: gradient into drpn (same layout as rpn).
// ------------------------------------------------------------------------------------------------
constexpr int LOSS_BLOCKS = 256;

__device__ __forceinline__ float huber1(float diff) {  // loss_utils.py:28-37 with delta = 1
  const float a = fabsf(diff);
  return a < 1.f ? 0.5f * (diff * diff) : (a - 0.5f);
}
__device__ __forceinline__ float huber1_grad(float diff) { return fabsf(diff) < 1.f ? diff : (diff > 0.f ? 1.f : -1.f); }

__global__ __launch_bounds__(256) void rpn_loss_partial_kernel(const float* __restrict__ rpn, int ld, int A, int total,
                                                              const float* __restrict__ labels,
                                                              const float* __restrict__ targets,
                                                              const float* __restrict__ inside,
                                                              const float* __restrict__ outside,
                                                              float* __restrict__ partial) {
  __shared__ float red[4];
  float ce = 0.f, cnt = 0.f, box = 0.f;
  const int per = (total + LOSS_BLOCKS - 1) / LOSS_BLOCKS;
  const int lo = blockIdx.x * per, hi = min(lo + per, total);
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const int a = i % A, pix = i / A;
    const float* row = rpn + (size_t)pix * ld;
    const float lab = labels[i];
    if (lab >= 0.f) {
      const float bg = row[a], fg = row[A + a];
      const float m = fmaxf(bg, fg);
      const float lse = m + (float)log((double)exp_f32(bg - m) + (double)exp_f32(fg - m));
      ce += lse - (lab > 0.5f ? fg : bg);
      cnt += 1.f;
    }
    const float4 t = reinterpret_cast<const float4*>(targets)[i];
    const float4 iw = reinterpret_cast<const float4*>(inside)[i];
    const float4 ow = reinterpret_cast<const float4*>(outside)[i];
    const float* p = row + 2 * A + 4 * a;
    box += ow.x * huber1(p[0] * iw.x - t.x * iw.x) + ow.y * huber1(p[1] * iw.y - t.y * iw.y) +
           ow.z * huber1(p[2] * iw.z - t.z * iw.z) + ow.w * huber1(p[3] * iw.w - t.w * iw.w);
  }
  ce = block_sum_256(ce, red);
  cnt = block_sum_256(cnt, red);
  box = block_sum_256(box, red);
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 3 + 0] = ce;
    partial[blockIdx.x * 3 + 1] = cnt;
    partial[blockIdx.x * 3 + 2] = box;
  }
}

// losses[0] = rpn cross entropy (mean), losses[1] = rpn box loss, losses[2] = labelled-anchor count
__global__ __launch_bounds__(64) void rpn_loss_final_kernel(const float* __restrict__ partial, float* __restrict__ losses) {
  if (threadIdx.x != 0) return;
  float ce = 0.f, cnt = 0.f, box = 0.f;
  for (int b = 0; b < LOSS_BLOCKS; ++b) {
    ce += partial[b * 3 + 0];
    cnt += partial[b * 3 + 1];
    box += partial[b * 3 + 2];
  }
  losses[0] = cnt > 0.f ? ce / cnt : 0.f;
  losses[1] = box;
  losses[2] = cnt;
}

__global__ __launch_bounds__(256) void rpn_loss_grad_kernel(const float* __restrict__ rpn, int ld, int A, int total,
                                                           const float* __restrict__ labels,
                                                           const float* __restrict__ targets,
                                                           const float* __restrict__ inside,
                                                           const float* __restrict__ outside,
                                                           const float* __restrict__ losses, float g_ce, float g_box,
                                                           float* __restrict__ drpn) {
  const float cnt = losses[2];
  const float inv = cnt > 0.f ? g_ce / cnt : 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int a = i % A, pix = i / A;
    const float* row = rpn + (size_t)pix * ld;
    float* drow = drpn + (size_t)pix * ld;
    const float lab = labels[i];
    float dbg = 0.f, dfg = 0.f;
    if (lab >= 0.f) {
      const float bg = row[a], fg = row[A + a];
      const float m = fmaxf(bg, fg);
      const float eb = exp_f32(bg - m), ef = exp_f32(fg - m);
      const float pf = ef / (eb + ef), pb = eb / (eb + ef);
      dbg = (pb - (lab > 0.5f ? 0.f : 1.f)) * inv;
      dfg = (pf - (lab > 0.5f ? 1.f : 0.f)) * inv;
    }
    drow[a] = dbg;
    drow[A + a] = dfg;
    const float4 t = reinterpret_cast<const float4*>(targets)[i];
    const float4 iw = reinterpret_cast<const float4*>(inside)[i];
    const float4 ow = reinterpret_cast<const float4*>(outside)[i];
    const float* p = row + 2 * A + 4 * a;
    float* dp = drow + 2 * A + 4 * a;
    dp[0] = g_box * ow.x * huber1_grad(p[0] * iw.x - t.x * iw.x) * iw.x;
    dp[1] = g_box * ow.y * huber1_grad(p[1] * iw.y - t.y * iw.y) * iw.y;
    dp[2] = g_box * ow.z * huber1_grad(p[2] * iw.z - t.z * iw.z) * iw.z;
    dp[3] = g_box * ow.w * huber1_grad(p[3] * iw.w - t.w * iw.w) * iw.w;
  }
}

// ------------------------------------------------------------------------------------------------
// Detection losses on R sampled RoIs (R <= 4096): F.cross_entropy(cls_score, labels) (mean) and
// smooth_l1_loss('DET', bbox_pred, targets, inside, outside) = mean over RoIs of the row sums.
// One workgroup: fixed-order reductions, then the gradients.  losses[0] = CE, losses[1] = box loss.
// ------------------------------------------------------------------------------------------------
// LiDAR 'DET' stage (loss_utils.py:61-77): the element `sin_elem` of every E-group (the yaw) goes through sin()
// before the Huber term (cfg.LIDAR.EN_RY_SIN), and each element is scaled by cfg.LIDAR.REG_LOSS_WEIGHT.
struct DetLossOpt {
  float w[8];
  int sin_elem;   // -1: plain smooth-L1 on every element (image detector)
};

__global__ __launch_bounds__(256) void det_loss_kernel(const float* __restrict__ cls_score, const float* __restrict__ labels,
                                                      int R, int K, const float* __restrict__ bbox_pred,
                                                      const float* __restrict__ targets, const float* __restrict__ inside,
                                                      const float* __restrict__ outside, int E, DetLossOpt opt,
                                                      const float* __restrict__ bbox_var, float* __restrict__ dvar,
                                                      float g_ce, float g_box,
                                                      float* __restrict__ losses, float* __restrict__ dcls,
                                                      float* __restrict__ dbox) {
  __shared__ float red[4];
  float ce = 0.f, box = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    const float* s = cls_score + (size_t)r * K;
    float m = s[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, s[k]);
    double sum = 0.0;
    for (int k = 0; k < K; ++k) sum += (double)exp_f32(s[k] - m);
    const int lab = (int)labels[r];
    ce += (m + (float)log(sum)) - s[lab];
    if (dcls)
      for (int k = 0; k < K; ++k)
        dcls[(size_t)r * K + k] = ((float)((double)exp_f32(s[k] - m) / sum) - (k == lab ? 1.f : 0.f)) * (g_ce / (float)R);
    const int cols = E * K;
    float rowsum = 0.f;
    for (int q = 0; q < cols; ++q) {
      const size_t o = (size_t)r * cols + q;
      float diff = bbox_pred[o] * inside[o] - targets[o] * inside[o];
      const int e = q % E;
      float chain = inside[o];
      if (e == opt.sin_elem) {
        chain = chain * cosf(diff);
        diff = sinf(diff);
      }
      const float we = e < 8 ? opt.w[e] : 1.f;
      if (bbox_var) {
        // aleatoric attenuation (loss_utils.py:82-85): (0.5 * loss * exp(-s) + 0.5 * s) * inside, s = predicted log-variance
        const float sv = bbox_var[o], ev = expf(-sv);
        const float l = huber1(diff) * we;
        rowsum += outside[o] * ((0.5f * l * ev + 0.5f * sv) * inside[o]);
        if (dbox) dbox[o] = g_box / (float)R * outside[o] * inside[o] * (0.5f * ev) * (huber1_grad(diff) * we) * chain;
        if (dvar) dvar[o] = g_box / (float)R * outside[o] * inside[o] * (0.5f - 0.5f * l * ev);
        continue;
      }
      rowsum += outside[o] * (huber1(diff) * we);
      if (dbox) dbox[o] = g_box / (float)R * outside[o] * (huber1_grad(diff) * we) * chain;
    }
    box += rowsum;
  }
  ce = block_sum_256(ce, red);
  box = block_sum_256(box, red);
  if (threadIdx.x == 0) {
    losses[0] = ce / (float)R;
    losses[1] = box / (float)R;
  }
}

unsigned grid_for(size_t items, unsigned cap = 8192) { return (unsigned)std::min<size_t>((items + 255) / 256, cap); }

}  // namespace

extern "C" int frcnn_act_bwd(const float* dy, const float* y, const float* scale, int relu, int64_t rows, int k,
                             float* d_conv, float* d_res, void* stream_) {
  FRCNN_REQUIRE(dy && d_conv && rows > 0 && k > 0 && k % 4 == 0 && (!relu || y), "act_bwd: bad arguments (k%%4==0)");
  const size_t n4 = (size_t)rows * (k / 4);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n4)), dim3(256), 0, static_cast<hipStream_t>(stream_), dy, y, scale,
                     relu, n4, k / 4, d_conv, d_res);
  return check_launch("act_bwd_kernel");
}

extern "C" int frcnn_spatial_mean_fwd(const float* x, float* out, int rows, int pooled, int c, void* stream_) {
  FRCNN_REQUIRE(x && out && rows > 0 && pooled > 0 && c > 0 && c % 4 == 0, "spatial_mean_fwd: bad arguments (c%%4==0)");
  hipLaunchKernelGGL(spatial_mean_fwd_kernel, dim3(grid_for((size_t)rows * (c / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), x, rows, pooled, c / 4, out);
  return check_launch("spatial_mean_fwd_kernel");
}

extern "C" int frcnn_spatial_mean_bwd(const float* dout, float* dx, int rows, int pooled, int c, void* stream_) {
  FRCNN_REQUIRE(dout && dx && rows > 0 && pooled > 0 && c > 0 && c % 4 == 0, "spatial_mean_bwd: bad arguments (c%%4==0)");
  hipLaunchKernelGGL(spatial_mean_bwd_kernel, dim3(grid_for((size_t)rows * pooled * pooled * (c / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), dout, rows, pooled, c / 4, dx);
  return check_launch("spatial_mean_bwd_kernel");
}

extern "C" int frcnn_upsample_bilinear_add_fwd(const float* x, const float* lateral, float* out, int n, int h, int w,
                                               int out_h, int out_w, int c, void* stream_) {
  FRCNN_REQUIRE(x && lateral && out && n > 0 && h > 0 && w > 0 && out_h > 0 && out_w > 0 && c > 0 && c % 4 == 0,
                "upsample_bilinear_add_fwd: bad arguments (c%%4==0)");
  const size_t total = (size_t)n * out_h * out_w * (c / 4);
  hipLaunchKernelGGL(upsample_add_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream_), x,
                     lateral, n, h, w, out_h, out_w, c / 4, out);
  return check_launch("upsample_add_fwd_kernel");
}

extern "C" int frcnn_upsample_bilinear_bwd(const float* dout, float* dx, int n, int h, int w, int out_h, int out_w,
                                           int c, void* stream_) {
  FRCNN_REQUIRE(dout && dx && n > 0 && h > 0 && w > 0 && out_h > 0 && out_w > 0 && c > 0 && c % 4 == 0,
                "upsample_bilinear_bwd: bad arguments (c%%4==0)");
  const size_t total = (size_t)n * h * w * (c / 4);
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream_), dout, n,
                     h, w, out_h, out_w, c / 4, dx);
  return check_launch("upsample_bwd_kernel");
}

extern "C" int frcnn_roi_align_bwd(const float* dout, int h, int w, int c, const float* rois, const int* roi_count,
                                   int num_rois, int pooled, float spatial_scale, int sampling_ratio,
                                   const int* level_of_roi, int level, float* dfeat, void* stream_) {
  FRCNN_REQUIRE(dout && rois && dfeat && h > 0 && w > 0 && c > 0 && num_rois > 0 && pooled > 0,
                "roi_align_bwd: bad arguments");
  const size_t total = (size_t)num_rois * pooled * pooled * c;
  hipLaunchKernelGGL(roi_align_bwd_nhwc, dim3(grid_for(total, 1u << 20)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     dout, h, w, c, rois, roi_count, num_rois, pooled, spatial_scale, sampling_ratio, level_of_roi, level,
                     dfeat);
  return check_launch("roi_align_bwd_nhwc");
}

extern "C" size_t frcnn_rpn_loss_ws_bytes(void) { return (size_t)LOSS_BLOCKS * 3 * sizeof(float); }

extern "C" int frcnn_rpn_loss(const float* rpn, int ld, int num_anchors, int hw, const float* labels,
                              const float* targets, const float* inside, const float* outside, float grad_ce,
                              float grad_box, float* losses, float* drpn, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FRCNN_REQUIRE(rpn && labels && targets && inside && outside && losses && hw > 0 && num_anchors > 0 &&
                    ld >= 6 * num_anchors,
                "rpn_loss: bad arguments (ld >= 6A)");
  if (!ws || ws_bytes < frcnn_rpn_loss_ws_bytes())
    return fail(FRCNN_ERR_WS, "rpn_loss: workspace %zu < %zu bytes", ws_bytes, frcnn_rpn_loss_ws_bytes());
  const int total = hw * num_anchors;
  float* partial = static_cast<float*>(ws);
  hipLaunchKernelGGL(rpn_loss_partial_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, stream, rpn, ld, num_anchors, total, labels,
                     targets, inside, outside, partial);
  int rc = check_launch("rpn_loss_partial_kernel");
  if (rc != FRCNN_OK) return rc;
  hipLaunchKernelGGL(rpn_loss_final_kernel, dim3(1), dim3(64), 0, stream, partial, losses);
  rc = check_launch("rpn_loss_final_kernel");
  if (rc != FRCNN_OK || !drpn) return rc;
  if (ld > 6 * num_anchors) {  // padding columns of the fused head carry no gradient
    hipError_t e = fill_bytes(drpn, 0, (size_t)hw * ld * sizeof(float), stream);
    if (e != hipSuccess) return fail(FRCNN_ERR_LAUNCH, "rpn_loss: memset: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(rpn_loss_grad_kernel, dim3(grid_for((size_t)total, 4096)), dim3(256), 0, stream, rpn, ld, num_anchors,
                     total, labels, targets, inside, outside, losses, grad_ce, grad_box, drpn);
  return check_launch("rpn_loss_grad_kernel");
}

namespace {
int launch_det_loss(const float* cls_score, const float* labels, int num_rois, int num_classes, const float* bbox_pred,
                    const float* targets, const float* inside, const float* outside, int bbox_elem, DetLossOpt opt,
                    float grad_ce, float grad_box, float* losses, float* dcls, float* dbox, void* stream_,
                    const float* bbox_var = nullptr, float* dvar = nullptr) {
  FRCNN_REQUIRE(cls_score && labels && bbox_pred && targets && inside && outside && losses && num_rois > 0 &&
                    num_rois <= 4096 && num_classes > 1 && bbox_elem > 0,
                "det_loss: bad arguments (num_rois <= 4096)");
  hipLaunchKernelGGL(det_loss_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream_), cls_score, labels, num_rois,
                     num_classes, bbox_pred, targets, inside, outside, bbox_elem, opt, bbox_var, dvar, grad_ce, grad_box,
                     losses, dcls, dbox);
  return check_launch("det_loss_kernel");
}
}  // namespace

extern "C" int frcnn_det_loss(const float* cls_score, const float* labels, int num_rois, int num_classes,
                              const float* bbox_pred, const float* targets, const float* inside, const float* outside,
                              int bbox_elem, float grad_ce, float grad_box, float* losses, float* dcls, float* dbox,
                              void* stream_) {
  DetLossOpt opt;
  for (int e = 0; e < 8; ++e) opt.w[e] = 1.f;
  opt.sin_elem = -1;
  return launch_det_loss(cls_score, labels, num_rois, num_classes, bbox_pred, targets, inside, outside, bbox_elem, opt,
                         grad_ce, grad_box, losses, dcls, dbox, stream_);
}

extern "C" int frcnn_det_loss_lidar(const float* cls_score, const float* labels, int num_rois, int num_classes,
                                    const float* bbox_pred, const float* targets, const float* inside,
                                    const float* outside, const float* reg_loss_weight_host, int ry_sin, float grad_ce,
                                    float grad_box, float* losses, float* dcls, float* dbox, void* stream_) {
  DetLossOpt opt;
  for (int e = 0; e < 8; ++e) opt.w[e] = (reg_loss_weight_host && e < 7) ? reg_loss_weight_host[e] : 1.f;
  opt.sin_elem = ry_sin ? 6 : -1;
  return launch_det_loss(cls_score, labels, num_rois, num_classes, bbox_pred, targets, inside, outside, 7, opt, grad_ce,
                         grad_box, losses, dcls, dbox, stream_);
}

extern "C" int frcnn_det_loss_aleatoric(const float* cls_score, const float* labels, int num_rois, int num_classes,
                                        const float* bbox_pred, const float* bbox_var, const float* targets,
                                        const float* inside, const float* outside, int bbox_elem,
                                        const float* reg_loss_weight_host, int ry_sin, float grad_ce, float grad_box,
                                        float* losses, float* dcls, float* dbox, float* dvar, void* stream_) {
  FRCNN_REQUIRE(bbox_var, "det_loss_aleatoric: null bbox_var");
  DetLossOpt opt;
  for (int e = 0; e < 8; ++e) opt.w[e] = (reg_loss_weight_host && e < bbox_elem) ? reg_loss_weight_host[e] : 1.f;
  opt.sin_elem = (ry_sin && bbox_elem == 7) ? 6 : -1;
  return launch_det_loss(cls_score, labels, num_rois, num_classes, bbox_pred, targets, inside, outside, bbox_elem, opt,
                         grad_ce, grad_box, losses, dcls, dbox, stream_, bbox_var, dvar);
}

// ------------------------------------------------------------------------------------------------
// Statistics of T stochastic forward passes (MC-dropout / logit sampling), lib/utils/loss_utils.py:114-141:
//   bbox variance  (sum x^2 - (sum x)^2 / T) / (T - 1), clamped at 0             (compute_bbox_var)
//   entropy of the mean softmax (log2) and the mutual information H(mean p) - mean H(p)
//                                                        (categorical_entropy / categorical_mutual_information)
// One thread per output element / RoI; T and K are small (10 samples, a few classes).
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void mc_bbox_var_kernel(const float* __restrict__ samples, int T, size_t n,
                                                         float* __restrict__ var) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f, q = 0.f;
    for (int t = 0; t < T; ++t) {
      const float v = samples[(size_t)t * n + i];
      s += v;
      q += v * v;
    }
    float r = q + -(s * s) / (float)T;
    r = r / (float)(T - 1);
    var[i] = r > 0.f ? r : 0.f;
  }
}

__global__ __launch_bounds__(256) void mc_cls_stats_kernel(const float* __restrict__ scores, int T, int N, int K,
                                                          float* __restrict__ mean_prob, float* __restrict__ entropy,
                                                          float* __restrict__ mutual_info, float* __restrict__ prob_var) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    float plogp = 0.f;                       // sum over samples of sum_k p log2 p
    for (int k = 0; k < K; ++k) mean_prob[(size_t)n * K + k] = 0.f;
    if (prob_var)
      for (int k = 0; k < K; ++k) prob_var[(size_t)n * K + k] = 0.f;
    for (int t = 0; t < T; ++t) {
      const float* s = scores + ((size_t)t * N + n) * K;
      float m = s[0];
      for (int k = 1; k < K; ++k) m = fmaxf(m, s[k]);
      float den = 0.f;
      for (int k = 0; k < K; ++k) den += expf(s[k] - m);
      float acc = 0.f;
      for (int k = 0; k < K; ++k) {
        const float p = expf(s[k] - m) / den;
        mean_prob[(size_t)n * K + k] += p;
        if (prob_var) prob_var[(size_t)n * K + k] += p * p;
        acc += p * log2f(p);
      }
      plogp += acc;
    }
    float h = 0.f;
    for (int k = 0; k < K; ++k) {
      const float sum = mean_prob[(size_t)n * K + k];
      if (prob_var) {       // compute_bbox_var (loss_utils.py:114-120) applied to the softmax samples
        float r = prob_var[(size_t)n * K + k] + -(sum * sum) / (float)T;
        r = T > 1 ? r / (float)(T - 1) : 0.f;
        prob_var[(size_t)n * K + k] = r > 0.f ? r : 0.f;
      }
      const float p = sum / (float)T;
      mean_prob[(size_t)n * K + k] = p;
      h += p * log2f(p);
    }
    entropy[n] = -h;
    mutual_info[n] = plogp / (float)T + -h;
  }
}

// mean over the T leading samples of a (T, n) stack, summed in sample order
__global__ __launch_bounds__(256) void mc_mean_kernel(const float* __restrict__ samples, int T, size_t n,
                                                     float* __restrict__ mean) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += samples[(size_t)t * n + i];
    mean[i] = s / (float)T;
  }
}

// nn.Dropout in train() mode on `repeat` stochastic copies of x (n_in): y[t][i] = keep(t, i) ? x[i] / (1 - p) : 0 with
// keep = uniform01(seed, stream, t * n_in + i) >= p.  repeat > 1 starts the Monte-Carlo passes of the epistemic heads
// from ONE deterministic activation (lib/model/test.py:74-77: E_NUM_SAMPLE passes per frame).
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float* __restrict__ x, size_t n_in, int repeat, float p,
                                                         float scale, uint32_t seed,
                                                         const uint32_t* __restrict__ seed_dev, uint32_t stream,
                                                         float* __restrict__ y) {
  if (seed_dev) seed += *seed_dev;      // per-frame seed from device memory (a replayed hipGraph keeps `seed` itself)
  const size_t total = n_in * (size_t)repeat;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (size_t)gridDim.x * blockDim.x) {
    const float v = x[j % n_in];
    y[j] = uniform01(seed, stream, (uint32_t)j) >= p ? v * scale : 0.f;
  }
}
// dx[i] = sum_t keep(t, i) * dy[t][i] / (1 - p), summed in t order
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* __restrict__ dy, size_t n_in, int repeat, float p,
                                                         float scale, uint32_t seed,
                                                         const uint32_t* __restrict__ seed_dev, uint32_t stream,
                                                         float* __restrict__ dx) {
  if (seed_dev) seed += *seed_dev;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_in; i += (size_t)gridDim.x * blockDim.x) {
    float g = 0.f;
    for (int t = 0; t < repeat; ++t) {
      const size_t j = (size_t)t * n_in + i;
      if (uniform01(seed, stream, (uint32_t)j) >= p) g += dy[j] * scale;
    }
    dx[i] = g;
  }
}

// logit_distort (loss_utils.py:143-147): samples[s][i] = score[i] + sqrt(var[i]) * N(0,1)(seed; s * n + i).
// var_is_log: the head predicts s = log(var) (the convention of the box-variance head, lib/model/test.py:82); var_out
// (n, may be NULL) then receives exp(s), the a_cls_var column of the detections.
__global__ __launch_bounds__(256) void logit_distort_kernel(const float* __restrict__ score, const float* __restrict__ var,
                                                           size_t n, int S, uint32_t seed,
                                                           const uint32_t* __restrict__ seed_dev, uint32_t stream,
                                                           int var_is_log, float* __restrict__ out,
                                                           float* __restrict__ var_out) {
  if (seed_dev) seed += *seed_dev;
  const size_t total = n * (size_t)S;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (size_t)gridDim.x * blockDim.x) {
    const size_t i = j % n;
    const float v = var_is_log ? expf(var[i]) : var[i];
    out[j] = score[i] + sqrtf(v) * normal01(seed, stream, (uint32_t)j);
    if (var_out && j < n) var_out[i] = v;
  }
}

__global__ __launch_bounds__(256) void exp_kernel(const float* __restrict__ x, size_t n, float* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = expf(x[i]);
}

// bayesian_cross_entropy (loss_utils.py:149-169) with the reparameterised draws of logit_distort_kernel:
//   loss = mean_n -log( mean_s softmax(score_n + sqrt(var_n) * eps_sn)[target_n] )
// One thread per RoI; gradients w.r.t. score and var (through sqrt(var) * eps), scaled by grad / N.
__global__ __launch_bounds__(256) void bayes_ce_kernel(const float* __restrict__ score, const float* __restrict__ var,
                                                      const float* __restrict__ labels, int N, int K, int S,
                                                      uint32_t seed, const uint32_t* __restrict__ seed_dev,
                                                      uint32_t stream, int var_is_log, float grad,
                                                      float* __restrict__ per_roi, float* __restrict__ dscore,
                                                      float* __restrict__ dvar) {
  constexpr int KMAX = 16;
  if (seed_dev) seed += *seed_dev;
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const int tgt = (int)labels[n];
    float sc[KMAX], sd[KMAX], gs[KMAX], gv[KMAX];
    for (int k = 0; k < K; ++k) {
      sc[k] = score[(size_t)n * K + k];
      const float v = var[(size_t)n * K + k];
      sd[k] = sqrtf(var_is_log ? expf(v) : v);
      gs[k] = gv[k] = 0.f;
    }
    float avg = 0.f;
    for (int s = 0; s < S; ++s) {
      float z[KMAX], e[KMAX];
      float m = -INFINITY;
      for (int k = 0; k < K; ++k) {
        e[k] = normal01(seed, stream, (uint32_t)(((size_t)s * N + n) * K + k));
        z[k] = sc[k] + sd[k] * e[k];
        m = fmaxf(m, z[k]);
      }
      float den = 0.f;
      for (int k = 0; k < K; ++k) { z[k] = expf(z[k] - m); den += z[k]; }
      const float pt = z[tgt] / den;
      avg += pt;
      // d pt / d z_k = pt * ([k == tgt] - p_k)
      for (int k = 0; k < K; ++k) {
        const float d = pt * ((k == tgt ? 1.f : 0.f) - z[k] / den);
        gs[k] += d;
        gv[k] += d * e[k];
      }
    }
    avg = avg / (float)S;
    per_roi[n] = -logf(avg);
    // d(-log avg)/dz summed over s = -(1 / (S avg)) * sum_s dpt/dz ; dz/dvar = eps / (2 sqrt(var))
    const float c = -grad / ((float)N * (float)S * avg);
    for (int k = 0; k < K; ++k) {
      if (dscore) dscore[(size_t)n * K + k] = c * gs[k];
      // z = score + sd * eps: d z / d var = eps / (2 sd); with var = exp(s): d z / d s = eps * sd / 2
      if (dvar) dvar[(size_t)n * K + k] = var_is_log ? c * gv[k] * sd[k] * 0.5f : (sd[k] > 0.f ? c * gv[k] / (2.0f * sd[k]) : 0.f);
    }
  }
}
__global__ __launch_bounds__(256) void mean_reduce_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];      // fixed partition, fixed order
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = s / (float)n;
}
}  // namespace

extern "C" int frcnn_mc_bbox_var(const float* samples, int num_samples, int64_t elems, float* var, void* stream_) {
  FRCNN_REQUIRE(samples && var && num_samples > 1 && elems > 0, "mc_bbox_var: bad arguments (at least 2 samples)");
  hipLaunchKernelGGL(mc_bbox_var_kernel, dim3(grid_for((size_t)elems)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     samples, num_samples, (size_t)elems, var);
  return check_launch("mc_bbox_var_kernel");
}

extern "C" int frcnn_mc_cls_stats(const float* cls_score_samples, int num_samples, int num_rois, int num_classes,
                                  float* mean_prob, float* entropy, float* mutual_info, float* prob_var, void* stream_) {
  FRCNN_REQUIRE(cls_score_samples && mean_prob && entropy && mutual_info && num_samples > 0 && num_rois > 0 &&
                    num_classes > 1,
                "mc_cls_stats: bad arguments");
  hipLaunchKernelGGL(mc_cls_stats_kernel, dim3(grid_for((size_t)num_rois)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), cls_score_samples, num_samples, num_rois, num_classes, mean_prob,
                     entropy, mutual_info, prob_var);
  return check_launch("mc_cls_stats_kernel");
}

extern "C" int frcnn_mc_mean(const float* samples, int num_samples, int64_t elems, float* mean, void* stream_) {
  FRCNN_REQUIRE(samples && mean && num_samples > 0 && elems > 0, "mc_mean: bad arguments");
  hipLaunchKernelGGL(mc_mean_kernel, dim3(grid_for((size_t)elems)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     samples, num_samples, (size_t)elems, mean);
  return check_launch("mc_mean_kernel");
}

extern "C" int frcnn_dropout_fwd(const float* x, int64_t elems, int repeat, float p, uint32_t seed,
                                 const uint32_t* seed_dev, uint32_t stream_id, float* y, void* stream_) {
  FRCNN_REQUIRE(x && y && elems > 0 && repeat > 0 && p >= 0.f && p < 1.f && elems * (int64_t)repeat < ((int64_t)1 << 32),
                "dropout_fwd: bad arguments (0 <= p < 1, elems * repeat < 2^32)");
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3(grid_for((size_t)elems * repeat)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), x, (size_t)elems, repeat, p, 1.0f / (1.0f - p), seed, seed_dev,
                     stream_id, y);
  return check_launch("dropout_fwd_kernel");
}

extern "C" int frcnn_dropout_bwd(const float* dy, int64_t elems, int repeat, float p, uint32_t seed,
                                 const uint32_t* seed_dev, uint32_t stream_id, float* dx, void* stream_) {
  FRCNN_REQUIRE(dy && dx && elems > 0 && repeat > 0 && p >= 0.f && p < 1.f && elems * (int64_t)repeat < ((int64_t)1 << 32),
                "dropout_bwd: bad arguments (0 <= p < 1, elems * repeat < 2^32)");
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for((size_t)elems)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     dy, (size_t)elems, repeat, p, 1.0f / (1.0f - p), seed, seed_dev, stream_id, dx);
  return check_launch("dropout_bwd_kernel");
}

extern "C" int frcnn_logit_distort(const float* score, const float* var, int64_t elems, int num_samples, uint32_t seed,
                                   const uint32_t* seed_dev, uint32_t stream_id, int var_is_log, float* samples,
                                   float* var_out, void* stream_) {
  FRCNN_REQUIRE(score && var && samples && elems > 0 && num_samples > 0 && elems * (int64_t)num_samples < ((int64_t)1 << 32),
                "logit_distort: bad arguments (elems * num_samples < 2^32)");
  hipLaunchKernelGGL(logit_distort_kernel, dim3(grid_for((size_t)elems * num_samples)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), score, var, (size_t)elems, num_samples, seed, seed_dev, stream_id,
                     var_is_log, samples, var_out);
  return check_launch("logit_distort_kernel");
}

extern "C" int frcnn_bayesian_cross_entropy(const float* cls_score, const float* cls_var, const float* labels,
                                            int num_rois, int num_classes, int num_samples, uint32_t seed,
                                            const uint32_t* seed_dev, uint32_t stream_id, int var_is_log, float grad,
                                            float* loss, float* per_roi,
                                            float* dscore, float* dvar, void* stream_) {
  FRCNN_REQUIRE(cls_score && cls_var && labels && loss && per_roi && num_rois > 0 && num_classes > 1 && num_classes <= 16 &&
                    num_samples > 0 && (int64_t)num_rois * num_classes * num_samples < ((int64_t)1 << 32),
                "bayesian_cross_entropy: bad arguments (2 <= classes <= 16)");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(bayes_ce_kernel, dim3(grid_for((size_t)num_rois)), dim3(256), 0, stream, cls_score, cls_var, labels,
                     num_rois, num_classes, num_samples, seed, seed_dev, stream_id, var_is_log, grad, per_roi, dscore, dvar);
  int rc = check_launch("bayes_ce_kernel");
  if (rc != FRCNN_OK) return rc;
  hipLaunchKernelGGL(mean_reduce_kernel, dim3(1), dim3(256), 0, stream, per_roi, num_rois, loss);
  return check_launch("mean_reduce_kernel");
}

extern "C" int frcnn_exp(const float* x, int64_t elems, float* y, void* stream_) {
  FRCNN_REQUIRE(x && y && elems > 0, "exp: bad arguments");
  hipLaunchKernelGGL(exp_kernel, dim3(grid_for((size_t)elems)), dim3(256), 0, static_cast<hipStream_t>(stream_), x,
                     (size_t)elems, y);
  return check_launch("exp_kernel");
}
