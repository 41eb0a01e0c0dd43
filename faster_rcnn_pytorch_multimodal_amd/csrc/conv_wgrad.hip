// Filter gradient of the convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32):
//
//     dw[k][r][s][c] = sum over output pixels m = (n, ho, wo) of  dy[m][k] * x[n, ho*stride-pad+r, wo*stride-pad+s, c]
//
// Replaces autograd's conv weight gradient for the trainable convolutions of lib/nets/resnet.py:98-127,
// lib/nets/fpn.py:33-39 and the RPN / tail layers (lib/model/train_val.py:458 -> loss.backward()).
//
// GEMM view: rows = k (output channels), columns = q = (r*S+s)*C + c, reduction over the M pixels.
// Both operands are contiguous along their ROW/COLUMN index and strided along the reduction index (the
// opposite of the forward kernel), so tiles are staged as [32 pixels][128 k] and [32 pixels][128 q] with
// 16-byte chunks along k / q and the MFMA fragments are ds_read_b32 with the lanes running along k / q
// (32 consecutive floats per half wave: conflict-free without padding).  A thread's q chunk — hence its
// filter tap (r, s) and channel c — is fixed for the whole kernel; only the pixel advances.
// The pixel range is split over gridDim.z; partial sums go to slabs that a second kernel adds in z order,
// so the result is deterministic.
#include "common.h"

#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <vector>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BR = 32;    // pixels per reduction step
constexpr int NUM_CU = 256;
// tile edge BT = 64 * TI (k rows x q columns), 2 x 2 waves each owning TI x TI MFMA tiles of 32 x 32:
//   TI = 2: 128 x 128, the arithmetic-intensity choice for large pixel counts;
//   TI = 1: 64 x 64, four times as many tiles -> fewer pixel splits (less slab traffic, longer K loops) when the
//           filter is small and the pixel count short (layer3 / layer4 of one frame: M = 2394 / 608).

struct WgradParams {
  const float* x;
  const float* dy;
  float* out;  // dw [K][RSC] (splits == 1) or slabs [splits][K][RSC]
  int H, W, C, K, R, S, stride, pad, Ho, Wo;
  int M, Q;  // pixels, R*S*C
  int steps, steps_per_split;
  int tiles_k, tiles_q;
};

template <int TI>
__device__ __forceinline__ void conv_wgrad_body(const WgradParams& p, const float* __restrict__ px,
                                                const float* __restrict__ pdy, float* __restrict__ pout) {
  constexpr int BT = 64 * TI;
  constexpr int CH = BT / 4;            // 16-byte chunks per staged pixel row
  constexpr int RP = 256 / CH;          // pixel rows staged per pass
  constexpr int NP = BR / RP;           // passes per reduction step
  __shared__ __attribute__((aligned(16))) float As[2][BR][BT];  // dy tile   [pixel][k]
  __shared__ __attribute__((aligned(16))) float Bs[2][BR][BT];  // x tile    [pixel][q]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tile_k = blockIdx.x / p.tiles_q, tile_q = blockIdx.x - tile_k * p.tiles_q;
  const int k0 = tile_k * BT, q0 = tile_q * BT;
  const int step_begin = blockIdx.z * p.steps_per_split;
  const int step_end = min(step_begin + p.steps_per_split, p.steps);

  // staging: thread t moves 16-byte chunk (t % CH) of pixel rows (t / CH) + RP*i, i < NP
  const int ch = t % CH, prow = t / CH;
  const int ka = k0 + ch * 4;            // first of this thread's 4 output channels
  const bool ka_ok = ka < p.K;           // K % 4 == 0: a chunk is all in or all out
  const int qb = q0 + ch * 4;            // first of this thread's 4 filter elements
  const bool qb_ok = qb < p.Q;
  const int tap = qb_ok ? qb / p.C : 0;
  const int cb = qb - tap * p.C;
  const int tr = tap / p.S, ts = tap - tr * p.S;

  f32x4 ra[NP], rb[NP];
  auto load_tiles = [&](int step) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int m = step * BR + prow + RP * i;
      const bool m_ok = m < p.M;
      ra[i] = (m_ok && ka_ok) ? *reinterpret_cast<const f32x4*>(pdy + (size_t)m * p.K + ka) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m_ok && qb_ok) {
        const int img = m / (p.Ho * p.Wo);
        const int rem = m - img * p.Ho * p.Wo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int hi = ho * p.stride - p.pad + tr, wi = wo * p.stride - p.pad + ts;
        if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
          v = *reinterpret_cast<const f32x4*>(px + ((size_t)(img * p.H + hi) * p.W + wi) * p.C + cb);
      }
      rb[i] = v;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      *reinterpret_cast<f32x4*>(&As[buf][prow + RP * i][ch * 4]) = ra[i];
      *reinterpret_cast<f32x4*>(&Bs[buf][prow + RP * i][ch * 4]) = rb[i];
    }
  };

  f32x16 acc[TI][TI];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  if (step_begin < step_end) {
    load_tiles(step_begin);
    store_tiles(0);
  }
  __syncthreads();
  int cur = 0;
  for (int step = step_begin; step < step_end; ++step) {
    const bool more = step + 1 < step_end;
    if (more) load_tiles(step + 1);
#pragma unroll
    for (int kk = 0; kk < BR / 2; ++kk) {
      float a[TI], b[TI];
#pragma unroll
      for (int i = 0; i < TI; ++i) a[i] = As[cur][2 * kk + lh][(wr * TI + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < TI; ++j) b[j] = Bs[cur][2 * kk + lh][(wc * TI + j) * 32 + l31];
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tiles(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // D: column (lane & 31) = q, row (r&3) + 8*(r>>2) + 4*(lane>>5) = k
  float* out = pout + (size_t)blockIdx.z * p.K * p.Q;
#pragma unroll
  for (int j = 0; j < TI; ++j) {
    const int q = q0 + (wc * TI + j) * 32 + l31;
    if (q >= p.Q) continue;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = k0 + (wr * TI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (k < p.K) out[(size_t)k * p.Q + q] = acc[i][j][r];
      }
  }
}

template <int TI>
__global__ __launch_bounds__(256, 2) void conv_wgrad_f32(const WgradParams p) {
  conv_wgrad_body<TI>(p, p.x, p.dy, p.out);
}

// Grouped form: blockIdx.y selects one of up to WG_MAX_GROUPS convolutions of IDENTICAL shape (the 22 repeated Bottlenecks
// of layer3, lib/nets/resnet.py:131-240): together they have enough output tiles to fill the chip WITHOUT splitting the
// pixel reduction, so every tile runs the whole M-pixel loop (75 steps instead of ~9) and no slabs are summed afterwards.
// Group g writes its (K, Q) result to p.out + g * K * Q.
constexpr int WG_MAX_GROUPS = 24;
struct WgradGroups {
  const float* x[WG_MAX_GROUPS];
  const float* dy[WG_MAX_GROUPS];
};
struct WgradOuts {
  float* grad[WG_MAX_GROUPS];
};
template <int TI>
__global__ __launch_bounds__(256, 2) void conv_wgrad_grouped_f32(const WgradParams p, const WgradGroups g) {
  const int grp = blockIdx.y;
  conv_wgrad_body<TI>(p, g.x[grp], g.dy[grp], p.out + (size_t)grp * p.K * p.Q);
}

// dw = sum_z slab[z] (z order), 16 bytes per thread
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int splits, size_t n4,
                                                          float* __restrict__ dw) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 v = reinterpret_cast<const f32x4*>(slabs)[i];
    for (int z = 1; z < splits; ++z) {
      const f32x4 u = reinterpret_cast<const f32x4*>(slabs)[(size_t)z * n4 + i];
      v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
    }
    reinterpret_cast<f32x4*>(dw)[i] = v;
  }
}

// db[k] = sum_m dy[m][k] in two deterministic passes: BIAS_GROUPS x (K/64) workgroups sum interleaved pixel
// subsets (4 waves each) into partial[g][k], then one pass adds the groups in g order.
constexpr int BIAS_GROUPS = 64;
// Slab sum + layout change + accumulation in one pass: grad (K, c_real, R, S) - a Conv2d / Linear parameter's own layout -
// += sum_z slabs[z] (K, R, S, C) restricted to the real channels.  One thread per OUTPUT element (coalesced read-modify-write of
// the gradient); the slab reads stride by C floats (small tensors).  z ascending: deterministic.
__global__ __launch_bounds__(256) void wgrad_accumulate_kernel(const float* __restrict__ slabs, int splits, int K, int R, int S,
                                                              int C, int c_real, float* __restrict__ grad) {
  const size_t total = (size_t)K * c_real * R * S, slab = (size_t)K * R * S * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int s = (int)(i % S);
    size_t t = i / S;
    const int r = (int)(t % R);
    t /= R;
    const int c = (int)(t % c_real);
    const int k = (int)(t / c_real);
    const size_t src = (((size_t)k * R + r) * S + s) * C + c;
    float v = slabs[src];
    for (int z = 1; z < splits; ++z) v += slabs[(size_t)z * slab + src];
    grad[i] += v;
  }
}

// the same for the grouped filter gradient: one (K, R, S, C) result per group, added into that group's parameter gradient
__global__ __launch_bounds__(256) void wgrad_accumulate_grouped_kernel(const float* __restrict__ results, int K, int R, int S,
                                                                      int C, int c_real, const WgradOuts o) {
  const size_t total = (size_t)K * c_real * R * S, slab = (size_t)K * R * S * C;
  const float* src0 = results + (size_t)blockIdx.y * slab;
  float* grad = o.grad[blockIdx.y];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int s = (int)(i % S);
    size_t t = i / S;
    const int r = (int)(t % R);
    t /= R;
    const int c = (int)(t % c_real);
    const int k = (int)(t / c_real);
    grad[i] += src0[(((size_t)k * R + r) * S + s) * C + c];
  }
}

__global__ __launch_bounds__(256) void bias_grad_accumulate_kernel(const float* __restrict__ partial, int K, int groups,
                                                                  float* __restrict__ db) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  float v = 0.f;
  for (int g = 0; g < groups; ++g) v += partial[(size_t)g * K + k];
  db[k] += v;
}

__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const float* __restrict__ dy, int M, int K,
                                                               float* __restrict__ partial) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (k < K)
    for (int m = blockIdx.y * 4 + wave; m < M; m += 4 * BIAS_GROUPS) s += dy[(size_t)m * K + k];
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && k < K)
    partial[(size_t)blockIdx.y * K + k] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}
__global__ __launch_bounds__(256) void bias_grad_final_kernel(const float* __restrict__ partial, int K,
                                                             float* __restrict__ db) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  float s = partial[k];
  for (int g = 1; g < BIAS_GROUPS; ++g) s += partial[(size_t)g * K + k];
  db[k] = s;
}

// Pixel splits for a given tile count: estimated cycles of the slowest CU (two workgroups per CU share the SIMDs)
// plus the reduction pass.  ti scales the per-step MFMA work (TI*TI tiles per wave).
int choose_splits(int tiles, int steps, int ti) {
  int best = 1;
  double best_t = 1e300;
  const double step_cyc = 2.0 * 1024.0 * ti * ti;
  for (int sp = 1; sp <= 64; ++sp) {
    const int sps = (steps + sp - 1) / sp;
    const int real = (steps + sps - 1) / sps;
    if (real != sp) continue;
    const long rounds = ((long)tiles * real + 2 * NUM_CU - 1) / (2 * NUM_CU);  // two workgroups per CU
    const double tcyc = rounds * (sps * step_cyc + 6000.0) + (real > 1 ? real * 300.0 : 0.0);
    if (tcyc < best_t) {
      best_t = tcyc;
      best = real;
    }
  }
  return best;
}

struct WgradPlan {
  int ti, splits;
};
inline int tiles_for(int k, int q, int ti) { return ((k + 64 * ti - 1) / (64 * ti)) * ((q + 64 * ti - 1) / (64 * ti)); }

// Default plan: the 128 x 128 tile and the modelled split; in autotune mode (frcnn_conv2d_set_autotune, shared with the
// forward kernel) the first call of a shape times both tile sizes around the modelled split and caches the fastest.
typedef std::array<int, 9> WgradKey;
std::map<WgradKey, WgradPlan> g_wgrad_plans;
std::mutex g_wgrad_mutex;

std::vector<WgradPlan> wgrad_candidates(int k, int q, int steps) {
  std::vector<WgradPlan> out;
  for (int ti = 2; ti >= 1; --ti) {
    const int base = choose_splits(tiles_for(k, q, ti), steps, ti);
    for (int sp : {base, std::max(1, base / 2), std::min(64, base * 2), 1}) {
      const int sps = (steps + sp - 1) / sp;
      const int real = (steps + sps - 1) / sps;
      bool dup = false;
      for (const WgradPlan& c : out) dup = dup || (c.ti == ti && c.splits == real);
      if (!dup && (size_t)real * k * q * sizeof(float) <= ((size_t)1 << 28)) out.push_back(WgradPlan{ti, real});
    }
  }
  return out;
}

bool wgrad_args_ok(int n, int h, int w, int c, int k, int r, int s, int stride, int pad) {
  return n > 0 && h > 0 && w > 0 && c > 0 && (c % 4) == 0 && k > 0 && (k % 4) == 0 && r > 0 && s > 0 && stride > 0 &&
         pad >= 0 && (h + 2 * pad - r) >= 0 && (w + 2 * pad - s) >= 0;
}

}  // namespace

void frcnn::clear_wgrad_plans() {
  std::lock_guard<std::mutex> lock(g_wgrad_mutex);
  g_wgrad_plans.clear();
}

namespace {

WgradKey wgrad_key(int n, int h, int w, int c, int k, int r, int s, int stride, int pad) {
  return WgradKey{n, h, w, c, k, r, s, stride, pad};
}

bool lookup_wgrad(const WgradKey& key, WgradPlan* pl) {
  std::lock_guard<std::mutex> lock(g_wgrad_mutex);
  auto it = g_wgrad_plans.find(key);
  if (it == g_wgrad_plans.end()) return false;
  *pl = it->second;
  return true;
}

size_t slab_bytes_of(const WgradPlan& pl, int k, int q) {
  return pl.splits > 1 ? (size_t)pl.splits * k * q * sizeof(float) : 0;
}

// main kernel + the slab reduction of one plan
int launch_wgrad(WgradParams p, const WgradPlan& pl, float* dw, void* ws, hipStream_t stream, bool slabs_only = false) {
  const int bt = 64 * pl.ti;
  p.tiles_k = (p.K + bt - 1) / bt;
  p.tiles_q = (p.Q + bt - 1) / bt;
  p.steps_per_split = (p.steps + pl.splits - 1) / pl.splits;
  p.out = (pl.splits > 1 || slabs_only) ? static_cast<float*>(ws) : dw;   // slabs_only: the caller reduces the slab(s) itself
  const dim3 grid(p.tiles_k * p.tiles_q, 1, pl.splits);
  if (pl.ti == 2) hipLaunchKernelGGL(conv_wgrad_f32<2>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(conv_wgrad_f32<1>, grid, dim3(256), 0, stream, p);
  int rc = frcnn::check_launch("conv_wgrad_f32");
  if (rc != FRCNN_OK) return rc;
  if (pl.splits > 1 && !slabs_only) {
    const size_t n4 = (size_t)p.K * p.Q / 4;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 4096)), dim3(256), 0,
                       stream, static_cast<const float*>(ws), pl.splits, n4, dw);
    rc = frcnn::check_launch("wgrad_reduce_kernel");
  }
  return rc;
}

// Time every candidate twice on the caller's tensors (best-of), outside stream capture only.
bool tune_wgrad(const WgradParams& p, float* dw, void* ws, size_t ws_avail, hipStream_t stream, WgradPlan* best) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return false;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess) return false;
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return false; }
  const std::vector<WgradPlan> cands = wgrad_candidates(p.K, p.Q, p.steps);
  std::vector<float> best_of(cands.size(), 1e30f);
  for (int pass = 0; pass < 2; ++pass)
    for (size_t ci = 0; ci < cands.size(); ++ci) {
      if (slab_bytes_of(cands[ci], p.K, p.Q) > ws_avail) continue;
      if (pass == 0 && launch_wgrad(p, cands[ci], dw, ws, stream) != FRCNN_OK) continue;   // warm-up
      (void)hipEventRecord(e0, stream);
      bool ok = true;
      for (int i = 0; i < 3 && ok; ++i) ok = launch_wgrad(p, cands[ci], dw, ws, stream) == FRCNN_OK;
      (void)hipEventRecord(e1, stream);
      float ms = 0.f;
      if (hipEventSynchronize(e1) != hipSuccess || !ok || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
      best_of[ci] = std::min(best_of[ci], ms);
    }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  float best_ms = 1e30f;
  bool found = false;
  for (size_t ci = 0; ci < cands.size(); ++ci)
    if (best_of[ci] < best_ms) { best_ms = best_of[ci]; *best = cands[ci]; found = true; }
  return found;
}

}  // namespace

extern "C" size_t frcnn_conv2d_bwd_weight_ws_bytes(int n, int h, int w, int c, int k, int r, int s, int stride,
                                                   int pad) {
  if (!wgrad_args_ok(n, h, w, c, k, r, s, stride, pad)) return 0;
  const int ho = (h + 2 * pad - r) / stride + 1, wo = (w + 2 * pad - s) / stride + 1;
  const long M = (long)n * ho * wo;
  const int q = r * s * c;
  const int steps = (int)((M + BR - 1) / BR);
  WgradPlan pl;
  size_t slabs;
  if (lookup_wgrad(wgrad_key(n, h, w, c, k, r, s, stride, pad), &pl)) {
    slabs = slab_bytes_of(pl, k, q);
  } else if (frcnn::autotune_enabled()) {      // room for the largest candidate of a shape that is about to be tuned
    slabs = 0;
    for (const WgradPlan& cand : wgrad_candidates(k, q, steps)) slabs = std::max(slabs, slab_bytes_of(cand, k, q));
  } else {
    pl = WgradPlan{2, choose_splits(tiles_for(k, q, 2), steps, 2)};
    slabs = slab_bytes_of(pl, k, q);
  }
  // slabs (when the pixel range is split; at least one, which frcnn_conv2d_bwd_weight_acc reduces from) + the
  // bias-gradient partials
  slabs = std::max(slabs, (size_t)k * q * sizeof(float));
  return frcnn::align_up(slabs, 256) + (size_t)BIAS_GROUPS * k * sizeof(float);
}

extern "C" int frcnn_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* db, int n, int h, int w,
                                       int c, int k, int r, int s, int stride, int pad, void* ws, size_t ws_bytes,
                                       void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FRCNN_REQUIRE(x && dy && dw, "conv2d_bwd_weight: null tensor");
  FRCNN_REQUIRE(wgrad_args_ok(n, h, w, c, k, r, s, stride, pad),
                "conv2d_bwd_weight: bad shape n=%d h=%d w=%d c=%d k=%d r=%d s=%d stride=%d pad=%d (need c%%4==0, k%%4==0)",
                n, h, w, c, k, r, s, stride, pad);
  WgradParams p;
  p.x = x; p.dy = dy; p.out = nullptr;
  p.H = h; p.W = w; p.C = c; p.K = k; p.R = r; p.S = s; p.stride = stride; p.pad = pad;
  p.Ho = (h + 2 * pad - r) / stride + 1;
  p.Wo = (w + 2 * pad - s) / stride + 1;
  const long M = (long)n * p.Ho * p.Wo;
  FRCNN_REQUIRE(M * (long)k < (1L << 31) && (long)n * h * w * c < (1L << 31), "conv2d_bwd_weight: tensor too large");
  p.M = (int)M;
  p.Q = r * s * c;
  p.steps = (p.M + BR - 1) / BR;
  p.steps_per_split = p.steps; p.tiles_k = p.tiles_q = 0;
  const size_t bias_bytes = db ? (size_t)BIAS_GROUPS * k * sizeof(float) : 0;
  const size_t slab_room = ws_bytes > bias_bytes ? ((ws_bytes - bias_bytes) / 256) * 256 : 0;   // slabs first, 256-aligned
  const WgradKey key = wgrad_key(n, h, w, c, k, r, s, stride, pad);
  WgradPlan pl;
  bool have = lookup_wgrad(key, &pl);
  if (!have && frcnn::autotune_enabled() && ws && tune_wgrad(p, dw, ws, slab_room, stream, &pl)) {
    std::lock_guard<std::mutex> lock(g_wgrad_mutex);
    g_wgrad_plans[key] = pl;
    have = true;
  }
  if (!have) pl = WgradPlan{2, choose_splits(tiles_for(k, p.Q, 2), p.steps, 2)};
  const size_t slab_bytes = frcnn::align_up(slab_bytes_of(pl, k, p.Q), 256);
  const size_t need = slab_bytes + bias_bytes;
  if (need > 0 && (!ws || ws_bytes < need))
    return frcnn::fail(FRCNN_ERR_WS, "conv2d_bwd_weight: workspace %zu < %zu bytes", ws_bytes, need);
  int rc = launch_wgrad(p, pl, dw, ws, stream);
  if (rc != FRCNN_OK) return rc;
  if (db) {
    float* partial = reinterpret_cast<float*>(static_cast<char*>(ws) + slab_bytes);
    hipLaunchKernelGGL(bias_grad_partial_kernel, dim3((k + 63) / 64, BIAS_GROUPS), dim3(256), 0, stream, dy, p.M, k,
                       partial);
    rc = frcnn::check_launch("bias_grad_partial_kernel");
    if (rc != FRCNN_OK) return rc;
    hipLaunchKernelGGL(bias_grad_final_kernel, dim3((k + 255) / 256), dim3(256), 0, stream, partial, k, db);
    rc = frcnn::check_launch("bias_grad_final_kernel");
  }
  return rc;
}

extern "C" int frcnn_conv2d_bwd_weight_acc(const float* x, const float* dy, float* grad_w, int c_real, float* grad_b, int n,
                                           int h, int w, int c, int k, int r, int s, int stride, int pad, void* ws,
                                           size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FRCNN_REQUIRE(x && dy && grad_w && c_real > 0 && c_real <= c, "conv2d_bwd_weight_acc: null tensor or c_real out of range");
  FRCNN_REQUIRE(wgrad_args_ok(n, h, w, c, k, r, s, stride, pad),
                "conv2d_bwd_weight_acc: bad shape n=%d h=%d w=%d c=%d k=%d r=%d s=%d stride=%d pad=%d (need c%%4==0, k%%4==0)",
                n, h, w, c, k, r, s, stride, pad);
  WgradParams p;
  p.x = x; p.dy = dy; p.out = nullptr;
  p.H = h; p.W = w; p.C = c; p.K = k; p.R = r; p.S = s; p.stride = stride; p.pad = pad;
  p.Ho = (h + 2 * pad - r) / stride + 1;
  p.Wo = (w + 2 * pad - s) / stride + 1;
  const long M = (long)n * p.Ho * p.Wo;
  FRCNN_REQUIRE(M * (long)k < (1L << 31) && (long)n * h * w * c < (1L << 31), "conv2d_bwd_weight_acc: tensor too large");
  p.M = (int)M;
  p.Q = r * s * c;
  p.steps = (p.M + BR - 1) / BR;
  p.steps_per_split = p.steps; p.tiles_k = p.tiles_q = 0;
  const size_t bias_bytes = grad_b ? (size_t)BIAS_GROUPS * k * sizeof(float) : 0;
  WgradPlan pl;
  if (!lookup_wgrad(wgrad_key(n, h, w, c, k, r, s, stride, pad), &pl))
    pl = WgradPlan{2, choose_splits(tiles_for(k, p.Q, 2), p.steps, 2)};      // plans are tuned by frcnn_conv2d_bwd_weight
  const size_t slab_bytes = frcnn::align_up(std::max(slab_bytes_of(pl, k, p.Q), (size_t)k * p.Q * sizeof(float)), 256);
  if (!ws || ws_bytes < slab_bytes + bias_bytes)
    return frcnn::fail(FRCNN_ERR_WS, "conv2d_bwd_weight_acc: workspace %zu < %zu bytes", ws_bytes, slab_bytes + bias_bytes);
  int rc = launch_wgrad(p, pl, nullptr, ws, stream, true);
  if (rc != FRCNN_OK) return rc;
  const size_t total = (size_t)k * c_real * r * s;
  hipLaunchKernelGGL(wgrad_accumulate_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 8192)), dim3(256), 0,
                     stream, static_cast<const float*>(ws), pl.splits, k, r, s, c, c_real, grad_w);
  rc = frcnn::check_launch("wgrad_accumulate_kernel");
  if (rc != FRCNN_OK || !grad_b) return rc;
  float* partial = reinterpret_cast<float*>(static_cast<char*>(ws) + slab_bytes);
  hipLaunchKernelGGL(bias_grad_partial_kernel, dim3((k + 63) / 64, BIAS_GROUPS), dim3(256), 0, stream, dy, p.M, k, partial);
  rc = frcnn::check_launch("bias_grad_partial_kernel");
  if (rc != FRCNN_OK) return rc;
  hipLaunchKernelGGL(bias_grad_accumulate_kernel, dim3((k + 255) / 256), dim3(256), 0, stream, partial, k, BIAS_GROUPS, grad_b);
  return frcnn::check_launch("bias_grad_accumulate_kernel");
}

// Filter gradients of `groups` convolutions of identical shape in one launch pair (conv_wgrad_grouped_f32 +
// wgrad_accumulate_grouped_kernel): grad_w[g] (K, c_real, R, S) += dW(x[g], dy[g]).  x / dy / grad_w are HOST arrays of device
// pointers.  No pixel split, fixed summation order: deterministic.
extern "C" size_t frcnn_conv2d_bwd_weight_acc_grouped_ws_bytes(int groups, int c, int k, int r, int s) {
  if (groups <= 0 || c <= 0 || k <= 0 || r <= 0 || s <= 0) return 0;
  return frcnn::align_up((size_t)groups * k * r * s * c * sizeof(float), 256);
}

extern "C" int frcnn_conv2d_bwd_weight_acc_grouped(const float* const* x, const float* const* dy, float* const* grad_w,
                                                   int groups, int c_real, int n, int h, int w, int c, int k, int r, int s,
                                                   int stride, int pad, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FRCNN_REQUIRE(x && dy && grad_w && groups > 0 && groups <= WG_MAX_GROUPS && c_real > 0 && c_real <= c,
                "conv2d_bwd_weight_acc_grouped: bad arguments (1..%d groups)", WG_MAX_GROUPS);
  FRCNN_REQUIRE(wgrad_args_ok(n, h, w, c, k, r, s, stride, pad),
                "conv2d_bwd_weight_acc_grouped: bad shape n=%d h=%d w=%d c=%d k=%d r=%d s=%d stride=%d pad=%d", n, h, w, c, k,
                r, s, stride, pad);
  WgradParams p;
  p.x = nullptr; p.dy = nullptr; p.out = static_cast<float*>(ws);
  p.H = h; p.W = w; p.C = c; p.K = k; p.R = r; p.S = s; p.stride = stride; p.pad = pad;
  p.Ho = (h + 2 * pad - r) / stride + 1;
  p.Wo = (w + 2 * pad - s) / stride + 1;
  const long M = (long)n * p.Ho * p.Wo;
  FRCNN_REQUIRE(M * (long)k < (1L << 31) && (long)n * h * w * c < (1L << 31), "conv2d_bwd_weight_acc_grouped: tensor too large");
  p.M = (int)M;
  p.Q = r * s * c;
  p.steps = (p.M + BR - 1) / BR;
  p.steps_per_split = p.steps;
  const size_t need = frcnn_conv2d_bwd_weight_acc_grouped_ws_bytes(groups, c, k, r, s);
  if (!ws || ws_bytes < need)
    return frcnn::fail(FRCNN_ERR_WS, "conv2d_bwd_weight_acc_grouped: workspace %zu < %zu bytes", ws_bytes, need);
  WgradGroups g;
  WgradOuts o;
  for (int i = 0; i < WG_MAX_GROUPS; ++i) {
    const int j = i < groups ? i : 0;
    FRCNN_REQUIRE(x[j] && dy[j] && grad_w[j], "conv2d_bwd_weight_acc_grouped: null tensor in group %d", j);
    g.x[i] = x[j]; g.dy[i] = dy[j]; o.grad[i] = grad_w[j];
  }
  // the 128 x 128 tile when the groups together still give every CU a workgroup, else 64 x 64
  const int ti = (long)tiles_for(k, p.Q, 2) * groups >= NUM_CU ? 2 : 1;
  const int bt = 64 * ti;
  p.tiles_k = (p.K + bt - 1) / bt;
  p.tiles_q = (p.Q + bt - 1) / bt;
  const dim3 grid(p.tiles_k * p.tiles_q, groups, 1);
  if (ti == 2) hipLaunchKernelGGL(conv_wgrad_grouped_f32<2>, grid, dim3(256), 0, stream, p, g);
  else hipLaunchKernelGGL(conv_wgrad_grouped_f32<1>, grid, dim3(256), 0, stream, p, g);
  int rc = frcnn::check_launch("conv_wgrad_grouped_f32");
  if (rc != FRCNN_OK) return rc;
  const size_t total = (size_t)k * c_real * r * s;
  hipLaunchKernelGGL(wgrad_accumulate_grouped_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 2048), groups),
                     dim3(256), 0, stream, static_cast<const float*>(ws), k, r, s, c, c_real, o);
  return frcnn::check_launch("wgrad_accumulate_grouped_kernel");
}
