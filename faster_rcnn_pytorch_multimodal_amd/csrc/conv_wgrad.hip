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
// The pixel range is split over gridDim.z; partial sums go to slabs that are added in z order, so the result is
// deterministic.
// Two kernels: conv_wgrad_f32 (rounds 1-4: register-staged tiles, slabs added by wgrad_reduce_kernel / wgrad_accumulate_kernel;
// plans 1 / 2, kept for operands beyond 2 GB, callers without tile counters and A/B) and conv_wgrad_dma_f32 (round 5: LDS-DMA
// ring, ds_read_b64 fragments, the LAST workgroup of a tile adds the slabs and stores / accumulates itself; plans 3 / 4).
#include "common.h"

#include <algorithm>
#include <array>
#include <atomic>
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
  // conv_wgrad_dma_f32 only
  float* target;     // mode 0: dw [K][RSC], overwritten; mode 1: the parameter's gradient (K, c_real, R, S), accumulated into
  int* counters;     // one per tile, zero on entry and on exit (splits > 1)
  int mode, c_real, N, splits;
  unsigned xbytes, dybytes;
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

// ------------------------------------------------------------------------------------------------
// LDS-DMA form of the filter gradient (plan codes 3 = 64 x 64 tile, 4 = 128 x 128 tile).
//
// What the register-staged kernel above loses (profiles/r05_train_busy.txt: conv_wgrad_f32<1> 24.8 % MfmaUtil, 5 ms of the
// 18.7 ms of kernel time per training step, + 1.2 ms of wgrad_accumulate_kernel):
//   * its prefetch is ONE reduction step deep (registers), a step of the 64 x 64 tile is 1024 MFMA cycles per wave, a global
//     load under load takes longer: every step waits;
//   * every MFMA of the 64 x 64 tile needs two ds_read_b32 (nothing issues for free beside an fp32 MFMA);
//   * the pixel-split partial sums go through slabs, a reduction kernel and a third kernel that changes the layout and adds
//     into the parameter's gradient.
// Here:
//   * the [32 pixels][BT] tiles of dy and x go global -> LDS by buffer_load_dwordx4 ... lds into a ring (four stages of 16 KB
//     for the 64-tile: three steps in flight; two stages of 32 KB for the 128-tile), one barrier per step.  Rows past M and
//     chunks past K / Q are out of the buffer's range -> zeros, no selects.  dy addresses advance by one v_add per step; x
//     addresses likewise for a 1x1 / stride 1 layer (UNIT), else (img, ho, wo) advance incrementally (no division in the loop);
//   * fragments are ds_read_b64: a lane holds TWO adjacent k (q) per pixel, i.e. MFMA operand t covers rows k = base + 2 i + t.
//     The permutation only moves results between accumulators (undone in the epilogue); one b64 pair feeds four MFMAs.
//     64-tile: every wave computes the WHOLE 64 x 64 tile over a quarter of each step's pixels (LDS bytes are read once
//     instead of twice), the four partial tiles are added in wave order through LDS afterwards.  128-tile: 2 x 2 waves of
//     64 x 64 - the summation order of conv_wgrad_f32<2>, bit-identical to it;
//   * the epilogue goes through LDS to 16-byte chunks and finishes the job: splits == 1 -> store dw or add into the gradient
//     in the parameter's own layout; splits > 1 -> slab, device-scope release, tile counter; the LAST workgroup of a tile
//     adds the slabs in z order (deterministic whoever is last) and does the same.  No second or third kernel.
// ------------------------------------------------------------------------------------------------
constexpr unsigned WG_OOB = 0x80000000u;
constexpr int WG_LDS_FLOATS = 16384;   // 64 KB

template <int TI, bool UNIT>
__device__ __forceinline__ void conv_wgrad_dma_body(const WgradParams& p, const float* __restrict__ px,
                                                    const float* __restrict__ pdy, float* __restrict__ ptarget,
                                                    float* smem) {
  constexpr int BT = 64 * TI;
  constexpr int CH = BT / 4;             // 16-byte chunks per tile row
  constexpr int RPI = 64 / CH;           // tile rows per DMA instruction (1 KB)
  constexpr int PW = BR / RPI / 4;       // DMA instructions per wave, operand and stage
  constexpr int NST = TI == 1 ? 4 : 2;   // ring stages
  constexpr int DIST = NST - 1;          // steps in flight
  constexpr int STAGE = 2 * BR * BT;     // floats per stage: dy tile, then x tile
  constexpr int KK = TI == 1 ? 4 : 16;   // MFMA k-iterations (2 pixels each) per wave and step
  static_assert(NST * STAGE == WG_LDS_FLOATS, "ring = 64 KB");
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef float f32x2 __attribute__((ext_vector_type(2)));

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int tile_k = blockIdx.x / p.tiles_q, tile_q = blockIdx.x - tile_k * p.tiles_q;
  const int k0 = tile_k * BT, q0 = tile_q * BT;
  const int step_begin = blockIdx.z * p.steps_per_split;
  const int step_end = min(step_begin + p.steps_per_split, p.steps);
  const int nsteps = max(step_end - step_begin, 0);

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pdy), 0, (int)p.dybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(px), 0, (int)p.xbytes, 0x00020000);

  // ---- DMA sources: lane i of instruction j of this wave feeds tile row (wave PW + j) RPI + i / CH, chunk i % CH ------------
  const int lrow = lane / CH, lchunk = lane % CH;
  const int ka = k0 + lchunk * 4, qb = q0 + lchunk * 4;
  const bool ka_ok = ka < p.K, qb_ok = qb < p.Q;
  unsigned vA[PW], vB[PW];
  int s_img[PW], s_hi[PW], s_wi[PW], s_off[PW];   // general path: image, input row / column of this lane's tap, byte offset
  const int tap = qb_ok ? qb / p.C : 0;
  const int cb = qb - tap * p.C;
  const int tr = tap / p.S, ts = tap - tr * p.S;
  const int tr_off = tr - p.pad, ts_off = ts - p.pad;
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int m = step_begin * BR + (wave * PW + j) * RPI + lrow;
    vA[j] = ka_ok ? (unsigned)((m * p.K + ka) * 4) : WG_OOB;
    if (UNIT) {
      vB[j] = qb_ok ? (unsigned)((m * p.C + qb) * 4) : WG_OOB;
    } else {
      const int img = m / (p.Ho * p.Wo);
      const int rem = m - img * p.Ho * p.Wo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      s_img[j] = img;
      s_hi[j] = ho * p.stride + tr_off;
      s_wi[j] = wo * p.stride + ts_off;
      s_off[j] = (((img * p.H + s_hi[j]) * p.W + s_wi[j]) * p.C + cb) * 4;   // used only where (hi, wi) is inside the image
    }
  }
  const unsigned stepA = (unsigned)(BR * p.K * 4), stepB = (unsigned)(BR * p.C * 4);
  // General path: BR pixels further in (img, ho, wo) is one carry per digit; the input position (hi, wi) and the byte offset
  // are carried along with additions only (the first version recomputed the offset from (img, ho, wo): five quarter-rate
  // integer multiplies per DMA instruction, ~540 of a step's 1900 cycles on a 3x3 layer).
  const int d_img = BR / (p.Ho * p.Wo), d_rem = BR - d_img * p.Ho * p.Wo;
  const int d_ho = d_rem / p.Wo, d_wo = d_rem - d_ho * p.Wo;
  const int dwi = d_wo * p.stride, wrap_w = p.Wo * p.stride, dhi = d_ho * p.stride, wrap_h = p.Ho * p.stride;
  const int wlim = wrap_w + ts_off, hlim = wrap_h + tr_off;             // wo == Wo, ho == Ho in (wi, hi) terms
  const int row_b = p.W * p.C * 4, img_b = p.H * row_b;
  const int o_step = dwi * p.C * 4 + dhi * row_b + d_img * img_b;
  const int o_c1 = p.stride * row_b - wrap_w * p.C * 4;                  // wo wraps: one output row down
  const int o_c2 = img_b - wrap_h * row_b;                               // ho wraps: next image

  auto issue = [&](int stage) {
    float* sA = smem + stage * STAGE + wave * PW * 256;
    float* sB = sA + BR * BT;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)(sA + j * 256), 16, (int)vA[j], 0, 0, 0);
      vA[j] += stepA;
    }
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      unsigned off;
      if (UNIT) {
        off = vB[j];
        vB[j] += stepB;
      } else {
        const bool ok = qb_ok && s_img[j] < p.N && (unsigned)s_hi[j] < (unsigned)p.H && (unsigned)s_wi[j] < (unsigned)p.W;
        off = ok ? (unsigned)s_off[j] : WG_OOB;
        const int wi = s_wi[j] + dwi;
        const bool c1 = wi >= wlim;
        s_wi[j] = wi - (c1 ? wrap_w : 0);
        const int hi = s_hi[j] + dhi + (c1 ? p.stride : 0);
        const bool c2 = hi >= hlim;
        s_hi[j] = hi - (c2 ? wrap_h : 0);
        s_img[j] += d_img + (c2 ? 1 : 0);
        s_off[j] += o_step + (c1 ? o_c1 : 0) + (c2 ? o_c2 : 0);
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr)(sB + j * 256), 16, (int)off, 0, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment address of k-iteration 0: pixel row (first pixel of this wave) + lh, column pair 2 * l31 of this wave's columns
  const int wr = wave >> 1, wc = wave & 1;
  const int prow0 = TI == 1 ? wave * 8 : 0;
  const int fa = (prow0 + lh) * BT + (TI == 1 ? 0 : wr * 64) + 2 * l31;
  const int fb = BR * BT + (prow0 + lh) * BT + (TI == 1 ? 0 : wc * 64) + 2 * l31;

#pragma unroll
  for (int s = 0; s < DIST; ++s)
    if (s < nsteps) issue(s);
  int stage = 0, fill = DIST % NST;
  for (int s = 0; s < nsteps; ++s) {
    if (DIST > 1 && s + DIST - 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW * (DIST > 1 ? DIST - 1 : 0)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (s + DIST < nsteps) issue(fill);
    const float* base = smem + stage * STAGE;
    f32x2 a2 = *reinterpret_cast<const f32x2*>(base + fa);
    f32x2 b2 = *reinterpret_cast<const f32x2*>(base + fb);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      f32x2 an = a2, bn = b2;
      if (kk + 1 < KK) {
        an = *reinterpret_cast<const f32x2*>(base + fa + (kk + 1) * 2 * BT);
        bn = *reinterpret_cast<const f32x2*>(base + fb + (kk + 1) * 2 * BT);
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[0], b2[0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[0], b2[1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[1], b2[0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[1], b2[1], acc[1][1], 0, 0, 0);
      a2 = an;
      b2 = bn;
    }
    stage = stage + 1 == NST ? 0 : stage + 1;
    fill = fill + 1 == NST ? 0 : fill + 1;
  }

  // ---- epilogue: accumulators -> LDS (tile row k, 16-byte chunks along q) ----------------------------------------------------
  // D of MFMA (ta, tb): column l31 -> q pair member tb of column pair l31, row i = (r & 3) + 8 (r >> 2) + 4 lh -> k = 2 i + ta
  __syncthreads();
  {
    float* P = TI == 1 ? smem + wave * (64 * 64) : smem + (wr * 64) * BT + wc * 64;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
        *reinterpret_cast<f32x2*>(P + (2 * i + ta) * BT + 2 * l31) = f32x2{acc[ta][0][r], acc[ta][1][r]};
      }
  }
  __syncthreads();
  constexpr int NCH = BT * BT / 4 / 256;   // chunks per thread
  const size_t KQ = (size_t)p.K * p.Q;
  auto emit = [&](int k, int q, f32x4 v) {
    if (p.mode == 0) {
      *reinterpret_cast<f32x4*>(ptarget + (size_t)k * p.Q + q) = v;
    } else if (p.R * p.S == 1 && p.c_real == p.C) {
      f32x4* g = reinterpret_cast<f32x4*>(ptarget + (size_t)k * p.C + q);
      f32x4 o = *g;
      o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3] += v[3];
      *g = o;
    } else {
      const int tp = q / p.C, c = q - tp * p.C;
      const int r = tp / p.S, s2 = tp - r * p.S;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < p.c_real) ptarget[(((size_t)k * p.c_real + c + e) * p.R + r) * p.S + s2] += v[e];
    }
  };
  // Slabs travel between workgroups on different XCDs (one L2 each): they are stored and loaded at AGENT scope (sc1: written
  // through / read past the XCD's L2) instead of bracketing plain accesses with device-scope fences - a release fence writes
  // back and an acquire fence invalidates the WHOLE L2 of the XCD, once per workgroup (measured: 24 splits of layer3's
  // 1x1 took 202 us with the fences, the unsplit launch 52 us).
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)p.splits * KQ * 4), 0x00020000);
  const unsigned zoff = (unsigned)((size_t)blockIdx.z * KQ * 4);
#pragma unroll
  for (int u = 0; u < NCH; ++u) {
    const int c = t + 256 * u;
    const int kl = c / CH, qc = c - kl * CH;
    f32x4 v;
    if (TI == 1) {
      const float* P = smem + kl * 64 + qc * 4;
      v = *reinterpret_cast<const f32x4*>(P);
#pragma unroll
      for (int w2 = 1; w2 < 4; ++w2) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(P + w2 * (64 * 64));
        v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
      }
    } else {
      v = *reinterpret_cast<const f32x4*>(smem + kl * BT + qc * 4);
    }
    const int k = k0 + kl, q = q0 + qc * 4;
    if (k < p.K && q < p.Q) {
      if (p.splits == 1) emit(k, q, v);
      else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rS, zoff + (unsigned)((k * p.Q + q) * 4), 0, 16);
    }
  }
  if (p.splits == 1) return;
  // ---- last arriver of this tile adds the slabs in z order --------------------------------------------------------------------
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's slab stores have reached the coherence point
  __syncthreads();
  if (t == 0) {
    int* ctr = p.counters + blockIdx.x;
    const int old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old == p.splits - 1;
    if (s_last) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next launch
  }
  __syncthreads();
  if (!s_last) return;
#pragma unroll 1
  for (int u = 0; u < NCH; ++u) {
    const int c = t + 256 * u;
    const int kl = c / CH, qc = c - kl * CH;
    const int k = k0 + kl, q = q0 + qc * 4;
    if (k >= p.K || q >= p.Q) continue;
    const unsigned off = (unsigned)((k * p.Q + q) * 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    // eight slab loads in flight at a time; added in z order
    for (int z0 = 0; z0 < p.splits; z0 += 8) {
      f32x4 o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (z0 + i < p.splits)
          o[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rS, off + (unsigned)((size_t)(z0 + i) * KQ * 4), 0, 16));
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (z0 + i < p.splits) {
          if (z0 + i == 0) v = o[i];
          else { v[0] += o[i][0]; v[1] += o[i][1]; v[2] += o[i][2]; v[3] += o[i][3]; }
        }
    }
    emit(k, q, v);
  }
}

template <int TI, bool UNIT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_dma_f32(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float wg_smem[];
  conv_wgrad_dma_body<TI, UNIT>(p, p.x, p.dy, p.target, wg_smem);
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

template <int TI, bool UNIT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_dma_grouped_f32(const WgradParams p, const WgradGroups g, const WgradOuts o) {
  extern __shared__ __attribute__((aligned(16))) float wg_smem[];
  const int grp = blockIdx.y;
  conv_wgrad_dma_body<TI, UNIT>(p, g.x[grp], g.dy[grp], o.grad[grp], wg_smem);
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

// ti: 1 / 2 = conv_wgrad_f32<1 / 2> (+ reduction / accumulation kernels), 3 / 4 = conv_wgrad_dma_f32<1 / 2> (64 / 128 tile,
// reduction and accumulation in its own epilogue)
struct WgradPlan {
  int ti, splits;
};
inline int tile_factor(int ti) { return ti >= 3 ? ti - 2 : ti; }
inline int tiles_for(int k, int q, int ti) {
  const int bt = 64 * tile_factor(ti);
  return ((k + bt - 1) / bt) * ((q + bt - 1) / bt);
}
std::atomic<int> g_wgrad_variant{0};   // frcnn_conv2d_wgrad_set_variant: 0 = every kernel, 1 = conv_wgrad_f32 only, 2 = DMA only
std::atomic<int> g_wgrad_force_ti{0}, g_wgrad_force_splits{0};   // frcnn_conv2d_wgrad_set_plan (tests): 0 = not forced

// Default plan: the 128 x 128 tile and the modelled split; in autotune mode (frcnn_conv2d_set_autotune, shared with the
// forward kernel) the first call of a shape times both tile sizes around the modelled split and caches the fastest.
typedef std::array<int, 9> WgradKey;
std::map<WgradKey, WgradPlan> g_wgrad_plans;
std::mutex g_wgrad_mutex;

// dma: the LDS-DMA kernels may be used (operands within a 2 GB buffer; a tile counter per tile at hand when splits > 1)
std::vector<WgradPlan> wgrad_candidates(int k, int q, int steps, bool dma, bool counters) {
  std::vector<WgradPlan> out;
  const int variant = g_wgrad_variant.load();
  for (int ti = 4; ti >= 1; --ti) {
    if (ti >= 3 && (!dma || variant == 1)) continue;
    if (ti <= 2 && variant == 2 && dma) continue;
    const int base = choose_splits(tiles_for(k, q, ti), steps, tile_factor(ti));
    for (int sp : {base, std::max(1, base / 2), std::min(64, base * 2), 1}) {
      const int sps = (steps + sp - 1) / sp;
      const int real = (steps + sps - 1) / sps;
      bool dup = false;
      for (const WgradPlan& c : out) dup = dup || (c.ti == ti && c.splits == real);
      if (ti >= 3 && real > 1 && !counters) continue;
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

// Shape -> kernel parameters (plan-independent part); false when a tensor is too large for the 32-bit indices
bool fill_params(WgradParams* p, const float* x, const float* dy, int n, int h, int w, int c, int k, int r, int s, int stride,
                 int pad) {
  p->x = x; p->dy = dy; p->out = nullptr; p->target = nullptr; p->counters = nullptr;
  p->H = h; p->W = w; p->C = c; p->K = k; p->R = r; p->S = s; p->stride = stride; p->pad = pad;
  p->Ho = (h + 2 * pad - r) / stride + 1;
  p->Wo = (w + 2 * pad - s) / stride + 1;
  const long M = (long)n * p->Ho * p->Wo;
  if (M * (long)k >= (1L << 31) || (long)n * h * w * c >= (1L << 31)) return false;
  p->M = (int)M;
  p->N = n;
  p->Q = r * s * c;
  p->steps = (p->M + BR - 1) / BR;
  p->steps_per_split = p->steps;
  p->tiles_k = p->tiles_q = 0;
  p->mode = 0; p->c_real = c; p->splits = 1;
  p->xbytes = (unsigned)std::min<long>((long)n * h * w * c * 4, 0x7fffffffL);
  p->dybytes = (unsigned)std::min<long>(M * k * 4, 0x7fffffffL);
  return true;
}
// conv_wgrad_dma_f32 addresses its operands as byte offsets into 2 GB buffers (rows past M included: they must not wrap)
bool dma_ok(const WgradParams& p) {
  const long lim = (1L << 31) - 64;
  return ((long)p.M + 2 * BR) * p.K * 4 < lim && ((long)p.M + 2 * BR) * p.C * 4 < lim &&
         (long)p.N * p.H * p.W * p.C * 4 < lim;
}
inline bool unit_conv(const WgradParams& p) { return p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0; }

template <int TI, bool UNIT>
int launch_dma(const WgradParams& p, dim3 grid, hipStream_t stream) {
  constexpr size_t lds = (size_t)WG_LDS_FLOATS * sizeof(float);
  static std::atomic<bool> configured{false};   // idempotent attribute call: a race only repeats it
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_dma_f32<TI, UNIT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return frcnn::fail(FRCNN_ERR_LAUNCH, "conv_wgrad_dma_f32: set LDS size: %s", hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL((conv_wgrad_dma_f32<TI, UNIT>), grid, dim3(256), lds, stream, p);
  return frcnn::check_launch("conv_wgrad_dma_f32");
}
template <int TI, bool UNIT>
int launch_dma_grouped(const WgradParams& p, dim3 grid, const WgradGroups& g, const WgradOuts& o, hipStream_t stream) {
  constexpr size_t lds = (size_t)WG_LDS_FLOATS * sizeof(float);
  static std::atomic<bool> configured{false};
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_dma_grouped_f32<TI, UNIT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return frcnn::fail(FRCNN_ERR_LAUNCH, "conv_wgrad_dma_grouped_f32: set LDS size: %s", hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL((conv_wgrad_dma_grouped_f32<TI, UNIT>), grid, dim3(256), lds, stream, p, g, o);
  return frcnn::check_launch("conv_wgrad_dma_grouped_f32");
}

// One plan, start to finish.  mode 0: target = dw [K][R][S][C], overwritten; mode 1: target = the parameter's gradient
// (K, c_real, R, S), accumulated into.  Plans 1 / 2 need the reduction (mode 0, splits > 1) or accumulation (mode 1) kernel
// behind the main kernel; plans 3 / 4 finish in their own epilogue.
int run_wgrad(WgradParams p, const WgradPlan& pl, int mode, float* target, int c_real, void* ws, int* counters,
              hipStream_t stream) {
  const int bt = 64 * tile_factor(pl.ti);
  p.tiles_k = (p.K + bt - 1) / bt;
  p.tiles_q = (p.Q + bt - 1) / bt;
  p.steps_per_split = (p.steps + pl.splits - 1) / pl.splits;
  p.splits = pl.splits;
  p.mode = mode;
  p.c_real = c_real;
  const dim3 grid(p.tiles_k * p.tiles_q, 1, pl.splits);
  if (pl.ti >= 3) {
    if (pl.splits > 1 && !counters) return frcnn::fail(FRCNN_ERR_ARG, "conv_wgrad_dma_f32: split plan without tile counters");
    if ((size_t)pl.splits * p.K * p.Q * sizeof(float) >= ((size_t)1 << 31))
      return frcnn::fail(FRCNN_ERR_ARG, "conv_wgrad_dma_f32: %d slabs of %d x %d floats exceed a 2 GB buffer", pl.splits, p.K, p.Q);
    p.out = static_cast<float*>(ws);
    p.target = target;
    p.counters = counters;
    const bool unit = unit_conv(p);
    if (pl.ti == 3) return unit ? launch_dma<1, true>(p, grid, stream) : launch_dma<1, false>(p, grid, stream);
    return unit ? launch_dma<2, true>(p, grid, stream) : launch_dma<2, false>(p, grid, stream);
  }
  p.out = (pl.splits > 1 || mode == 1) ? static_cast<float*>(ws) : target;
  if (pl.ti == 2) hipLaunchKernelGGL(conv_wgrad_f32<2>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(conv_wgrad_f32<1>, grid, dim3(256), 0, stream, p);
  int rc = frcnn::check_launch("conv_wgrad_f32");
  if (rc != FRCNN_OK) return rc;
  if (mode == 1) {
    const size_t total = (size_t)p.K * c_real * p.R * p.S;
    hipLaunchKernelGGL(wgrad_accumulate_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 8192)), dim3(256), 0,
                       stream, static_cast<const float*>(ws), pl.splits, p.K, p.R, p.S, p.C, c_real, target);
    return frcnn::check_launch("wgrad_accumulate_kernel");
  }
  if (pl.splits > 1) {
    const size_t n4 = (size_t)p.K * p.Q / 4;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 4096)), dim3(256), 0,
                       stream, static_cast<const float*>(ws), pl.splits, n4, target);
    rc = frcnn::check_launch("wgrad_reduce_kernel");
  }
  return rc;
}

// plan without tuning: the 128 x 128 tile and the modelled split - the DMA kernel when it may be used
WgradPlan default_plan(const WgradParams& p, bool counters) {
  const int variant = g_wgrad_variant.load();
  const int sp2 = choose_splits(tiles_for(p.K, p.Q, 2), p.steps, 2);
  if (variant != 1 && dma_ok(p) && (sp2 == 1 || counters)) return WgradPlan{4, sp2};
  return WgradPlan{2, sp2};
}
// a cached plan is usable for this call (the caller may have no counters; the variant switch may have changed)
bool plan_usable(const WgradPlan& pl, const WgradParams& p, bool counters) {
  const int variant = g_wgrad_variant.load();
  if (pl.ti >= 3) return variant != 1 && dma_ok(p) && (pl.splits == 1 || counters);
  return variant != 2 || !dma_ok(p);
}

// Time every candidate twice on the caller's tensors (best-of), outside stream capture only.
bool tune_wgrad(const WgradParams& p, float* dw, void* ws, size_t ws_avail, int* counters, hipStream_t stream,
                WgradPlan* best) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return false;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess) return false;
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return false; }
  const std::vector<WgradPlan> cands = wgrad_candidates(p.K, p.Q, p.steps, dma_ok(p), counters != nullptr);
  std::vector<float> best_of(cands.size(), 1e30f);
  for (int pass = 0; pass < 2; ++pass)
    for (size_t ci = 0; ci < cands.size(); ++ci) {
      if (slab_bytes_of(cands[ci], p.K, p.Q) > ws_avail) continue;
      if (pass == 0 && run_wgrad(p, cands[ci], 0, dw, p.C, ws, counters, stream) != FRCNN_OK) continue;   // warm-up
      (void)hipEventRecord(e0, stream);
      bool ok = true;
      for (int i = 0; i < 3 && ok; ++i) ok = run_wgrad(p, cands[ci], 0, dw, p.C, ws, counters, stream) == FRCNN_OK;
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

int bias_gradient(const float* dy, int M, int k, float* partial, float* db, bool accumulate, hipStream_t stream) {
  hipLaunchKernelGGL(bias_grad_partial_kernel, dim3((k + 63) / 64, BIAS_GROUPS), dim3(256), 0, stream, dy, M, k, partial);
  int rc = frcnn::check_launch("bias_grad_partial_kernel");
  if (rc != FRCNN_OK) return rc;
  if (accumulate) {
    hipLaunchKernelGGL(bias_grad_accumulate_kernel, dim3((k + 255) / 256), dim3(256), 0, stream, partial, k, BIAS_GROUPS, db);
    return frcnn::check_launch("bias_grad_accumulate_kernel");
  }
  hipLaunchKernelGGL(bias_grad_final_kernel, dim3((k + 255) / 256), dim3(256), 0, stream, partial, k, db);
  return frcnn::check_launch("bias_grad_final_kernel");
}

}  // namespace

unsigned long long frcnn::wgrad_settings_word() {
  return (unsigned long long)g_wgrad_variant.load() | ((unsigned long long)g_wgrad_force_ti.load() << 4) |
         ((unsigned long long)g_wgrad_force_splits.load() << 8);
}

extern "C" int frcnn_conv2d_wgrad_set_plan(int kernel, int splits) {
  if (kernel < 0 || kernel > 4 || splits < 0 || splits > 64 || (kernel > 0 && splits < 1))
    return frcnn::fail(FRCNN_ERR_ARG, "conv2d_wgrad_set_plan: kernel 0 (not forced) or 1..4, 1..64 pixel splits");
  g_wgrad_force_ti.store(kernel);
  g_wgrad_force_splits.store(kernel ? splits : 0);
  return FRCNN_OK;
}

extern "C" int frcnn_conv2d_wgrad_set_variant(int variant) {
  if (variant < 0 || variant > 2)
    return frcnn::fail(FRCNN_ERR_ARG, "conv2d_wgrad_set_variant: 0 (every kernel), 1 (conv_wgrad_f32 only), 2 (conv_wgrad_dma_f32 where it applies)");
  g_wgrad_variant.store(variant);
  frcnn::clear_wgrad_plans();
  return FRCNN_OK;
}

extern "C" int frcnn_conv2d_bwd_weight_counters(int c, int k, int r, int s) {
  if (c <= 0 || k <= 0 || r <= 0 || s <= 0) return 0;
  return tiles_for(k, r * s * c, 1);
}

extern "C" size_t frcnn_conv2d_bwd_weight_ws_bytes(int n, int h, int w, int c, int k, int r, int s, int stride,
                                                   int pad) {
  if (!wgrad_args_ok(n, h, w, c, k, r, s, stride, pad)) return 0;
  WgradParams p;
  if (!fill_params(&p, nullptr, nullptr, n, h, w, c, k, r, s, stride, pad)) return 0;
  WgradPlan pl;
  size_t slabs = 0;
  if (lookup_wgrad(wgrad_key(n, h, w, c, k, r, s, stride, pad), &pl)) slabs = slab_bytes_of(pl, k, p.Q);
  // room for whatever this call may still choose: the largest candidate of a shape that is about to be tuned, the default
  // plans (with and without tile counters) otherwise and in case the cached plan is not usable by the caller
  if (frcnn::autotune_enabled())
    for (const WgradPlan& cand : wgrad_candidates(k, p.Q, p.steps, dma_ok(p), true)) slabs = std::max(slabs, slab_bytes_of(cand, k, p.Q));
  slabs = std::max(slabs, slab_bytes_of(default_plan(p, true), k, p.Q));
  slabs = std::max(slabs, slab_bytes_of(default_plan(p, false), k, p.Q));
  if (g_wgrad_force_ti.load()) slabs = std::max(slabs, (size_t)std::max(1, std::min(g_wgrad_force_splits.load(), p.steps)) * k * p.Q * sizeof(float));
  // at least one slab (the accumulating form of plans 1 / 2 reduces from it) + the bias-gradient partials
  slabs = std::max(slabs, (size_t)k * p.Q * sizeof(float));
  return frcnn::align_up(slabs, 256) + (size_t)BIAS_GROUPS * k * sizeof(float);
}

namespace {
// shared body of frcnn_conv2d_bwd_weight (mode 0) and frcnn_conv2d_bwd_weight_acc (mode 1)
int bwd_weight(const char* who, int mode, const float* x, const float* dy, float* target, int c_real, float* bias_target, int n,
               int h, int w, int c, int k, int r, int s, int stride, int pad, void* ws, size_t ws_bytes, int* counters,
               hipStream_t stream) {
  WgradParams p;
  if (!fill_params(&p, x, dy, n, h, w, c, k, r, s, stride, pad)) return frcnn::fail(FRCNN_ERR_ARG, "%s: tensor too large", who);
  const size_t bias_bytes = bias_target ? (size_t)BIAS_GROUPS * k * sizeof(float) : 0;
  const size_t slab_room = ws_bytes > bias_bytes ? ((ws_bytes - bias_bytes) / 256) * 256 : 0;   // slabs first, 256-aligned
  const WgradKey key = wgrad_key(n, h, w, c, k, r, s, stride, pad);
  WgradPlan pl;
  bool have = lookup_wgrad(key, &pl) && plan_usable(pl, p, counters != nullptr);
  if (const int fti = g_wgrad_force_ti.load()) {
    pl = WgradPlan{fti, std::max(1, std::min(g_wgrad_force_splits.load(), p.steps))};
    if (pl.ti >= 3 && (!dma_ok(p) || (pl.splits > 1 && !counters)))
      return frcnn::fail(FRCNN_ERR_ARG, "%s: the forced plan (%d, %d) does not apply to this call", who, pl.ti, pl.splits);
    have = true;
  }
  if (!have && mode == 0 && frcnn::autotune_enabled() && ws && tune_wgrad(p, target, ws, slab_room, counters, stream, &pl)) {
    std::lock_guard<std::mutex> lock(g_wgrad_mutex);
    g_wgrad_plans[key] = pl;
    have = true;
  }
  if (!have) pl = default_plan(p, counters != nullptr);
  size_t slab_bytes = slab_bytes_of(pl, k, p.Q);
  if (mode == 1 && pl.ti <= 2) slab_bytes = std::max(slab_bytes, (size_t)k * p.Q * sizeof(float));
  slab_bytes = frcnn::align_up(slab_bytes, 256);
  const size_t need = slab_bytes + bias_bytes;
  if (need > 0 && (!ws || ws_bytes < need)) return frcnn::fail(FRCNN_ERR_WS, "%s: workspace %zu < %zu bytes", who, ws_bytes, need);
  int rc = run_wgrad(p, pl, mode, target, c_real, ws, counters, stream);
  if (rc != FRCNN_OK || !bias_target) return rc;
  return bias_gradient(dy, p.M, k, reinterpret_cast<float*>(static_cast<char*>(ws) + slab_bytes), bias_target, mode == 1, stream);
}
}  // namespace

extern "C" int frcnn_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* db, int n, int h, int w,
                                       int c, int k, int r, int s, int stride, int pad, void* ws, size_t ws_bytes,
                                       int* counters, void* stream_) {
  FRCNN_REQUIRE(x && dy && dw, "conv2d_bwd_weight: null tensor");
  FRCNN_REQUIRE(wgrad_args_ok(n, h, w, c, k, r, s, stride, pad),
                "conv2d_bwd_weight: bad shape n=%d h=%d w=%d c=%d k=%d r=%d s=%d stride=%d pad=%d (need c%%4==0, k%%4==0)",
                n, h, w, c, k, r, s, stride, pad);
  return bwd_weight("conv2d_bwd_weight", 0, x, dy, dw, c, db, n, h, w, c, k, r, s, stride, pad, ws, ws_bytes, counters,
                    static_cast<hipStream_t>(stream_));
}

extern "C" int frcnn_conv2d_bwd_weight_acc(const float* x, const float* dy, float* grad_w, int c_real, float* grad_b, int n,
                                           int h, int w, int c, int k, int r, int s, int stride, int pad, void* ws,
                                           size_t ws_bytes, int* counters, void* stream_) {
  FRCNN_REQUIRE(x && dy && grad_w && c_real > 0 && c_real <= c, "conv2d_bwd_weight_acc: null tensor or c_real out of range");
  FRCNN_REQUIRE(wgrad_args_ok(n, h, w, c, k, r, s, stride, pad),
                "conv2d_bwd_weight_acc: bad shape n=%d h=%d w=%d c=%d k=%d r=%d s=%d stride=%d pad=%d (need c%%4==0, k%%4==0)",
                n, h, w, c, k, r, s, stride, pad);
  return bwd_weight("conv2d_bwd_weight_acc", 1, x, dy, grad_w, c_real, grad_b, n, h, w, c, k, r, s, stride, pad, ws, ws_bytes,
                    counters, static_cast<hipStream_t>(stream_));
}

// Filter gradients of `groups` convolutions of identical shape in one launch: grad_w[g] (K, c_real, R, S) += dW(x[g], dy[g]).
// x / dy / grad_w are HOST arrays of device pointers.  No pixel split, fixed summation order: deterministic.  The LDS-DMA kernel
// adds into the gradients itself; conv_wgrad_grouped_f32 (variant 1, or operands beyond 2 GB) goes through `ws` and
// wgrad_accumulate_grouped_kernel.
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
  FRCNN_REQUIRE(fill_params(&p, nullptr, nullptr, n, h, w, c, k, r, s, stride, pad), "conv2d_bwd_weight_acc_grouped: tensor too large");
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
  if (g_wgrad_variant.load() != 1 && dma_ok(p)) {
    p.mode = 1; p.c_real = c_real; p.splits = 1;
    const bool unit = unit_conv(p);
    if (ti == 1) return unit ? launch_dma_grouped<1, true>(p, grid, g, o, stream) : launch_dma_grouped<1, false>(p, grid, g, o, stream);
    return unit ? launch_dma_grouped<2, true>(p, grid, g, o, stream) : launch_dma_grouped<2, false>(p, grid, g, o, stream);
  }
  const size_t need = frcnn_conv2d_bwd_weight_acc_grouped_ws_bytes(groups, c, k, r, s);
  if (!ws || ws_bytes < need)
    return frcnn::fail(FRCNN_ERR_WS, "conv2d_bwd_weight_acc_grouped: workspace %zu < %zu bytes", ws_bytes, need);
  p.out = static_cast<float*>(ws);
  if (ti == 2) hipLaunchKernelGGL(conv_wgrad_grouped_f32<2>, grid, dim3(256), 0, stream, p, g);
  else hipLaunchKernelGGL(conv_wgrad_grouped_f32<1>, grid, dim3(256), 0, stream, p, g);
  int rc = frcnn::check_launch("conv_wgrad_grouped_f32");
  if (rc != FRCNN_OK) return rc;
  const size_t total = (size_t)k * c_real * r * s;
  hipLaunchKernelGGL(wgrad_accumulate_grouped_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 2048), groups),
                     dim3(256), 0, stream, static_cast<const float*>(ws), k, r, s, c, c_real, o);
  return frcnn::check_launch("wgrad_accumulate_grouped_kernel");
}
