// Implicit-GEMM convolution forward on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces the ATen/cuDNN conv2d + batch_norm(eval) + add + relu chain of the reference's
// lib/nets/resnet.py:98-127 (Bottleneck.forward), :152-156 (stem), lib/nets/fpn.py:33-39 and the RPN
// convs of the (missing) lib/nets/network.py.
//
// GEMM view:  M = N*Ho*Wo output pixels, N = K output channels, Kdim = R*S*C (tap-major, channel-minor).
// Activations are NHWC and filters KRSC, so both operands are contiguous along Kdim: the global->LDS
// staging moves 16-byte chunks along Kdim and the MFMA fragments are ds_read_b128 along Kdim.
//
// Workgroup = WM x WN waves, each wave owns TM x TN tiles of 32x32 outputs (block tile BM = 32*TM*WM
// pixels, BN = 32*TN*WN channels), BK = 32.  The large-GEMM configuration is 8 waves (two per SIMD)
// on a 256x128 tile, ONE workgroup per CU: both waves of a SIMD belong to the same workgroup, so they
// reach the per-K-step barrier together and the SIMDs carry identical work (two co-resident 4-wave
// workgroups couple through their barriers instead and leave ~20 % of the MFMA slots empty).
// LDS rows are padded to 36 floats: the 16-lane groups of ds_read_b128 then hit 16 distinct 16-byte
// slots (36*m mod 64 = 4*(9m mod 16) is a bijection).
//
// Software pipeline of one K-step (4 groups of k=8, each TM*TN*4 MFMAs per wave):
//   group 0,1 : MFMAs on fragments prefetched one group ahead
//   then      : the register-staged global tile of step s+1 is written to the other LDS buffer
//   group 2   : MFMAs; the global loads of step s+2 are issued (they land ~3/4 step later)
//   barrier   : placed BEFORE the last group, whose fragments are already in registers
//   group 3   : MFMAs, overlapped with the first fragment reads of the next buffer
// so neither the LDS write/barrier/read turn-around nor the HBM/L2 latency is exposed.
//
// The MFMA computes an exact k-ordered fp32 fma chain, so results are deterministic and independent
// of the tile configuration for split_k == 1.
#include "common.h"

#include <atomic>
#include <type_traits>
#include <hip/hip_ext.h>

#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <vector>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDS_PITCH = 36;  // floats per LDS row (32 + 4 pad)
constexpr int NUM_CU = 256;

struct ConvParams {
  const float* x;
  const float* w;
  const float* scale;
  const float* shift;
  const float* res;
  float* y;
  float* partial;  // split-K slabs [splits][M][K] or nullptr
  int H, W, C, K, R, S, stride, pad, Ho, Wo;
  int M;     // N*Ho*Wo
  int Ktot;  // R*S*C
  int ksteps;
  int steps_per_split;
  int tiles_m, tiles_n;
  int relu;
  int ys, Hy, Wy;  // output pixel (ho, wo) is written at (ho*ys, wo*ys) of an Hy x Wy map (ys = 1: dense)
  // grouped launch (blockIdx.y = group): element offsets of a group's activations / filter / output.  Used by the
  // Winograd path (16 independent GEMMs in one launch); 0 for an ordinary convolution (gridDim.y = 1).
  size_t gx, gw, gy;
  // optional activation-backward epilogue (data-gradient calls): y = mask[m][n] > 0 ? y * mscale[n] : 0 - the ReLU /
  // folded-BatchNorm backward of the layer BELOW, applied to this layer's input gradient before it is stored
  const float* mask;
  const float* mscale;
  const float* u_pre;   // host side: Winograd-transformed filter supplied by the caller (frcnn_conv2d_fwd_pre) or nullptr
  // Winograd grouped GEMM with the INPUT TRANSFORM fused into the A-tile load (conv_igemm_f32<..., WINO = true>): x is the
  // layer's NHWC input (wiH x wiW pixels, C channels), GEMM row m is the 2x2-output tile (n, ty, tx) of a wth x wtw grid and
  // blockIdx.y the transform component
  int wiH, wiW, wth, wtw;
  int epi_lds;   // 1: the register-staged kernels transpose their accumulator tiles through LDS before storing (conv_epilogue_lds)
  unsigned xbytes, wbytes;   // byte range of (a group's) activations / filter for the buffer-load kernel (0: range >= 2 GB, kernel not usable)
  const float* zero;   // device address of g_zero_page (resolved once on the host: a kernel argument costs no s_getpc / s_load in the K loop)
};

// XCD-aware bijective remap (guide T1): blocks b and b+8 share an XCD; give each XCD a contiguous
// range of logical tiles so neighbouring tiles (shared A rows / B columns) hit the same L2.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

// Logical tile -> (tile_m, tile_n).  Tiles are walked in column groups of RASTER_GN n-tiles (m fastest inside a
// group): the tiles an XCD works on concurrently (a contiguous range of this order, see xcd_remap) then share at most
// RASTER_GN filter column blocks, so the filter slice they re-read stays inside the XCD's 4 MB L2 instead of the whole
// K x C filter cycling through it once per pixel row.
constexpr int RASTER_GN = 8;
__device__ __forceinline__ void tile_coords(int tile, int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
  const int per_group = RASTER_GN * tiles_m;
  const int group = tile / per_group, within = tile - group * per_group;
  const int gn = min(RASTER_GN, tiles_n - group * RASTER_GN);
  tile_n = group * RASTER_GN + within % gn;
  tile_m = within / gn;
}

// Epilogue shared by the conv kernels.  The MFMA operands are (weights, activations), so D has the PIXEL on the
// lane (col = lane&31) and the CHANNEL in the registers: row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Registers
// 4g..4g+3 are four consecutive channels -> every access is a 16-byte vector per lane, and all residual loads of a
// 32x32 tile are issued before its first store.
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wr, int wc,
                                              int lane, int gy = -1, int gz = -1) {
  if (gy < 0) gy = blockIdx.y;                     // (the persistent kernel walks groups / splits itself)
  if (gz < 0) gz = blockIdx.z;
  const size_t goff = (size_t)gy * p.gy;   // grouped launches have neither a residual nor split-K slabs
  // The MFMA operands are (weights, activations), so D has the PIXEL on the lane (col = lane&31) and
  // the CHANNEL in the registers: row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Registers 4g..4g+3 are four
  // consecutive channels -> every access of the epilogue is a 16-byte vector per lane, and all
  // residual loads of a 32x32 tile are issued before its first store (res may alias nothing we
  // write, but the compiler cannot know: batching keeps 4 loads in flight instead of a
  // load->wait->store chain per element).
  const int mlane = lane & 31, nhalf = 4 * (lane >> 5);
  const bool vec = (p.K & 3) == 0;
  float* const slab = p.partial ? p.partial + (size_t)gz * p.M * p.K : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + (wr * TM + i) * 32 + mlane;
    if (m >= p.M) continue;
    size_t orow = (size_t)m * p.K + goff;  // row of y / residual
    if (p.ys != 1) {
      const int img = m / (p.Ho * p.Wo);
      const int rem = m - img * p.Ho * p.Wo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      orow = ((size_t)(img * p.Hy + ho * p.ys) * p.Wy + wo * p.ys) * p.K + goff;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nb = n0 + (wc * TN + j) * 32 + nhalf;
      const size_t row = (size_t)m * p.K;
      if (vec) {
        if (slab) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = nb + 8 * g;
            if (n < p.K)
              *reinterpret_cast<f32x4*>(slab + row + n) =
                  f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          }
          continue;
        }
        f32x4 rv[4], mv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nb + 8 * g;
          rv[g] = (p.res && n < p.K) ? *reinterpret_cast<const f32x4*>(p.res + orow + n) : f32x4{0.f, 0.f, 0.f, 0.f};
          mv[g] = (p.mask && n < p.K) ? *reinterpret_cast<const f32x4*>(p.mask + orow + n) : f32x4{1.f, 1.f, 1.f, 1.f};
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nb + 8 * g;
          if (n >= p.K) continue;
          const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
          const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
          const f32x4 ms = (p.mask && p.mscale) ? *reinterpret_cast<const f32x4*>(p.mscale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = acc[i][j][4 * g + e] * sc[e] + sh[e];
            t += rv[g][e];
            t = p.relu ? fmaxf(t, 0.f) : t;
            if (p.mask) t = mv[g][e] > 0.f ? t * ms[e] : 0.f;
            v[e] = t;
          }
          *reinterpret_cast<f32x4*>(p.y + orow + n) = v;
        }
      } else {
        // K % 4 != 0: rows are not 16-byte aligned, element-wise accesses
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = nb + (r & 3) + 8 * (r >> 2);
          if (n >= p.K) continue;
          if (slab) { slab[row + n] = acc[i][j][r]; continue; }
          float t = acc[i][j][r] * (p.scale ? p.scale[n] : 1.f) + (p.shift ? p.shift[n] : 0.f);
          if (p.res) t += p.res[orow + n];
          t = p.relu ? fmaxf(t, 0.f) : t;
          if (p.mask) t = p.mask[orow + n] > 0.f ? t * (p.mscale ? p.mscale[n] : 1.f) : 0.f;
          p.y[orow + n] = t;
        }
      }
    }
  }
}

// The same epilogue with the accumulator tiles TRANSPOSED through LDS first.  In the MFMA layout a lane holds 4 consecutive
// channels of ITS pixel, so a store instruction of conv_epilogue writes 64 pieces of 16 bytes (32 pixels x 2 halves), each in
// a different row of the output - and reads the residual the same way.  Here a wave writes its 32 x 32 tile to a private LDS
// patch (the K loop is over: the staging buffers are free) and reads it back with 8 lanes per pixel: every global access of
// the wave is then 8 rows x 128 contiguous bytes.  Same per-element arithmetic -> bit-identical results.  K % 4 == 0 only.
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue_lds(const ConvParams& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wr, int wc,
                                                  int lane, float* __restrict__ patch, int gy = -1, int gz = -1) {
  if (gy < 0) gy = blockIdx.y;
  if (gz < 0) gz = blockIdx.z;
  const size_t goff = (size_t)gy * p.gy;
  float* const slab = p.partial ? p.partial + (size_t)gz * p.M * p.K : nullptr;
  const int wpix = lane & 31, whalf = 4 * (lane >> 5);       // write phase: MFMA layout
  const int rrow = lane >> 3, rcol = 4 * (lane & 7);         // read phase: 8 lanes x 16 B per pixel row
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(patch + wpix * LDS_PITCH + 8 * g + whalf) =
            f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
      const int n = n0 + (wc * TN + j) * 32 + rcol;
      const bool nok = n < p.K;
      f32x4 v[4], rv[4], mv[4];
      size_t at[4];
      bool ok[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = rrow + 8 * it;
        const int m = m0 + (wr * TM + i) * 32 + row;
        v[it] = *reinterpret_cast<const f32x4*>(patch + row * LDS_PITCH + rcol);
        ok[it] = nok && m < p.M;
        size_t orow = (size_t)m * p.K + goff;
        if (p.ys != 1 && ok[it]) {
          const int img = m / (p.Ho * p.Wo);
          const int rem = m - img * p.Ho * p.Wo;
          const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
          orow = ((size_t)(img * p.Hy + ho * p.ys) * p.Wy + wo * p.ys) * p.K + goff;
        }
        at[it] = slab ? (size_t)m * p.K + n : orow + n;
        rv[it] = (!slab && p.res && ok[it]) ? *reinterpret_cast<const f32x4*>(p.res + at[it]) : f32x4{0.f, 0.f, 0.f, 0.f};
        mv[it] = (!slab && p.mask && ok[it]) ? *reinterpret_cast<const f32x4*>(p.mask + at[it]) : f32x4{1.f, 1.f, 1.f, 1.f};
      }
      if (slab) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
          if (ok[it]) *reinterpret_cast<f32x4*>(slab + at[it]) = v[it];
        continue;
      }
      const f32x4 sc = (p.scale && nok) ? *reinterpret_cast<const f32x4*>(p.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 sh = (p.shift && nok) ? *reinterpret_cast<const f32x4*>(p.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 ms = (p.mask && p.mscale && nok) ? *reinterpret_cast<const f32x4*>(p.mscale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        if (!ok[it]) continue;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = v[it][e] * sc[e] + sh[e];
          t += rv[it][e];
          t = p.relu ? fmaxf(t, 0.f) : t;
          if (p.mask) t = mv[it][e] > 0.f ? t * ms[e] : 0.f;
          o[e] = t;
        }
        *reinterpret_cast<f32x4*>(p.y + at[it]) = o;
      }
    }
  }
}

// 256 zero bytes: out-of-range lanes (padding taps, M / K tails) load from here instead of branching around the load, so
// every load of a tile is issued unconditionally (and the compiler can count them: partial vmcnt waits)
__device__ float g_zero_page[64];

// (the one-accumulator wave of the 64x64 tile is held to 128 VGPRs: four workgroups per CU, as its 36 KB of LDS allow)
template <int WM, int WN, int TM, int TN, bool ALIGNED, bool WINO = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_igemm_f32(const ConvParams p) {
  static_assert(!WINO || ALIGNED, "the fused Winograd input transform needs C % 32 == 0");
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int RPP = NT / 8;                // tile rows staged per pass (8 threads x 16 B cover one row)
  constexpr int PA = BM / RPP, PB = BN / RPP;  // 16-byte chunks per thread per K-step
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                        // [2][BM][36]
  float* Bs = smem + 2 * BM * LDS_PITCH;   // [2][BN][36]

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const float* const px = p.x + (size_t)blockIdx.y * p.gx;
  const float* const pw = p.w + (size_t)blockIdx.y * p.gw;

  const int ntiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  int tile_m, tile_n;
  tile_coords(tile, p.tiles_m, p.tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int step_begin = blockIdx.z * p.steps_per_split;
  const int step_end = min(step_begin + p.steps_per_split, p.ksteps);

  // ---- per-thread staging geometry: thread t moves chunk kc of rows (t>>3) + RPP*i ---------------
  const int kc = t & 7, row0 = t >> 3;
  int a_base[PA], a_hi0[PA], a_wi0[PA];
  // WINO: V[xi][tile][c] = (B^T d B)[ci][cj] of the tile's 4x4 input patch d (rows 2ty-1.., cols 2tx-1.., zero outside the
  // map) is a signed sum of FOUR patch pixels: with B^T's rows (1,0,-1,0), (0,1,1,0), (0,-1,1,0), (0,1,0,-1) row ci combines
  // patch rows ra, rb as d[ra] + sr d[rb] and column cj likewise, in the order wino_input_kernel evaluates them
  // (rows first, then columns), so the fused GEMM is bit-identical to the two-launch form.
  constexpr int WPA = WINO ? PA : 1, WQ = WINO ? 4 : 1;
  int w_off[WPA][WQ];      // element offsets of the four pixels (-1: outside the map -> zero page)
  float w_sr = 1.f, w_sc = 1.f;
  if constexpr (WINO) {
    const int xi = blockIdx.y, ci = xi >> 2, cj = xi & 3;
    const int ra = ci == 0 ? 0 : (ci == 2 ? 2 : 1), rb = ci == 2 ? 1 : (ci == 3 ? 3 : 2);
    const int ca = cj == 0 ? 0 : (cj == 2 ? 2 : 1), cb = cj == 2 ? 1 : (cj == 3 ? 3 : 2);
    w_sr = ci == 1 ? 1.f : -1.f;
    w_sc = cj == 1 ? 1.f : -1.f;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int m = m0 + i * RPP + row0;
      const int tx = m % p.wtw, t2 = m / p.wtw;
      const int ty = t2 % p.wth, img = t2 / p.wth;
      const int hr[2] = {2 * ty - 1 + ra, 2 * ty - 1 + rb}, wc[2] = {2 * tx - 1 + ca, 2 * tx - 1 + cb};
#pragma unroll
      for (int q = 0; q < 4; ++q) {      // q = 0: (ra, ca), 1: (rb, ca), 2: (ra, cb), 3: (rb, cb)
        const int hi = hr[q & 1], wi = wc[q >> 1];
        const bool ok = m < p.M && (unsigned)hi < (unsigned)p.wiH && (unsigned)wi < (unsigned)p.wiW;
        w_off[i][q] = ok ? ((img * p.wiH + hi) * p.wiW + wi) * p.C : -1;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m0 + i * RPP + row0;
    if (!WINO && m < p.M) {
      const int img = m / (p.Ho * p.Wo);
      const int rem = m - img * p.Ho * p.Wo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_hi0[i] = ho * p.stride - p.pad;
      a_wi0[i] = wo * p.stride - p.pad;
      a_base[i] = ((img * p.H + a_hi0[i]) * p.W + a_wi0[i]) * p.C;
    } else {
      a_hi0[i] = -(1 << 20);  // never in range
      a_wi0[i] = 0;
      a_base[i] = 0;
    }
  }

  // two register sets: the tiles of steps s+1 and s+2 are in flight while step s computes
  f32x4 ra[2][PA], rb[2][PB];
  f32x4 rw[2][WPA][WQ];    // WINO: the four raw pixels per A chunk stay in flight; they are combined when the tile is stored
  typedef std::integral_constant<int, 0> Set0;
  typedef std::integral_constant<int, 1> Set1;
  // tap state for the ALIGNED path (C % 32 == 0: a K-step never straddles a tap)
  int tr = 0, ts = 0, tc = 0;
  if (ALIGNED) {
    const int kf = step_begin * BK;
    const int tap = kf / p.C;
    tc = kf - tap * p.C;
    tr = tap / p.S;
    ts = tap - tr * p.S;
  }

  // load_tiles is called for consecutive steps (the ALIGNED tap state advances by one step per call)
  auto load_tiles = [&](int step, auto set) {
    constexpr int SL = decltype(set)::value;
    const int kflat = step * BK + kc * 4;
    if constexpr (WINO) {
      const int koff = tc + kc * 4;          // R = S = 1: a K-step is a channel block
#pragma unroll
      for (int i = 0; i < PA; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rw[SL][i][q] = *reinterpret_cast<const f32x4*>(w_off[i][q] >= 0 ? px + w_off[i][q] + koff : p.zero);
      tc += BK;
    } else if (ALIGNED) {
      const int koff = (tr * p.W + ts) * p.C + tc + kc * 4;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int hi = a_hi0[i] + tr, wi = a_wi0[i] + ts;
        const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        ra[SL][i] = *reinterpret_cast<const f32x4*>(ok ? px + a_base[i] + koff : p.zero);
      }
      tc += BK;
      if (tc == p.C) {
        tc = 0;
        if (++ts == p.S) { ts = 0; ++tr; }
      }
    } else {
      const int tap = kflat / p.C;
      const int c = kflat - tap * p.C;
      const int r = tap / p.S, s = tap - r * p.S;
      const int koff = (r * p.W + s) * p.C + c;
      const bool kok = kflat < p.Ktot;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int hi = a_hi0[i] + r, wi = a_wi0[i] + s;
        const bool ok = kok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        ra[SL][i] = *reinterpret_cast<const f32x4*>(ok ? px + a_base[i] + koff : p.zero);
      }
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int n = n0 + j * RPP + row0;
      const bool ok = n < p.K && kflat < p.Ktot;
      rb[SL][j] = *reinterpret_cast<const f32x4*>(ok ? pw + (size_t)n * p.Ktot + kflat : p.zero);
    }
  };
  auto store_tiles = [&](int buf, auto set) {
    constexpr int SL = decltype(set)::value;
    float* a = As + buf * BM * LDS_PITCH + row0 * LDS_PITCH + kc * 4;
    float* b = Bs + buf * BN * LDS_PITCH + row0 * LDS_PITCH + kc * 4;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      if constexpr (WINO) {
        const f32x4 lo = rw[SL][i][0] + w_sr * rw[SL][i][1];      // r[ci][ca] = d[ra][ca] +- d[rb][ca]
        const f32x4 hi = rw[SL][i][2] + w_sr * rw[SL][i][3];      // r[ci][cb]
        *reinterpret_cast<f32x4*>(a + i * RPP * LDS_PITCH) = lo + w_sc * hi;
      } else {
        *reinterpret_cast<f32x4*>(a + i * RPP * LDS_PITCH) = ra[SL][i];
      }
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) *reinterpret_cast<f32x4*>(b + j * RPP * LDS_PITCH) = rb[SL][j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets: lane (l&31) = row inside the 32x32 tile, (l>>5) selects k in {4h..4h+3}
  const int frag = (lane & 31) * LDS_PITCH + 4 * (lane >> 5);
  const int a_frag = (wr * TM * 32) * LDS_PITCH + frag;
  const int b_frag = (wc * TN * 32) * LDS_PITCH + frag;

  f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];  // two fragment sets: MFMAs on one, LDS reads into the other
  auto read_frags = [&](f32x4* fa, f32x4* fb, int buf, int kk) {
    const float* Ab = As + buf * BM * LDS_PITCH + a_frag + kk * 8;
    const float* Bb = Bs + buf * BN * LDS_PITCH + b_frag + kk * 8;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDS_PITCH);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDS_PITCH);
  };
  // operands are (weights, activations): D col (lane) = pixel, D row (register) = channel
  auto mma = [&](const f32x4* fa, const f32x4* fb) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][q], fa[i][q], acc[i][j], 0, 0, 0);
  };

  const int nsteps = step_end - step_begin;
  if (nsteps > 0) {
    load_tiles(step_begin, Set0{});
    store_tiles(0, Set0{});
    if (nsteps > 1) load_tiles(step_begin + 1, Set0{});  // tile t >= 1 waits in set (t - 1) & 1 until step t - 1 stores it
    if (nsteps > 2) load_tiles(step_begin + 2, Set1{});
  }
  __syncthreads();
  if (nsteps > 0) read_frags(fa0, fb0, 0, 0);

  int cur = 0;
  // one K-step; `set` holds tile it + 1 and is refilled with tile it + 3 once that one is in LDS.  STEADY: tiles it+1 and
  // it+3 exist - no conditions around the loads, so the compiler counts them and waits for the OLDER set only.
  auto kstep = [&](int it, auto set, auto steady) {
    constexpr bool STEADY = decltype(steady)::value;
    const bool more = STEADY || it + 1 < nsteps;
    read_frags(fa1, fb1, cur, 1);
    mma(fa0, fb0);
    read_frags(fa0, fb0, cur, 2);
    mma(fa1, fb1);
    if (more) store_tiles(cur ^ 1, set);
    read_frags(fa1, fb1, cur, 3);
    mma(fa0, fb0);
    if (STEADY || it + 3 < nsteps) load_tiles(step_begin + it + 3, set);
    __syncthreads();  // buffer cur^1 complete; every wave has its last fragments of buffer cur in registers
    if (more) read_frags(fa0, fb0, cur ^ 1, 0);
    mma(fa1, fb1);
    cur ^= 1;
  };
  int it = 0;
  for (; it + 4 < nsteps; it += 2) {
    kstep(it, Set0{}, std::true_type{});
    kstep(it + 1, Set1{}, std::true_type{});
  }
  for (; it < nsteps; it += 2) {
    kstep(it, Set0{}, std::false_type{});
    if (it + 1 < nsteps) kstep(it + 1, Set1{}, std::false_type{});
  }

  if (p.epi_lds && (p.K & 3) == 0)   // (uniform branch; every wave is past the K loop's last barrier and reads no LDS any more)
    conv_epilogue_lds<TM, TN>(p, acc, m0, n0, wr, wc, lane, smem + wave * 32 * LDS_PITCH);
  else
    conv_epilogue<TM, TN>(p, acc, m0, n0, wr, wc, lane);
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant for the 8-wave tiles (C % 32 == 0): the A/B tiles go global -> LDS with
// global_load_lds_dwordx4 (no VGPR staging, no ds_write pass), three LDS stages, loads issued two K-steps
// ahead, ONE raw s_barrier per K-step behind a counted vmcnt.
//   LDS stage = [BM + BN rows][32 floats], 128-byte rows WITHOUT padding: an LDS-DMA wave instruction writes
//   lane i at base + 16*i, i.e. 8 whole rows per instruction.  Bank conflicts of the ds_read_b128 fragment
//   reads (lane -> row) are removed by an XOR swizzle of the 16-byte chunk index with (row >> 1) & 7, applied
//   on the per-lane GLOBAL source address and again on the read address (guide rule 21).
//   Out-of-range lanes (padding taps, M / K tails) read a zero page instead of branching.
// Numerics are those of conv_igemm_f32 (same k order), so split_k == 1 results are bit-identical.
// ------------------------------------------------------------------------------------------------

template <int WM, int WN>
__global__ __launch_bounds__(512, 2) void conv_igemm_dma_f32(const ConvParams p) {
  constexpr int TM = 2, TN = 2;
  constexpr int BM = 64 * WM, BN = 64 * WN;
  static_assert(WM * WN == 8, "8 waves");
  constexpr int PA = BM / 64, PB = BN / 64;          // LDS-DMA instructions (8 rows each) per wave per K-step
  constexpr int STAGE = (BM + BN) * 32;              // floats per stage
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [3][BM + BN][32]

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const float* const px = p.x + (size_t)blockIdx.y * p.gx;
  const float* const pw = p.w + (size_t)blockIdx.y * p.gw;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  int tile_m, tile_n;
  tile_coords(tile, p.tiles_m, p.tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int step_begin = blockIdx.z * p.steps_per_split;
  const int step_end = min(step_begin + p.steps_per_split, p.ksteps);
  const int nsteps = step_end - step_begin;

  // ---- per-lane sources: lane i of instruction q feeds row 8q + (i >> 3), physical chunk i & 7 -------------
  const int lrow = lane >> 3, lchunk = lane & 7;
  int a_base[PA], a_hi0[PA], a_wi0[PA], a_swz[PA];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int row = (wave * PA + j) * 8 + lrow;
    const int m = m0 + row;
    a_swz[j] = (lchunk ^ ((row >> 1) & 7)) * 4;
    if (m < p.M) {
      const int img = m / (p.Ho * p.Wo);
      const int rem = m - img * p.Ho * p.Wo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_hi0[j] = ho * p.stride - p.pad;
      a_wi0[j] = wo * p.stride - p.pad;
      a_base[j] = ((img * p.H + a_hi0[j]) * p.W + a_wi0[j]) * p.C;
    } else {
      a_hi0[j] = -(1 << 20);
      a_wi0[j] = 0;
      a_base[j] = 0;
    }
  }
  const float* b_src[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int row = (wave * PB + j) * 8 + lrow;
    const int n = n0 + row;
    b_src[j] = n < p.K ? pw + (size_t)n * p.Ktot + (lchunk ^ ((row >> 1) & 7)) * 4 : nullptr;
  }
  int tr, ts, tc;
  {
    const int kf = step_begin * BK;
    const int tap = kf / p.C;
    tc = kf - tap * p.C;
    tr = tap / p.S;
    ts = tap - tr * p.S;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  // issue() is called for consecutive steps (the tap state advances by one step per call)
  auto issue = [&](int step, int stage) {
    float* sA = smem + stage * STAGE + (wave * PA) * 8 * 32;
    float* sB = smem + stage * STAGE + BM * 32 + (wave * PB) * 8 * 32;
    const int koff = (tr * p.W + ts) * p.C + tc;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      const int hi = a_hi0[j] + tr, wi = a_wi0[j] + ts;
      const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      const float* src = ok ? px + a_base[j] + koff + a_swz[j] : p.zero;
      __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(sA + j * 8 * 32), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const float* src = b_src[j] ? b_src[j] + step * BK : p.zero;
      __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(sB + j * 8 * 32), 16, 0, 0);
    }
    tc += BK;
    if (tc == p.C) {
      tc = 0;
      if (++ts == p.S) { ts = 0; ++tr; }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment reads: row = tile row of lane & 31, chunk 2*kk + (lane >> 5), XOR-swizzled
  int fa_row[TM], fa_x[TM], fb_row[TN], fb_x[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = (wr * TM + i) * 32 + (lane & 31);
    fa_row[i] = row * 32;
    fa_x[i] = (row >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = (wc * TN + j) * 32 + (lane & 31);
    fb_row[j] = BM * 32 + row * 32;
    fb_x[j] = (row >> 1) & 7;
  }
  const int lh = lane >> 5;
  f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  auto read_frags = [&](f32x4* fa, f32x4* fb, int stage, int kk) {
    const float* base = smem + stage * STAGE;
    const int c = 2 * kk + lh;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(base + fa_row[i] + ((c ^ fa_x[i]) << 2));
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(base + fb_row[j] + ((c ^ fb_x[j]) << 2));
  };
  auto mma = [&](const f32x4* fa, const f32x4* fb) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][q], fa[i][q], acc[i][j], 0, 0, 0);
  };

  // Schedule of one K-step (4 groups of 16 MFMAs per wave).  The barrier sits in the MIDDLE of the step:
  //   group 0, 1   on stage s (its group-0 fragments were read at the end of the previous step)
  //   vmcnt + barrier: stage s+1 has landed for every wave, and every wave is past step s-1
  //   issue the loads of step s+2 into the stage step s-1 used
  //   group 2, then the group-0 fragment reads of stage s+1, group 3
  // so neither the barrier nor the first fragment reads of a stage leave the MFMA pipe without queued work.
  if (nsteps > 0) issue(step_begin, 0);
  if (nsteps > 1) issue(step_begin + 1, 1);
  if (nsteps > 0) {
    if (nsteps > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_frags(fa0, fb0, 0, 0);
  }
  int stage = 0;
  for (int s = 0; s < nsteps; ++s) {
    const int next = stage == 2 ? 0 : stage + 1;
    read_frags(fa1, fb1, stage, 1);
    mma(fa0, fb0);
    read_frags(fa0, fb0, stage, 2);
    mma(fa1, fb1);
    if (s + 1 < nsteps) {
      // only the loads of step s+1 are outstanding here (step s+2 is issued below)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < nsteps) issue(step_begin + s + 2, stage == 0 ? 2 : stage - 1);
    }
    read_frags(fa1, fb1, stage, 3);
    mma(fa0, fb0);
    if (s + 1 < nsteps) read_frags(fa0, fb0, next, 0);
    mma(fa1, fb1);
    stage = next;
  }
  if (p.epi_lds && (p.K & 3) == 0) {
    __syncthreads();   // the last step has no barrier behind its fragment reads: every wave must be done with the stages
    conv_epilogue_lds<TM, TN>(p, acc, m0, n0, wr, wc, lane, smem + wave * 32 * LDS_PITCH);
  } else {
    conv_epilogue<TM, TN>(p, acc, m0, n0, wr, wc, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA 128x128 tile, 4 waves, TWO LDS stages (64 KB): two workgroups per CU.  For the GEMMs with many
// output channels and a short reduction (layer4's 512 -> 2048 / 1024 -> 2048 expansions, the Winograd GEMMs): the
// 64x64 tile the autotuner otherwise picks there pulls 4x the bytes per FLOP from the L2s (6.4 TB/s, 1.88 GB per call),
// the 256x128 tile keeps one workgroup per CU whose epilogue nobody overlaps.  Same operand layout, swizzle, zero page
// and k order as conv_igemm_dma_f32: bit-identical results for split_k == 1.
// ------------------------------------------------------------------------------------------------
template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_igemm_dma2_f32(const ConvParams p) {
  constexpr int TM = 2, TN = 2;
  constexpr int BM = 64 * WM, BN = 64 * WN;
  static_assert(WM * WN == 4, "4 waves");
  constexpr int PA = BM / 32, PB = BN / 32;          // LDS-DMA instructions (8 rows each) per wave per K-step
  constexpr int STAGE = (BM + BN) * 32;              // floats per stage
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][BM + BN][32]

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const float* const px = p.x + (size_t)blockIdx.y * p.gx;
  const float* const pw = p.w + (size_t)blockIdx.y * p.gw;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  int tile_m, tile_n;
  tile_coords(tile, p.tiles_m, p.tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int step_begin = blockIdx.z * p.steps_per_split;
  const int step_end = min(step_begin + p.steps_per_split, p.ksteps);
  const int nsteps = step_end - step_begin;

  // ---- per-lane sources: lane i of instruction q feeds row 8q + (i >> 3), physical chunk i & 7 -------------
  const int lrow = lane >> 3, lchunk = lane & 7;
  int a_base[PA], a_hi0[PA], a_wi0[PA], a_swz[PA];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int row = (wave * PA + j) * 8 + lrow;
    const int m = m0 + row;
    a_swz[j] = (lchunk ^ ((row >> 1) & 7)) * 4;
    if (m < p.M) {
      const int img = m / (p.Ho * p.Wo);
      const int rem = m - img * p.Ho * p.Wo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_hi0[j] = ho * p.stride - p.pad;
      a_wi0[j] = wo * p.stride - p.pad;
      a_base[j] = ((img * p.H + a_hi0[j]) * p.W + a_wi0[j]) * p.C;
    } else {
      a_hi0[j] = -(1 << 20);
      a_wi0[j] = 0;
      a_base[j] = 0;
    }
  }
  const float* b_src[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int row = (wave * PB + j) * 8 + lrow;
    const int n = n0 + row;
    b_src[j] = n < p.K ? pw + (size_t)n * p.Ktot + (lchunk ^ ((row >> 1) & 7)) * 4 : nullptr;
  }
  int tr, ts, tc;
  {
    const int kf = step_begin * BK;
    const int tap = kf / p.C;
    tc = kf - tap * p.C;
    tr = tap / p.S;
    ts = tap - tr * p.S;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  // issue() is called for consecutive steps (the tap state advances by one step per call)
  auto issue = [&](int step, int stage) {
    float* sA = smem + stage * STAGE + (wave * PA) * 8 * 32;
    float* sB = smem + stage * STAGE + BM * 32 + (wave * PB) * 8 * 32;
    const int koff = (tr * p.W + ts) * p.C + tc;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      const int hi = a_hi0[j] + tr, wi = a_wi0[j] + ts;
      const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      const float* src = ok ? px + a_base[j] + koff + a_swz[j] : p.zero;
      __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(sA + j * 8 * 32), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const float* src = b_src[j] ? b_src[j] + step * BK : p.zero;
      __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(sB + j * 8 * 32), 16, 0, 0);
    }
    tc += BK;
    if (tc == p.C) {
      tc = 0;
      if (++ts == p.S) { ts = 0; ++tr; }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment reads: row = tile row of lane & 31, chunk 2*kk + (lane >> 5), XOR-swizzled
  int fa_row[TM], fa_x[TM], fb_row[TN], fb_x[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = (wr * TM + i) * 32 + (lane & 31);
    fa_row[i] = row * 32;
    fa_x[i] = (row >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = (wc * TN + j) * 32 + (lane & 31);
    fb_row[j] = BM * 32 + row * 32;
    fb_x[j] = (row >> 1) & 7;
  }
  const int lh = lane >> 5;
  f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  auto read_frags = [&](f32x4* fa, f32x4* fb, int stage, int kk) {
    const float* base = smem + stage * STAGE;
    const int c = 2 * kk + lh;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(base + fa_row[i] + ((c ^ fa_x[i]) << 2));
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(base + fb_row[j] + ((c ^ fb_x[j]) << 2));
  };
  auto mma = [&](const f32x4* fa, const f32x4* fb) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][q], fa[i][q], acc[i][j], 0, 0, 0);
  };

  // Two stages, one K-step of look-ahead: the barrier at the END of step s says "every wave has its last fragments of
  // stage s in registers and every wave's part of stage s+1 has landed", so the loads of step s+2 may overwrite stage s
  // right after it.  The fragment reads that follow a barrier leave a bubble in this workgroup's MFMA stream; the
  // second workgroup of the CU (64 KB of LDS each) fills it.
  if (nsteps > 0) {
    issue(step_begin, 0);
    if (nsteps > 1) issue(step_begin + 1, 1);
    if (nsteps > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int s = 0; s < nsteps; ++s) {
    const int stage = s & 1;
    read_frags(fa0, fb0, stage, 0);
    read_frags(fa1, fb1, stage, 1);
    mma(fa0, fb0);
    read_frags(fa0, fb0, stage, 2);
    mma(fa1, fb1);
    read_frags(fa1, fb1, stage, 3);
    mma(fa0, fb0);
    if (s + 1 < nsteps) {
      // the last fragments of this stage are in registers (the MFMAs below only read registers): wait for the loads of
      // step s+1 (the only ones outstanding), meet the other waves, refill this stage with step s+2
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < nsteps) issue(step_begin + s + 2, stage);
    }
    mma(fa1, fb1);
  }
  if (p.epi_lds && (p.K & 3) == 0) {
    __syncthreads();
    conv_epilogue_lds<TM, TN>(p, acc, m0, n0, wr, wc, lane, smem + wave * 32 * LDS_PITCH);
  } else {
    conv_epilogue<TM, TN>(p, acc, m0, n0, wr, wc, lane);
  }
}


// ------------------------------------------------------------------------------------------------
// LDS-DMA through BUFFER loads (buffer_load_dwordx4 ... lds), three stages, any 4- or 8-wave tile: written for the small
// tiles, above all the 64x64 tile whose waves own ONE 32x32 accumulator.
//
// Why (profiles/r05_mfma_chain.txt, profiles/r05_conv_pmc_probe.md): on gfx950 nothing a wave issues hides behind its own
// fp32 MFMAs for free - beside a dependent v_mfma_f32_32x32x2_f32 chain one wave per SIMD pays ~7 cycles per v_fma, ~5 per
// SALU instruction, ~66 per ds_write_b128 and ~85 per global_load_dwordx4 of MFMA time.  The register-staged 64x64 kernel
// issues, per 16 MFMAs (1024 cycles), 4 global loads + 4 ds_write_b128 + ~27 VALU + ~25 SALU instructions: a wave alone on
// its SIMD keeps the matrix pipe busy 53 % of its life, and 2.7 such waves per SIMD only reach 50 %.  This kernel removes the
// instructions instead of rescheduling them:
//   * tiles go global -> LDS by DMA: no VGPR staging, no ds_write pass, no vmcnt -> ds_write dependency;
//   * addresses are  buffer resource (SGPRs) + per-lane byte offset (VGPR, constant while the tap is) + a scalar offset that
//     advances with the K-step: ZERO vector instructions per K-step on a 1x1 layer (a 3x3 layer recomputes its per-lane
//     offsets once per tap, i.e. every C/32 steps);
//   * out-of-range lanes (padding taps, M / K tails) carry the offset 2^31: the buffer's range check returns zeros, no
//     zero page, no select.  (Activation and filter ranges must be < 2 GB: the host falls back to the other kernels.)
// LDS layout, XOR swizzle, k order and epilogue are those of conv_igemm_dma_f32: results are bit-identical to every other
// kernel for split_k == 1.
// ------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(64 * WM * WN, (TM * TN == 1) ? 3 : 2) void conv_igemm_buf_f32(const ConvParams p) {
  constexpr int NW = WM * WN;
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int PA = BM / (8 * NW), PB = BN / (8 * NW);   // DMA instructions (8 rows of 128 bytes each) per wave per K-step
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must be a multiple of the DMA pass");
  constexpr int STAGE = (BM + BN) * 32;                   // floats per stage
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [3][BM + BN][32]

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  int tile_m, tile_n;
  tile_coords(tile, p.tiles_m, p.tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int step_begin = blockIdx.z * p.steps_per_split;
  const int step_end = min(step_begin + p.steps_per_split, p.ksteps);
  const int nsteps = step_end - step_begin;

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x + (size_t)blockIdx.y * p.gx), 0, (int)p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.w + (size_t)blockIdx.y * p.gw), 0, (int)p.wbytes, 0x00020000);

  // ---- per-lane sources: lane i of instruction j feeds row 8 (wave PA + j) + (i >> 3), physical chunk i & 7 -------------
  const int lrow = lane >> 3, lchunk = lane & 7;
  int a_pix[PA], a_hi0[PA], a_wi0[PA];      // byte offset of pixel (img, hi0, wi0), -1 for a row past M
  unsigned a_swz[PA], vA[PA], vB[PB];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int row = (wave * PA + j) * 8 + lrow;
    const int m = m0 + row;
    a_swz[j] = (unsigned)((lchunk ^ ((row >> 1) & 7)) * 16);
    if (m < p.M) {
      const int img = m / (p.Ho * p.Wo);
      const int rem = m - img * p.Ho * p.Wo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_hi0[j] = ho * p.stride - p.pad;
      a_wi0[j] = wo * p.stride - p.pad;
      a_pix[j] = ((img * p.H + a_hi0[j]) * p.W + a_wi0[j]) * p.C * 4;
    } else {
      a_hi0[j] = -(1 << 20);
      a_wi0[j] = 0;
      a_pix[j] = 0;
    }
    vA[j] = OOB;
  }
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int row = (wave * PB + j) * 8 + lrow;
    const int n = n0 + row;
    vB[j] = n < p.K ? (unsigned)(n * p.Ktot * 4 + (lchunk ^ ((row >> 1) & 7)) * 16) : OOB;
  }
  int tr, ts, tc;
  {
    const int kf = step_begin * BK;
    const int tap = kf / p.C;
    tc = kf - tap * p.C;
    tr = tap / p.S;
    ts = tap - tr * p.S;
  }
  bool new_tap = true;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  // issue() is called for consecutive steps (the tap state advances by one step per call)
  auto issue = [&](int step, int stage) {
    float* sA = smem + stage * STAGE + (wave * PA) * 8 * 32;
    float* sB = smem + stage * STAGE + BM * 32 + (wave * PB) * 8 * 32;
    if (new_tap) {                          // uniform: once per tap (once per kernel on a 1x1 layer)
      const int toff = (tr * p.W + ts) * p.C * 4;
#pragma unroll
      for (int j = 0; j < PA; ++j) {
        const int hi = a_hi0[j] + tr, wi = a_wi0[j] + ts;
        const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        vA[j] = ok ? (unsigned)(a_pix[j] + toff) + a_swz[j] : OOB;
      }
    }
    const int sa = tc * 4, sb = step * (BK * 4);
#pragma unroll
    for (int j = 0; j < PA; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)(sA + j * 8 * 32), 16, (int)vA[j], sa, 0, 0);
#pragma unroll
    for (int j = 0; j < PB; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr)(sB + j * 8 * 32), 16, (int)vB[j], sb, 0, 0);
    tc += BK;
    new_tap = tc == p.C;
    if (new_tap) {
      tc = 0;
      if (++ts == p.S) { ts = 0; ++tr; }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment reads: row = tile row of lane & 31, logical chunk 2 kk + (lane >> 5), XOR-swizzled; the four chunk offsets of a
  // row are kept in registers (no address arithmetic in the loop)
  const int lh = lane >> 5;
  int fa_off[TM][4], fb_off[TN][4];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = (wr * TM + i) * 32 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fa_off[i][kk] = row * 32 + (((2 * kk + lh) ^ ((row >> 1) & 7)) << 2);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = (wc * TN + j) * 32 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fb_off[j][kk] = BM * 32 + row * 32 + (((2 * kk + lh) ^ ((row >> 1) & 7)) << 2);
  }
  f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  auto read_frags = [&](f32x4* fa, f32x4* fb, int stage, int kk) {
    const float* base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(base + fa_off[i][kk]);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(base + fb_off[j][kk]);
  };
  auto mma = [&](const f32x4* fa, const f32x4* fb) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][q], fa[i][q], acc[i][j], 0, 0, 0);
  };

  // Schedule of one K-step as in conv_igemm_dma_f32: the barrier sits in the MIDDLE of the step (stage s+1 has landed for
  // every wave and every wave is past step s-1), the loads of step s+2 are issued right behind it.
  if (nsteps > 0) issue(step_begin, 0);
  if (nsteps > 1) issue(step_begin + 1, 1);
  if (nsteps > 0) {
    if (nsteps > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_frags(fa0, fb0, 0, 0);
  }
  int stage = 0;
  for (int s = 0; s < nsteps; ++s) {
    const int next = stage == 2 ? 0 : stage + 1;
    read_frags(fa1, fb1, stage, 1);
    mma(fa0, fb0);
    read_frags(fa0, fb0, stage, 2);
    mma(fa1, fb1);
    if (s + 1 < nsteps) {
      // only the loads of step s+1 are outstanding here (step s+2 is issued below)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < nsteps) issue(step_begin + s + 2, stage == 0 ? 2 : stage - 1);
    }
    read_frags(fa1, fb1, stage, 3);
    mma(fa0, fb0);
    if (s + 1 < nsteps) read_frags(fa0, fb0, next, 0);
    mma(fa1, fb1);
    stage = next;
  }
  if (p.epi_lds && (p.K & 3) == 0) {
    __syncthreads();   // the last step has no barrier behind its fragment reads: every wave must be done with the stages
    conv_epilogue_lds<TM, TN>(p, acc, m0, n0, wr, wc, lane, smem + wave * 32 * LDS_PITCH);
  } else {
    conv_epilogue<TM, TN>(p, acc, m0, n0, wr, wc, lane);
  }
}


// ------------------------------------------------------------------------------------------------
// PERSISTENT form of the buffer-load LDS-DMA kernel for the 64x64 tile (plan tile index 13).
//
// Section 4.10 of DESIGN.md: a 64x64 tile of a short reduction (layer1-3: 2-16 K-steps, ~1-8 us of MFMA work) is launched,
// computes its addresses, waits for its first two DMA round trips, runs its K loop, stores and exits - the skeleton costs
// as much as the loop, and co-resident workgroups only partly cover each other's skeletons.  Here a workgroup is resident for
// the whole launch (grid = 2 workgroups per CU at most) and walks its WORK ITEMS (tile x group x K-split, item i, i + grid,
// ...) as ONE stream of K-steps through the same three-stage LDS ring: while the last K-steps of an item compute, the DMA of the
// next item's first steps is already in flight (a producer cursor runs two steps ahead of the consumer cursor, across item
// boundaries), so from its second item on a workgroup pays neither dispatch nor address set-up nor a cold first load before
// its MFMAs - only the epilogue (LDS transpose in a patch of its own, stores) sits between two items' MFMAs.
// Same LDS image, k order and epilogue arithmetic as conv_igemm_buf_f32: bit-identical for split_k == 1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void conv_igemm_pbuf_f32(const ConvParams p, int groups, int splits) {
  constexpr int WN = 2, TM = 1, TN = 1;
  constexpr int BM = 64, BN = 64, PA = 2, PB = 2;       // 4 waves x 8 rows per DMA instruction: 2 + 2 instructions per wave and step
  constexpr int STAGE = (BM + BN) * 32;                 // floats per stage
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [3][BM + BN][32] ring + [4 waves][32][36] epilogue patches
  float* const patch = smem + 3 * STAGE + (threadIdx.x >> 6) * 32 * LDS_PITCH;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int total = ntiles * groups * splits;
  const int stride = gridDim.x;
  typedef __attribute__((address_space(3))) void* lds_ptr;

  // item -> (tile, group, split); tiles in the XCD-aware raster order of the other kernels
  auto item_coords = [&](int item, int& m0, int& n0, int& g, int& z) {
    const int tl = item % ntiles, gz = item / ntiles;
    g = gz % groups;
    z = gz / groups;
    int tile_m, tile_n;
    tile_coords(xcd_remap(tl, ntiles), p.tiles_m, p.tiles_n, tile_m, tile_n);
    m0 = tile_m * BM;
    n0 = tile_n * BN;
  };
  auto item_steps = [&](int z, int& begin) {
    begin = z * p.steps_per_split;
    return min(begin + p.steps_per_split, p.ksteps) - begin;
  };

  // ---- producer cursor: the item / step whose tiles are issued next ------------------------------------------------------
  const int lrow = lane >> 3, lchunk = lane & 7;
  int p_item = blockIdx.x, p_step = 0, p_nst = 0, p_begin = 0, p_stage = 0;
  int a_pix[PA], a_hi0[PA], a_wi0[PA];
  unsigned a_swz[PA], vA[PA], vB[PB];
  int tr = 0, ts = 0, tc = 0;
  bool new_tap = true;
  __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.xbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.wbytes, 0x00020000);
  auto producer_setup = [&]() {                        // geometry of item p_item (uniform: p_item < total)
    int m0, n0, g, z;
    item_coords(p_item, m0, n0, g, z);
    p_nst = item_steps(z, p_begin);
    p_step = 0;
    rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)g * p.gx), 0, (int)p.xbytes, 0x00020000);
    rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w + (size_t)g * p.gw), 0, (int)p.wbytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      const int row = (wave * PA + j) * 8 + lrow;
      const int m = m0 + row;
      a_swz[j] = (unsigned)((lchunk ^ ((row >> 1) & 7)) * 16);
      if (m < p.M) {
        const int img = m / (p.Ho * p.Wo);
        const int rem = m - img * p.Ho * p.Wo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        a_hi0[j] = ho * p.stride - p.pad;
        a_wi0[j] = wo * p.stride - p.pad;
        a_pix[j] = ((img * p.H + a_hi0[j]) * p.W + a_wi0[j]) * p.C * 4;
      } else {
        a_hi0[j] = -(1 << 20);
        a_wi0[j] = 0;
        a_pix[j] = 0;
      }
      vA[j] = OOB;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int row = (wave * PB + j) * 8 + lrow;
      const int n = n0 + row;
      vB[j] = n < p.K ? (unsigned)(n * p.Ktot * 4 + (lchunk ^ ((row >> 1) & 7)) * 16) : OOB;
    }
    const int kf = p_begin * BK;
    const int tap = kf / p.C;
    tc = kf - tap * p.C;
    tr = tap / p.S;
    ts = tap - tr * p.S;
    new_tap = true;
  };
  // issues the tiles of the producer's step into the producer's stage and advances the cursor; false when the stream is over
  auto issue_next = [&]() -> bool {
    if (p_item >= total) return false;
    float* sA = smem + p_stage * STAGE + (wave * PA) * 8 * 32;
    float* sB = smem + p_stage * STAGE + BM * 32 + (wave * PB) * 8 * 32;
    if (new_tap) {
      const int toff = (tr * p.W + ts) * p.C * 4;
#pragma unroll
      for (int j = 0; j < PA; ++j) {
        const int hi = a_hi0[j] + tr, wi = a_wi0[j] + ts;
        const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        vA[j] = ok ? (unsigned)(a_pix[j] + toff) + a_swz[j] : OOB;
      }
    }
    const int sa = tc * 4, sb = (p_begin + p_step) * (BK * 4);
#pragma unroll
    for (int j = 0; j < PA; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)(sA + j * 8 * 32), 16, (int)vA[j], sa, 0, 0);
#pragma unroll
    for (int j = 0; j < PB; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr)(sB + j * 8 * 32), 16, (int)vB[j], sb, 0, 0);
    tc += BK;
    new_tap = tc == p.C;
    if (new_tap) {
      tc = 0;
      if (++ts == p.S) { ts = 0; ++tr; }
    }
    p_stage = p_stage == 2 ? 0 : p_stage + 1;
    if (++p_step == p_nst) {
      p_item += stride;
      if (p_item < total) producer_setup();
    }
    return true;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  const int lh = lane >> 5;
  int fa_off[4], fb_off[4];
  {
    const int rowa = wr * 32 + (lane & 31), rowb = wc * 32 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      fa_off[kk] = rowa * 32 + (((2 * kk + lh) ^ ((rowa >> 1) & 7)) << 2);
      fb_off[kk] = BM * 32 + rowb * 32 + (((2 * kk + lh) ^ ((rowb >> 1) & 7)) << 2);
    }
  }
  f32x4 fa0, fb0, fa1, fb1;
  auto read_frags = [&](f32x4& fa, f32x4& fb, int stage, int kk) {
    const float* base = smem + stage * STAGE;
    fa = *reinterpret_cast<const f32x4*>(base + fa_off[kk]);
    fb = *reinterpret_cast<const f32x4*>(base + fb_off[kk]);
  };
  auto mma = [&](const f32x4& fa, const f32x4& fb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[q], fa[q], acc[0][0], 0, 0, 0);
  };

  if (blockIdx.x >= total) return;                     // (uniform) more workgroups than items: nothing to do
  // The stream: step g of this workgroup's items lives in stage g % 3.  `ahead` = steps issued - steps consumed (1 or 2 at the
  // top of a step): ahead >= 2 <=> a next step exists.
  producer_setup();
  issue_next();                                        // stream step 0
  int ahead = issue_next() ? 2 : 1;                    // stream step 1
  if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_frags(fa0, fb0, 0, 0);
  int stage = 0;
  for (int item = blockIdx.x; item < total; item += stride) {
    int m0, n0, g, z, begin;
    item_coords(item, m0, n0, g, z);
    const int nst = item_steps(z, begin);
    for (int s = 0; s < nst; ++s) {
      const int next = stage == 2 ? 0 : stage + 1;
      const bool more = ahead >= 2;
      read_frags(fa1, fb1, stage, 1);
      mma(fa0, fb0);
      read_frags(fa0, fb0, stage, 2);
      mma(fa1, fb1);
      if (more) {
        // everything this wave has issued is waited for: the DMA of the next stream step and, behind an item boundary, the
        // previous item's stores (issued half a K-step ago)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (issue_next()) ++ahead;                     // stream step + 2 goes into the stage step - 1 used
      }
      read_frags(fa1, fb1, stage, 3);
      mma(fa0, fb0);
      if (more) read_frags(fa0, fb0, next, 0);
      mma(fa1, fb1);
      stage = next;
      --ahead;
    }
    if (p.epi_lds && (p.K & 3) == 0) conv_epilogue_lds<TM, TN>(p, acc, m0, n0, wr, wc, lane, patch, g, z);
    else conv_epilogue<TM, TN>(p, acc, m0, n0, wr, wc, lane, g, z);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  }
}

// Split-K second pass: y = act((sum_z partial[z]) * scale + shift + res), slabs summed in z order.
__global__ __launch_bounds__(256) void conv_splitk_epilogue(const float* __restrict__ partial, int splits,
                                                           size_t mk, int K, const float* scale,
                                                           const float* shift, const float* res, float* y,
                                                           int relu, const float* mask, const float* mscale) {
  for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < mk; o += (size_t)gridDim.x * blockDim.x) {
    float v = partial[o];
    for (int z = 1; z < splits; ++z) v += partial[(size_t)z * mk + o];
    const int n = (int)(o % K);
    v = v * (scale ? scale[n] : 1.f) + (shift ? shift[n] : 0.f);
    if (res) v += res[o];
    if (relu) v = fmaxf(v, 0.f);
    if (mask) v = mask[o] > 0.f ? v * (mscale ? mscale[n] : 1.f) : 0.f;
    y[o] = v;
  }
}

// Workgroup configurations: block tile (64*tm) x (64*tn) = (32*TM*WM) x (32*TN*WN).
struct TileCfg {
  int tm, tn;          // block tile in units of 64 pixels x 64 channels (the frcnn_conv2d_set_tile key)
  int wm, wn, wtm, wtn;  // wave grid and 32x32 tiles per wave
};
constexpr TileCfg kTiles[] = {
    {4, 2, 4, 2, 2, 2},  // 256x128, 8 waves, one workgroup per CU
    {2, 4, 2, 4, 2, 2},  // 128x256, 8 waves
    {2, 2, 2, 2, 2, 2},  // 128x128, 4 waves, two workgroups per CU
    {2, 1, 2, 2, 2, 1},  // 128x64
    {1, 2, 2, 2, 1, 2},  // 64x128
    {1, 1, 2, 2, 1, 1},  // 64x64
    {2, 2, 2, 2, 2, 2},  // 128x128, 4 waves, LDS-DMA with two stages: two workgroups per CU (C % 32 == 0; else as index 2)
    {1, 1, 2, 2, 1, 1},  // 64x64,  LDS-DMA through buffer loads, three stages (conv_igemm_buf_f32; C % 32 == 0 and < 2 GB operands, else as index 5)
    {2, 1, 2, 2, 2, 1},  // 128x64, the same kernel (else as index 3)
    {1, 2, 2, 2, 1, 2},  // 64x128, the same kernel (else as index 4)
    {2, 2, 2, 2, 2, 2},  // 128x128, the same kernel, 96 KB of LDS: one workgroup per CU (else as index 2)
    {4, 2, 4, 2, 2, 2},  // 256x128, the same kernel, 8 waves (else as index 0)
    {2, 4, 2, 4, 2, 2},  // 128x256, the same kernel, 8 waves (else as index 1)
    {1, 1, 2, 2, 1, 1},  // 64x64, PERSISTENT workgroups walking their tiles as one K-step stream (conv_igemm_pbuf_f32; else as index 7)
};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

struct Plan {
  int cfg, splits, steps_per_split;
  int algo = 0;   // 0 = implicit GEMM; 1 = Winograd F(2x2, 3x3) around a grouped GEMM that uses tile `cfg` (splits = 1)
  int fuse_in = 0;   // algo 1 only: the input transform runs inside the 64x64 GEMM's A-tile load (cfg 5, C % 32 == 0)
};
bool winograd_ok(int r, int s, int stride, int pad, int c, int k, int out_stride);
size_t winograd_ws_bytes(int n, int h, int w, int c, int k);
// test / tuning hook: 0 = the autotuner may pick either form, 1 = implicit GEMM only, 2 = Winograd wherever it applies
std::atomic<int> g_algo_mode{0};   // atomic: set from one thread while another may launch
// test / tuning hook (frcnn_conv2d_set_algo bit 4): may the tuner try / forced Winograd use the fused input transform?
std::atomic<int> g_wino_fuse{1};
// test / tuning hook (frcnn_conv2d_set_algo bit 6): 1 = the convolution kernels store through the LDS transpose
std::atomic<int> g_epi_lds{1};

// device address of g_zero_page, resolved once (hipGetSymbolAddress is a host-side lookup: legal during stream capture)
const float* zero_page_address() {
  static std::atomic<const float*> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  const float* z = cached[dev].load();
  if (!z) {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_zero_page)) != hipSuccess) return nullptr;
    z = static_cast<const float*>(ptr);
    cached[dev].store(z);
  }
  return z;
}

// test / tuning hook: force the block tile (0 = automatic choice)
std::atomic<int> g_force_tm{0}, g_force_tn{0};

// Pick the block tile and the K split that minimise the estimated time on 256 CUs.  Units: MFMA
// issue cycles of one SIMD (64 per v_mfma_f32_32x32x2_f32); a CU runs one workgroup's K-step in
// waves_per_simd * TM*TN*16 MFMAs (co-resident 4-wave workgroups share the SIMDs, which the model
// counts as running one after the other).
Plan choose_plan(int M, int K, int ksteps, int forced_splits) {
  static const int split_cand[] = {1, 2, 3, 4, 6, 8, 12, 16};
  Plan best{2, 1, ksteps};
  double best_t = 1e300;
  for (int ci = 0; ci < kNumTiles; ++ci) {
    const TileCfg& c = kTiles[ci];
    if (g_force_tm > 0 && (c.tm != g_force_tm || c.tn != g_force_tn)) continue;
    const int bm = 64 * c.tm, bn = 64 * c.tn;
    const long tiles = (long)((M + bm - 1) / bm) * ((K + bn - 1) / bn);
    const int waves_per_simd = c.wm * c.wn / 4;
    for (int sp : split_cand) {
      if (forced_splits > 0 && sp != forced_splits) continue;
      if (forced_splits <= 0 && sp > 1 && ksteps / sp < 4) continue;
      const int sps = (ksteps + sp - 1) / sp;
      const int real_splits = (ksteps + sps - 1) / sps;
      if (real_splits != sp && forced_splits <= 0) continue;
      const long blocks = tiles * real_splits;
      const long rounds = (blocks + NUM_CU - 1) / NUM_CU;
      const double step_cyc = waves_per_simd * c.wtm * c.wtn * 16 * 64 + 300.0;
      double tcyc = rounds * (sps * step_cyc + 5000.0);
      if (real_splits > 1) {
        // slab write + read-back at ~3 TB/s (2.4 GHz -> 1250 B/cycle) + one more launch
        tcyc += (double)M * K * 4.0 * (real_splits + 1) / 1250.0 + 4000.0;
      }
      if (tcyc < best_t) {
        best_t = tcyc;
        best = Plan{ci, real_splits, sps};
      }
    }
  }
  if (forced_splits > 0 && best_t == 1e300) {
    const int sps = (ksteps + forced_splits - 1) / forced_splits;
    int ci = 5;   // 64x64
    for (int i = kNumTiles - 1; i >= 0; --i)   // first entry with the forced tile (index 6 repeats the 128x128 shape)
      if (g_force_tm > 0 && kTiles[i].tm == g_force_tm && kTiles[i].tn == g_force_tn) ci = i;
    best = Plan{ci, (ksteps + sps - 1) / sps, sps};
  }
  return best;
}

// ---- plan cache / autotuner -----------------------------------------------------------------------
// The analytic model above ranks (tile, split) pairs well for large GEMMs but not for the small, latency-bound
// layers (layer1..3 at one frame): there the 64x64 tile without a K split usually wins by 10-40 %.  With
// frcnn_conv2d_set_autotune(1) the first call of a shape outside stream capture times every candidate on the
// caller's own tensors (HIP events on the launch stream) and caches the winner; later calls — including the
// captured ones — look the plan up.  Off by default: results for split_k = 0 then depend only on the model.
typedef std::array<int, 10> ShapeKey;
std::map<ShapeKey, Plan> g_plan_cache;
std::mutex g_plan_mutex;
std::atomic<int> g_autotune{0};
constexpr size_t kTuneWsCap = (size_t)256 << 20;   // candidates whose split-K slabs exceed this are not tried
constexpr size_t kTuneWinoCap = (size_t)768 << 20;  // same for the Winograd workspace (16 x (tiles x (C + K)) floats)

// The last slot carries the output stride AND whether the call has a residual operand (+ kKeyResidual): a call with a
// residual cannot run as Winograd, so the two kinds of call of one shape are tuned and cached separately (a plan tuned for
// one used to push the other onto the untuned analytic plan for good).
constexpr int kKeyResidual = 256;
ShapeKey shape_key(int n, int h, int w, int c, int k, int r, int s, int stride, int pad, int out_stride,
                   bool has_residual = false) {
  return ShapeKey{n, h, w, c, k, r, s, stride, pad, out_stride + (has_residual ? kKeyResidual : 0)};
}

std::vector<Plan> tune_candidates(long M, int k, int ksteps, bool allow_split) {
  static const int split_cand[] = {1, 2, 3, 4, 6, 8, 12, 16};
  std::vector<Plan> out;
  for (int ci = 0; ci < kNumTiles; ++ci)
    for (int sp : split_cand) {
      if (sp > 1 && (!allow_split || ksteps / sp < 2)) continue;
      const int sps = (ksteps + sp - 1) / sp;
      if ((ksteps + sps - 1) / sps != sp) continue;
      if (sp > 1 && (size_t)sp * M * k * sizeof(float) > kTuneWsCap) continue;
      out.push_back(Plan{ci, sp, sps});
    }
  return out;
}

bool lookup_plan(const ShapeKey& key, Plan* pl) {
  std::lock_guard<std::mutex> lock(g_plan_mutex);
  auto it = g_plan_cache.find(key);
  if (it == g_plan_cache.end()) return false;
  *pl = it->second;
  return true;
}

// ---- per-dispatch timing (frcnn_conv2d_profile_begin / _end) -----------------------------------------------------------
// While a profile is open every kernel of frcnn_conv2d_fwd is launched through hipExtLaunchKernelGGL with its own
// start / stop events: the pair brackets THAT dispatch on the launch stream (begin -> end of the kernel, the quantity
// rocprofv3 --kernel-trace reports), without the event-packet overhead two separately recorded events add around a
// launch.  bench.py's `roofline` is computed from these durations.
struct ProfRec {
  hipEvent_t e0, e1;
  int call, kind;      // frcnn_conv2d_fwd call number since profile_begin; kind 0 = main kernel, 1 = split-K second pass
};
std::vector<ProfRec> g_prof;
std::atomic<bool> g_prof_on{false};
int g_prof_call = -1;

bool prof_events(int kind, hipEvent_t* e0, hipEvent_t* e1, hipStream_t stream) {
  if (!g_prof_on) return false;
  // a capturing stream cannot take the timed launch form (events would become graph nodes): plain launch, no record
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return false;
  if (hipEventCreate(e0) != hipSuccess) return false;
  if (hipEventCreate(e1) != hipSuccess) { (void)hipEventDestroy(*e0); return false; }
  g_prof.push_back(ProfRec{*e0, *e1, g_prof_call, kind});
  return true;
}

template <int WM, int WN, int TM, int TN, bool ALIGNED, bool WINO = false>
int launch_conv(const ConvParams& p, int splits, int groups, hipStream_t stream) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr size_t lds = (size_t)2 * (BM + BN) * LDS_PITCH * sizeof(float);
  static std::atomic<bool> configured{false};   // idempotent attribute call: a race only repeats it
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32<WM, WN, TM, TN, ALIGNED, WINO>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return frcnn::fail(FRCNN_ERR_LAUNCH, "conv: set LDS size: %s", hipGetErrorString(e));
    configured = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, groups, splits);
  hipEvent_t e0, e1;
  if (prof_events(0, &e0, &e1, stream))
    hipExtLaunchKernelGGL((conv_igemm_f32<WM, WN, TM, TN, ALIGNED, WINO>), grid, dim3(64 * WM * WN), (uint32_t)lds, stream, e0,
                          e1, 0, p);
  else
    hipLaunchKernelGGL((conv_igemm_f32<WM, WN, TM, TN, ALIGNED, WINO>), grid, dim3(64 * WM * WN), lds, stream, p);
  return frcnn::check_launch("conv_igemm_f32");
}

template <int WM, int WN>
int launch_conv_dma(const ConvParams& p, int splits, int groups, hipStream_t stream) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr size_t lds = (size_t)3 * (BM + BN) * 32 * sizeof(float);
  static std::atomic<bool> configured{false};   // idempotent attribute call: a race only repeats it
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_dma_f32<WM, WN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return frcnn::fail(FRCNN_ERR_LAUNCH, "conv: set LDS size: %s", hipGetErrorString(e));
    configured = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, groups, splits);
  hipEvent_t e0, e1;
  if (prof_events(0, &e0, &e1, stream))
    hipExtLaunchKernelGGL((conv_igemm_dma_f32<WM, WN>), grid, dim3(512), (uint32_t)lds, stream, e0, e1, 0, p);
  else
    hipLaunchKernelGGL((conv_igemm_dma_f32<WM, WN>), grid, dim3(512), lds, stream, p);
  return frcnn::check_launch("conv_igemm_dma_f32");
}

template <int WM, int WN, int TM, int TN>
int launch_conv_buf(const ConvParams& p, int splits, int groups, hipStream_t stream) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr size_t lds = (size_t)3 * (BM + BN) * 32 * sizeof(float);
  static std::atomic<bool> configured{false};   // idempotent attribute call: a race only repeats it
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_buf_f32<WM, WN, TM, TN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return frcnn::fail(FRCNN_ERR_LAUNCH, "conv: set LDS size: %s", hipGetErrorString(e));
    configured = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, groups, splits);
  hipEvent_t e0, e1;
  if (prof_events(0, &e0, &e1, stream))
    hipExtLaunchKernelGGL((conv_igemm_buf_f32<WM, WN, TM, TN>), grid, dim3(64 * WM * WN), (uint32_t)lds, stream, e0, e1, 0, p);
  else
    hipLaunchKernelGGL((conv_igemm_buf_f32<WM, WN, TM, TN>), grid, dim3(64 * WM * WN), lds, stream, p);
  return frcnn::check_launch("conv_igemm_buf_f32");
}

int launch_conv_pbuf(const ConvParams& p, int splits, int groups, hipStream_t stream) {
  constexpr size_t lds = ((size_t)3 * (64 + 64) * 32 + (size_t)4 * 32 * LDS_PITCH) * sizeof(float);
  static std::atomic<bool> configured{false};   // idempotent attribute call: a race only repeats it
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_pbuf_f32),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return frcnn::fail(FRCNN_ERR_LAUNCH, "conv: set LDS size: %s", hipGetErrorString(e));
    configured = true;
  }
  const long total = (long)p.tiles_m * p.tiles_n * groups * splits;
  dim3 grid((unsigned)std::min<long>(total, 2L * NUM_CU));      // two resident workgroups per CU (66 KB of LDS each)
  hipEvent_t e0, e1;
  if (prof_events(0, &e0, &e1, stream))
    hipExtLaunchKernelGGL(conv_igemm_pbuf_f32, grid, dim3(256), (uint32_t)lds, stream, e0, e1, 0, p, groups, splits);
  else
    hipLaunchKernelGGL(conv_igemm_pbuf_f32, grid, dim3(256), lds, stream, p, groups, splits);
  return frcnn::check_launch("conv_igemm_pbuf_f32");
}

int launch_conv_dma2(const ConvParams& p, int splits, int groups, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * (128 + 128) * 32 * sizeof(float);
  static std::atomic<bool> configured{false};   // idempotent attribute call: a race only repeats it
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_dma2_f32<2, 2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return frcnn::fail(FRCNN_ERR_LAUNCH, "conv: set LDS size: %s", hipGetErrorString(e));
    configured = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, groups, splits);
  hipEvent_t e0, e1;
  if (prof_events(0, &e0, &e1, stream))
    hipExtLaunchKernelGGL((conv_igemm_dma2_f32<2, 2>), grid, dim3(256), (uint32_t)lds, stream, e0, e1, 0, p);
  else
    hipLaunchKernelGGL((conv_igemm_dma2_f32<2, 2>), grid, dim3(256), lds, stream, p);
  return frcnn::check_launch("conv_igemm_dma2_f32");
}

// tuning hook: 0 = register-staged kernels only, 1 = LDS-DMA kernel for the 8-wave tiles when C % 32 == 0
std::atomic<int> g_use_dma{1};

bool conv_args_ok(int n, int h, int w, int c, int k, int r, int s, int stride, int pad) {
  return n > 0 && h > 0 && w > 0 && c > 0 && (c % 4) == 0 && k > 0 && r > 0 && s > 0 && stride > 0 && pad >= 0 &&
         (h + 2 * pad - r) >= 0 && (w + 2 * pad - s) >= 0;
}

}  // namespace

extern "C" int frcnn_conv2d_set_tile(int tm, int tn) {
  bool known = (tm == 0 && tn == 0);
  for (int i = 0; i < kNumTiles; ++i) known = known || (kTiles[i].tm == tm && kTiles[i].tn == tn);
  FRCNN_REQUIRE(known, "conv2d_set_tile: tiles are 64*tm x 64*tn with (tm,tn) in "
                       "{(4,2),(2,4),(2,2),(2,1),(1,2),(1,1)} ((0,0) = automatic)");
  g_force_tm = tm;
  g_force_tn = tn;
  return FRCNN_OK;
}

extern "C" int frcnn_conv2d_set_algo(int mode) {
  FRCNN_REQUIRE(mode >= 0 && (mode & 3) <= 2 && (mode & ~(3 | 16 | 32 | 64)) == 0,
                "conv2d_set_algo: mode %d (0 auto, 1 implicit GEMM only, 2 Winograd where it applies; +16: never fuse the "
                "Winograd input transform into the GEMM, +32: forced Winograd uses the 64x64 GEMM with the fused transform, +64: the "
                "register-staged kernels store straight from the MFMA layout instead of through the LDS transpose)", mode);
  g_algo_mode = mode & 3;
  g_wino_fuse = (mode & 16) ? 0 : ((mode & 32) ? 2 : 1);
  g_epi_lds = (mode & 64) ? 0 : 1;
  return FRCNN_OK;
}

extern "C" int frcnn_conv2d_set_staging(int use_lds_dma) {
  FRCNN_REQUIRE(use_lds_dma >= 0 && use_lds_dma <= 3, "conv2d_set_staging: mode %d (0 .. 3)", use_lds_dma);
  g_use_dma = use_lds_dma;
  return FRCNN_OK;
}

extern "C" size_t frcnn_conv2d_fwd_ws_bytes(int n, int h, int w, int c, int k, int r, int s, int stride, int pad,
                                            int split_k) {
  if (!conv_args_ok(n, h, w, c, k, r, s, stride, pad)) return 0;
  const int ho = (h + 2 * pad - r) / stride + 1, wo = (w + 2 * pad - s) / stride + 1;
  const long M = (long)n * ho * wo;
  const int ksteps = (r * s * c + BK - 1) / BK;
  // a Winograd plan can only be chosen for a call without a residual; the caller does not say here whether it has one,
  // so the eligible shapes get room for it whenever it may be picked
  const bool wino = winograd_ok(r, s, stride, pad, c, k, 1) && g_algo_mode != 1;
  const size_t wino_bytes = wino ? winograd_ws_bytes(n, h, w, c, k) : 0;
  if (split_k <= 0 && g_force_tm == 0) {
    // the caller does not say whether it has a residual: room for the cached plan of either kind of call.  A shape with
    // only one of the two cached keeps room for whatever the other may still be tuned to (below)
    Plan c0, c1;
    const bool h0 = lookup_plan(shape_key(n, h, w, c, k, r, s, stride, pad, 1, false), &c0);
    const bool h1 = lookup_plan(shape_key(n, h, w, c, k, r, s, stride, pad, 1, true), &c1);
    if (h0 || h1) {
      size_t need = 0;
      for (const Plan* pc : {h0 ? &c0 : nullptr, h1 ? &c1 : nullptr}) {
        if (!pc) continue;
        need = std::max(need, pc->algo == 1 ? wino_bytes : (pc->splits > 1 ? (size_t)pc->splits * M * k * sizeof(float) : 0));
      }
      if (g_algo_mode == 2) need = std::max(need, wino_bytes);
      if (h0 && h1) return need;
      if (!g_autotune) return need;
      size_t more = wino_bytes <= kTuneWinoCap ? wino_bytes : 0;
      for (const Plan& cand : tune_candidates(M, k, ksteps, true))
        if (cand.splits > 1) more = std::max(more, (size_t)cand.splits * M * k * sizeof(float));
      return std::max(need, more);
    }
  }
  if (split_k <= 0 && g_force_tm == 0 && g_autotune) {
    size_t need = wino_bytes <= kTuneWinoCap ? wino_bytes : 0;   // not tuned yet: room for every candidate the tuner may try
    for (const Plan& cand : tune_candidates(M, k, ksteps, true))
      if (cand.splits > 1) need = std::max(need, (size_t)cand.splits * M * k * sizeof(float));
    return need;
  }
  if (g_algo_mode == 2 && split_k <= 0 && g_force_tm == 0 && wino) return wino_bytes;
  const Plan pl = choose_plan((int)M, k, ksteps, split_k);
  return pl.splits > 1 ? (size_t)pl.splits * M * k * sizeof(float) : 0;
}

extern "C" int frcnn_conv2d_plan_algo(int n, int h, int w, int c, int k, int r, int s, int stride, int pad, int has_residual) {
  Plan pl;
  if (!lookup_plan(shape_key(n, h, w, c, k, r, s, stride, pad, 1, has_residual != 0), &pl)) return -1;
  return pl.algo;
}

extern "C" int frcnn_conv2d_set_autotune(int enable) {
  FRCNN_REQUIRE(enable >= 0 && enable <= 2, "conv2d_set_autotune: 0 off, 1 time each candidate alone, 2 time it under load");
  g_autotune = enable;
  return FRCNN_OK;
}

bool frcnn::autotune_enabled() { return g_autotune != 0; }

unsigned long long frcnn::conv_settings_word() {
  return (unsigned long long)g_algo_mode.load() | ((unsigned long long)g_wino_fuse.load() << 4) |
         ((unsigned long long)g_epi_lds.load() << 8) | ((unsigned long long)g_use_dma.load() << 12) |
         ((unsigned long long)g_force_tm.load() << 16) | ((unsigned long long)g_force_tn.load() << 24);
}

extern "C" int frcnn_conv2d_clear_plans(void) {
  {
    std::lock_guard<std::mutex> lock(g_plan_mutex);
    g_plan_cache.clear();
  }
  frcnn::clear_wgrad_plans();
  return FRCNN_OK;
}

// Plan table as plain ints, 13 per entry: the 10-int shape key, then tile index (+ 16 for a Winograd plan), splits,
// steps per split.
extern "C" int frcnn_conv2d_export_plans(int* out, int capacity_entries) {
  std::lock_guard<std::mutex> lock(g_plan_mutex);
  int n = 0;
  for (const auto& kv : g_plan_cache) {
    if (out && n < capacity_entries) {
      for (int i = 0; i < 10; ++i) out[n * 13 + i] = kv.first[i];
      out[n * 13 + 10] = kv.second.cfg + 16 * (kv.second.algo + kv.second.fuse_in);   // 2 = Winograd with the fused input transform
      out[n * 13 + 11] = kv.second.splits;
      out[n * 13 + 12] = kv.second.steps_per_split;
    }
    ++n;
  }
  return n;   // entries in the cache (may exceed capacity_entries: call again with a larger buffer)
}

extern "C" int frcnn_conv2d_import_plans(const int* in, int entries) {
  FRCNN_REQUIRE(in && entries >= 0, "conv2d_import_plans: null table");
  std::lock_guard<std::mutex> lock(g_plan_mutex);
  for (int e = 0; e < entries; ++e) {
    const int* row = in + e * 13;
    const int code = row[10] >> 4, cfg = row[10] & 15;
    const int algo = code >= 1 ? 1 : 0, fuse_in = code == 2 ? 1 : 0;
    FRCNN_REQUIRE(row[10] >= 0 && cfg < kNumTiles && code <= 2 && (!fuse_in || (cfg == 5 && row[3] % BK == 0)) && row[11] >= 1 && row[11] <= 64 && row[12] >= 1 &&
                      (algo == 0 || (row[11] == 1 && winograd_ok(row[5], row[6], row[7], row[8], row[3], row[4], row[9]))),   // a residual key (row[9] >= 256) fails winograd_ok: no Winograd plan for it
                  "conv2d_import_plans: entry %d is not a valid plan (tile %d, splits %d)", e, row[10], row[11]);
    ShapeKey key;
    for (int i = 0; i < 10; ++i) key[i] = row[i];
    Plan pl{cfg, row[11], row[12]};
    pl.algo = algo;
    pl.fuse_in = fuse_in;
    g_plan_cache[key] = pl;
  }
  return FRCNN_OK;
}

namespace {
// ------------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) for the 3x3 / stride 1 / pad 1 layers with large GEMMs (layer4's conv2 on the 300 RoIs, the RPN
// 3x3): Y = A^T [ (G g G^T) . (B^T d B) ] A per 2x2 output tile, i.e. 16 independent GEMMs over (tiles x C) x (C x K)
// instead of one over (pixels x 9C) x (9C x K): 2.25x fewer multiplications (1.72x on a 7x7 map, whose 4x4 tiles cover
// 8x8).  Same fp32 arithmetic type; the transforms only add and halve, and the reduction is 9x shorter, so the
// rounding error is that of the direct form or smaller (tests/test_gpu_parity.py compares both with float64).
// Four launches: filter transform (stateless ABI: recomputed per call, 16 KC floats), input transform, ONE grouped
// launch of the implicit-GEMM kernels as a 1x1 convolution (blockIdx.y = transform component), output transform with
// the BatchNorm scale / shift and ReLU.  Only the autotuner selects it (choose_plan never does).
// ------------------------------------------------------------------------------------------------
bool winograd_ok(int r, int s, int stride, int pad, int c, int k, int out_stride) {
  return r == 3 && s == 3 && stride == 1 && pad == 1 && (c % 4) == 0 && (k % 4) == 0 && out_stride == 1;
}

struct WinoGeom {
  int th, tw;        // 2x2 output tiles per image
  long T;            // tiles in the batch
  size_t u_off, v_off, m_off, bytes;   // workspace layout (bytes)
};
WinoGeom wino_geom(int n, int h, int w, int c, int k) {
  WinoGeom g;
  g.th = (h + 1) / 2;
  g.tw = (w + 1) / 2;
  g.T = (long)n * g.th * g.tw;
  g.u_off = 0;
  g.v_off = frcnn::align_up((size_t)16 * k * c * sizeof(float), 256);
  g.m_off = g.v_off + frcnn::align_up((size_t)16 * g.T * c * sizeof(float), 256);
  g.bytes = g.m_off + frcnn::align_up((size_t)16 * g.T * k * sizeof(float), 256);
  return g;
}

size_t winograd_ws_bytes(int n, int h, int w, int c, int k) { return wino_geom(n, h, w, c, k).bytes; }

// U[i*4+j][k][c] = (G g G^T)[i][j],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]];  one thread per (k, 4 channels)
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int K,
                                                         int C4) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)K * C4) return;
  const int c4 = (int)(idx % C4);
  const int k = (int)(idx / C4);
  const f32x4* src = reinterpret_cast<const f32x4*>(w) + (size_t)k * 9 * C4 + c4;
  f32x4 g[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 3; ++q) g[r][q] = src[(size_t)(r * 3 + q) * C4];
  f32x4 t[4][3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    t[0][q] = g[0][q];
    t[1][q] = (g[0][q] + g[1][q] + g[2][q]) * 0.5f;
    t[2][q] = (g[0][q] - g[1][q] + g[2][q]) * 0.5f;
    t[3][q] = g[2][q];
  }
  f32x4* dst = reinterpret_cast<f32x4*>(U) + (size_t)k * C4 + c4;
  const size_t plane = (size_t)K * C4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dst[(size_t)(i * 4 + 0) * plane] = t[i][0];
    dst[(size_t)(i * 4 + 1) * plane] = (t[i][0] + t[i][1] + t[i][2]) * 0.5f;
    dst[(size_t)(i * 4 + 2) * plane] = (t[i][0] - t[i][1] + t[i][2]) * 0.5f;
    dst[(size_t)(i * 4 + 3) * plane] = t[i][2];
  }
}

// V[i*4+j][t][c] = (B^T d B)[i][j] of the 4x4 input patch of tile t (rows 2ty-1.., cols 2tx-1.., zero outside the map);
// B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]].  One thread per (tile, 4 channels): lanes run along the channels.
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int H, int W,
                                                        int C4, int th, int tw, long T) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)T * C4) return;
  const int c4 = (int)(idx % C4);
  const long t = (long)(idx / C4);
  const int tx = (int)(t % tw);
  const long t2 = t / tw;
  const int ty = (int)(t2 % th);
  const int n = (int)(t2 / th);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int hi = 2 * ty - 1 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int wi = 2 * tx - 1 + j;
      const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      d[i][j] = ok ? reinterpret_cast<const f32x4*>(x)[((size_t)(n * H + hi) * W + wi) * C4 + c4] : zero;
    }
  }
  f32x4 r[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[0][j] = d[0][j] - d[2][j];
    r[1][j] = d[1][j] + d[2][j];
    r[2][j] = d[2][j] - d[1][j];
    r[3][j] = d[1][j] - d[3][j];
  }
  f32x4* dst = reinterpret_cast<f32x4*>(V) + (size_t)t * C4 + c4;
  const size_t plane = (size_t)T * C4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dst[(size_t)(i * 4 + 0) * plane] = r[i][0] - r[i][2];
    dst[(size_t)(i * 4 + 1) * plane] = r[i][1] + r[i][2];
    dst[(size_t)(i * 4 + 2) * plane] = r[i][2] - r[i][1];
    dst[(size_t)(i * 4 + 3) * plane] = r[i][1] - r[i][3];
  }
}

// y[2ty+a][2tx+b] = act((A^T m A)[a][b] * scale + shift),  A^T = [[1,1,1,0],[0,1,-1,-1]];  one thread per (tile, 4 channels)
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mo, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ y, int H,
                                                         int W, int K4, int th, int tw, long T, int relu,
                                                         const float* __restrict__ mask,
                                                         const float* __restrict__ mscale) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)T * K4) return;
  const int k4 = (int)(idx % K4);
  const long t = (long)(idx / K4);
  const int tx = (int)(t % tw);
  const long t2 = t / tw;
  const int ty = (int)(t2 % th);
  const int n = (int)(t2 / th);
  const f32x4* src = reinterpret_cast<const f32x4*>(Mo) + (size_t)t * K4 + k4;
  const size_t plane = (size_t)T * K4;
  f32x4 s0[4], s1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 m0 = src[(size_t)(0 * 4 + j) * plane], m1 = src[(size_t)(1 * 4 + j) * plane];
    const f32x4 m2 = src[(size_t)(2 * 4 + j) * plane], m3 = src[(size_t)(3 * 4 + j) * plane];
    s0[j] = m0 + m1 + m2;
    s1[j] = m1 - m2 - m3;
  }
  f32x4 o[2][2];
  o[0][0] = s0[0] + s0[1] + s0[2];
  o[0][1] = s0[1] - s0[2] - s0[3];
  o[1][0] = s1[0] + s1[1] + s1[2];
  o[1][1] = s1[1] - s1[2] - s1[3];
  const f32x4 sc = scale ? reinterpret_cast<const f32x4*>(scale)[k4] : f32x4{1.f, 1.f, 1.f, 1.f};
  const f32x4 sh = shift ? reinterpret_cast<const f32x4*>(shift)[k4] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int ho = 2 * ty + a;
    if (ho >= H) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int wo = 2 * tx + b;
      if (wo >= W) continue;
      f32x4 v = o[a][b] * sc + sh;
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      const size_t at = ((size_t)(n * H + ho) * W + wo) * K4 + k4;
      if (mask) {
        const f32x4 mv = reinterpret_cast<const f32x4*>(mask)[at];
        const f32x4 ms = mscale ? reinterpret_cast<const f32x4*>(mscale)[k4] : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = mv[e] > 0.f ? v[e] * ms[e] : 0.f;
      }
      reinterpret_cast<f32x4*>(y)[at] = v;
    }
  }
}

// launch one plan: main kernel + the split-K second pass (or the four launches of a Winograd plan)
int launch_gemm(ConvParams p, const Plan& pl, long M, int k, int groups, hipStream_t stream) {
  const TileCfg& tc = kTiles[pl.cfg];
  p.steps_per_split = pl.steps_per_split;
  const int bm = 64 * tc.tm, bn = 64 * tc.tn;
  p.tiles_m = (p.M + bm - 1) / bm;
  p.tiles_n = (k + bn - 1) / bn;
  const bool aligned = (p.C % BK) == 0;
  int rc;
  if (pl.fuse_in) {
    if (pl.cfg != 5 || !aligned) return frcnn::fail(FRCNN_ERR_ARG, "conv2d: fused Winograd input needs the 64x64 tile and C %% 32 == 0");
    return launch_conv<2, 2, 1, 1, true, true>(p, 1, groups, stream);
  }
#define FRCNN_CONV_CASE(WM_, WN_, TM_, TN_)                                           \
  rc = aligned ? launch_conv<WM_, WN_, TM_, TN_, true>(p, pl.splits, groups, stream) \
               : launch_conv<WM_, WN_, TM_, TN_, false>(p, pl.splits, groups, stream)
  switch (pl.cfg) {
    case 0:
      if (aligned && g_use_dma == 3 && p.xbytes && p.wbytes) rc = launch_conv_buf<4, 2, 2, 2>(p, pl.splits, groups, stream);
      else if (aligned && g_use_dma) rc = launch_conv_dma<4, 2>(p, pl.splits, groups, stream);
      else FRCNN_CONV_CASE(4, 2, 2, 2);
      break;
    case 1:
      if (aligned && g_use_dma == 3 && p.xbytes && p.wbytes) rc = launch_conv_buf<2, 4, 2, 2>(p, pl.splits, groups, stream);
      else if (aligned && g_use_dma) rc = launch_conv_dma<2, 4>(p, pl.splits, groups, stream);
      else FRCNN_CONV_CASE(2, 4, 2, 2);
      break;
    case 2:
      if (aligned && g_use_dma == 3 && p.xbytes && p.wbytes) rc = launch_conv_buf<2, 2, 2, 2>(p, pl.splits, groups, stream);
      else if (aligned && g_use_dma == 2) rc = launch_conv_dma2(p, pl.splits, groups, stream);   // test hook, see set_staging
      else FRCNN_CONV_CASE(2, 2, 2, 2);
      break;
    case 6:
      if (aligned && g_use_dma) rc = launch_conv_dma2(p, pl.splits, groups, stream);
      else FRCNN_CONV_CASE(2, 2, 2, 2);
      break;
    case 13:
      if (aligned && g_use_dma && p.xbytes && p.wbytes) rc = launch_conv_pbuf(p, pl.splits, groups, stream);
      else FRCNN_CONV_CASE(2, 2, 1, 1);
      break;
    case 10:
      if (aligned && g_use_dma && p.xbytes && p.wbytes) rc = launch_conv_buf<2, 2, 2, 2>(p, pl.splits, groups, stream);
      else FRCNN_CONV_CASE(2, 2, 2, 2);
      break;
    case 11:
      if (aligned && g_use_dma && p.xbytes && p.wbytes) rc = launch_conv_buf<4, 2, 2, 2>(p, pl.splits, groups, stream);
      else FRCNN_CONV_CASE(4, 2, 2, 2);
      break;
    case 12:
      if (aligned && g_use_dma && p.xbytes && p.wbytes) rc = launch_conv_buf<2, 4, 2, 2>(p, pl.splits, groups, stream);
      else FRCNN_CONV_CASE(2, 4, 2, 2);
      break;
    case 3:
      if (aligned && g_use_dma == 3 && p.xbytes && p.wbytes) rc = launch_conv_buf<2, 2, 2, 1>(p, pl.splits, groups, stream);
      else FRCNN_CONV_CASE(2, 2, 2, 1);
      break;
    case 4:
      if (aligned && g_use_dma == 3 && p.xbytes && p.wbytes) rc = launch_conv_buf<2, 2, 1, 2>(p, pl.splits, groups, stream);
      else FRCNN_CONV_CASE(2, 2, 1, 2);
      break;
    case 7:
      if (aligned && g_use_dma && p.xbytes && p.wbytes) rc = launch_conv_buf<2, 2, 1, 1>(p, pl.splits, groups, stream);
      else FRCNN_CONV_CASE(2, 2, 1, 1);
      break;
    case 8:
      if (aligned && g_use_dma && p.xbytes && p.wbytes) rc = launch_conv_buf<2, 2, 2, 1>(p, pl.splits, groups, stream);
      else FRCNN_CONV_CASE(2, 2, 2, 1);
      break;
    case 9:
      if (aligned && g_use_dma && p.xbytes && p.wbytes) rc = launch_conv_buf<2, 2, 1, 2>(p, pl.splits, groups, stream);
      else FRCNN_CONV_CASE(2, 2, 1, 2);
      break;
    default:
      if (aligned && g_use_dma == 3 && p.xbytes && p.wbytes) rc = launch_conv_buf<2, 2, 1, 1>(p, pl.splits, groups, stream);
      else FRCNN_CONV_CASE(2, 2, 1, 1);
      break;
  }
#undef FRCNN_CONV_CASE
  return rc;
}

// launches a 1-D grid kernel through the profiling events when a profile is open (kind 2 = Winograd transform)
template <typename... Args, typename... Actual>
int launch_1d(const char* what, void (*kernel)(Args...), size_t threads, hipStream_t stream, Actual... args) {
  const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
  hipEvent_t e0, e1;
  if (prof_events(2, &e0, &e1, stream)) hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, e0, e1, 0, args...);
  else hipLaunchKernelGGL(kernel, grid, block, 0, stream, args...);
  return frcnn::check_launch(what);
}

int launch_winograd(const ConvParams& p, const Plan& pl, const float* scale, const float* shift, float* y, int relu,
                    void* ws, hipStream_t stream) {
  const int n = p.M / (p.Ho * p.Wo);
  const WinoGeom g = wino_geom(n, p.H, p.W, p.C, p.K);
  char* base = static_cast<char*>(ws);
  float* U = reinterpret_cast<float*>(base + g.u_off);
  float* V = reinterpret_cast<float*>(base + g.v_off);
  float* Mo = reinterpret_cast<float*>(base + g.m_off);
  int rc = FRCNN_OK;
  if (p.u_pre) U = const_cast<float*>(p.u_pre);   // read-only from here on
  else rc = launch_1d("wino_filter_kernel", wino_filter_kernel, (size_t)p.K * (p.C / 4), stream, p.w, U, p.K, p.C / 4);
  if (rc != FRCNN_OK) return rc;
  if (!pl.fuse_in)
    rc = launch_1d("wino_input_kernel", wino_input_kernel, (size_t)g.T * (p.C / 4), stream, p.x, V, p.H, p.W, p.C / 4, g.th,
                   g.tw, g.T);
  if (rc != FRCNN_OK) return rc;
  // 16 GEMMs  Mo[xi] (T x K) = V[xi] (T x C) . U[xi]^T (K x C)  as ONE grouped 1x1 convolution over a 1 x T "image"
  ConvParams q;
  q.x = V; q.w = U; q.scale = nullptr; q.shift = nullptr; q.res = nullptr; q.y = Mo; q.partial = nullptr;
  q.H = 1; q.W = (int)g.T; q.C = p.C; q.K = p.K; q.R = 1; q.S = 1; q.stride = 1; q.pad = 0; q.Ho = 1; q.Wo = (int)g.T;
  q.M = (int)g.T;
  q.Ktot = p.C;
  q.ksteps = (p.C + BK - 1) / BK;
  q.relu = 0;
  q.ys = 1; q.Hy = 0; q.Wy = 0;
  q.gx = (size_t)g.T * p.C; q.gw = (size_t)p.K * p.C; q.gy = (size_t)g.T * p.K;
  q.u_pre = nullptr;
  q.mask = q.mscale = nullptr;
  q.wiH = q.wiW = q.wth = q.wtw = 0;
  q.epi_lds = p.epi_lds;
  q.zero = p.zero;
  {
    const size_t xb = q.gx * sizeof(float), wb = q.gw * sizeof(float);   // one transform component's slice
    q.xbytes = xb < ((size_t)1 << 31) ? (unsigned)xb : 0;
    q.wbytes = wb < ((size_t)1 << 31) ? (unsigned)wb : 0;
  }
  Plan gp{pl.cfg, 1, q.ksteps};
  if (pl.fuse_in) {          // the GEMM reads the layer's input itself: no V tensor
    q.x = p.x;
    q.gx = 0;
    q.wiH = p.H; q.wiW = p.W; q.wth = g.th; q.wtw = g.tw;
    gp.fuse_in = 1;
  }
  rc = launch_gemm(q, gp, g.T, p.K, 16, stream);
  if (rc != FRCNN_OK) return rc;
  return launch_1d("wino_output_kernel", wino_output_kernel, (size_t)g.T * (p.K / 4), stream, (const float*)Mo, scale, shift, y,
                   p.Ho, p.Wo, p.K / 4, g.th, g.tw, g.T, relu, p.mask, p.mscale);
}

int launch_plan(ConvParams p, const Plan& pl, long M, int k, const float* scale, const float* shift,
                const float* residual, float* y, int relu, void* ws, hipStream_t stream) {
  if (pl.algo == 1) return launch_winograd(p, pl, scale, shift, y, relu, ws, stream);
  p.partial = pl.splits > 1 ? static_cast<float*>(ws) : nullptr;
  int rc = launch_gemm(p, pl, M, k, 1, stream);
  if (rc != FRCNN_OK) return rc;
  if (pl.splits > 1) {
    const size_t mk = (size_t)M * k;
    const int blocks = (int)std::min<size_t>((mk + 255) / 256, 2048);
    hipEvent_t e0, e1;
    if (prof_events(1, &e0, &e1, stream))
      hipExtLaunchKernelGGL(conv_splitk_epilogue, dim3(blocks), dim3(256), 0, stream, e0, e1, 0,
                            (const float*)p.partial, pl.splits, mk, k, scale, shift, residual, y, relu, p.mask, p.mscale);
    else
      hipLaunchKernelGGL(conv_splitk_epilogue, dim3(blocks), dim3(256), 0, stream, p.partial, pl.splits, mk, k,
                         scale, shift, residual, y, relu, p.mask, p.mscale);
    return frcnn::check_launch("conv_splitk_epilogue");
  }
  return FRCNN_OK;
}

size_t plan_ws_bytes(const Plan& pl, const ConvParams& p, long M, int k) {
  if (pl.algo == 1) return wino_geom(p.M / (p.Ho * p.Wo), p.H, p.W, p.C, p.K).bytes;
  return pl.splits > 1 ? (size_t)pl.splits * M * k * sizeof(float) : 0;
}

// frcnn_conv2d_set_autotune(2): candidates are timed UNDER LOAD - kLoadCopies launches of the candidate in flight at once,
// one per stream (the caller's + three of the library's own).  The product keeps four frames in flight on four streams
// (model/frame_graph.FramePool), where what counts is the chip time a plan takes away from the other frames' kernels, not
// the latency of one launch on an idle chip: timed alone, a grid of many small tiles that fills 256 CUs once beats the
// larger tiles whose matrix pipe runs at 1.5x the efficiency; with four copies competing the ranking is by throughput.
// The copies read and write the SAME tensors: they compute identical values, so the races are between equal stores.
constexpr int kLoadCopies = 4;
struct LoadStreams {
  hipStream_t s[kLoadCopies - 1];
  hipEvent_t done[kLoadCopies - 1];
  bool ok = false;
};
LoadStreams& load_streams() {
  static LoadStreams ls;
  static std::once_flag once;
  std::call_once(once, [] {
    bool ok = true;
    for (int j = 0; j < kLoadCopies - 1 && ok; ++j)
      ok = hipStreamCreateWithFlags(&ls.s[j], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&ls.done[j], hipEventDisableTiming) == hipSuccess;
    ls.ok = ok;
  });
  return ls;
}

// time every candidate plan on the caller's tensors; returns false when tuning is not possible here
bool tune_plan(const ConvParams& p, long M, int k, const float* scale, const float* shift, const float* residual,
               float* y, int relu, void* ws, size_t ws_bytes, hipStream_t stream, bool allow_split, bool wino, Plan* best) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return false;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess) return false;
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return false; }
  // two passes over the candidates, each keeps its best time: a one-off disturbance (clock ramp, a neighbour stream)
  // can then neither crown a slow plan nor bury the fast one
  std::vector<Plan> cands;
  if (g_algo_mode != 2 || !wino) cands = tune_candidates(M, k, p.ksteps, allow_split);
  if (wino && g_algo_mode != 1) {
    // Winograd around the grouped GEMM, one candidate per GEMM tile that makes sense for (tiles x C) x (C x K)
    for (int cfg = 0; cfg < kNumTiles; ++cfg) {
      Plan pl{cfg, 1, (p.C + BK - 1) / BK};
      pl.algo = 1;
      cands.push_back(pl);
    }
    if ((p.C % BK) == 0 && g_wino_fuse) {   // the 64x64 GEMM with the input transform in its A-tile load
      Plan pl{5, 1, p.C / BK};
      pl.algo = 1;
      pl.fuse_in = 1;
      cands.push_back(pl);
    }
  }
  // Level 1: two passes over all candidates, each timed alone; a candidate keeps its best time (a one-off disturbance can
  // neither crown a slow plan nor bury the fast one).  Level 2: one such pass, then the kLoadFinalists fastest candidates are
  // timed under load (two passes) and ranked by that - every candidate under load would take 4x the tuning time for plans
  // that are already 1.3x off alone.
  std::vector<float> best_of(cands.size(), 1e30f);
  LoadStreams* ls = nullptr;
  if (g_autotune == 2) {
    ls = &load_streams();
    if (!ls->ok) ls = nullptr;     // no extra streams: fall back to timing alone
  }
  auto time_candidate = [&](size_t ci, bool warm, bool loaded) -> float {
    const Plan& pl = cands[ci];
    const size_t need = plan_ws_bytes(pl, p, M, k);
    if (need > ws_bytes || (need > 0 && !ws)) return 1e30f;
    if (warm && launch_plan(p, pl, M, k, scale, shift, residual, y, relu, ws, stream) != FRCNN_OK) return 1e30f;
    (void)hipEventRecord(e0, stream);
    const int reps = 3;
    bool ok = true;
    if (loaded) {
      for (int j = 0; j < kLoadCopies - 1; ++j) (void)hipStreamWaitEvent(ls->s[j], e0, 0);
      for (int i = 0; i < reps && ok; ++i) {
        ok = launch_plan(p, pl, M, k, scale, shift, residual, y, relu, ws, stream) == FRCNN_OK;
        for (int j = 0; j < kLoadCopies - 1 && ok; ++j)
          ok = launch_plan(p, pl, M, k, scale, shift, residual, y, relu, ws, ls->s[j]) == FRCNN_OK;
      }
      for (int j = 0; j < kLoadCopies - 1; ++j) {      // the caller's stream ends the region when every copy is done
        (void)hipEventRecord(ls->done[j], ls->s[j]);
        (void)hipStreamWaitEvent(stream, ls->done[j], 0);
      }
    } else {
      for (int i = 0; i < reps && ok; ++i) ok = launch_plan(p, pl, M, k, scale, shift, residual, y, relu, ws, stream) == FRCNN_OK;
    }
    (void)hipEventRecord(e1, stream);
    if (hipEventSynchronize(e1) != hipSuccess || !ok) return 1e30f;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return 1e30f;
    return ms;
  };
  for (int pass = 0; pass < (ls ? 1 : 2); ++pass)
    for (size_t ci = 0; ci < cands.size(); ++ci) best_of[ci] = std::min(best_of[ci], time_candidate(ci, pass == 0, false));
  if (ls) {
    constexpr size_t kLoadFinalists = 6;
    std::vector<size_t> order(cands.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return best_of[a] < best_of[b]; });
    std::vector<float> loaded(cands.size(), 1e30f);
    for (int pass = 0; pass < 2; ++pass)
      for (size_t r = 0; r < std::min(kLoadFinalists, order.size()); ++r) {
        const size_t ci = order[r];
        if (best_of[ci] >= 1e30f) continue;
        loaded[ci] = std::min(loaded[ci], time_candidate(ci, false, true));
      }
    best_of = loaded;
  }
  float best_ms = 1e30f;
  bool found = false;
  for (size_t ci = 0; ci < cands.size(); ++ci)
    if (best_of[ci] < best_ms) { best_ms = best_of[ci]; *best = cands[ci]; found = true; }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return found;
}

// Shared driver of the forward entry point and of the data-gradient entry point (which is a forward
// convolution of dy with the flipped/transposed filter).  out_stride > 1 scatters the output pixels onto
// a (hy x wy) map at stride out_stride (the map must be zero-filled by the caller); split-K is disabled then.
int run_conv(const float* x, const float* wgt, const float* scale, const float* shift, const float* residual,
             float* y, int n, int h, int w, int c, int k, int r, int s, int stride, int pad, int relu, int split_k,
             void* ws, size_t ws_bytes, hipStream_t stream, int out_stride, int hy, int wy, const float* u_pre = nullptr,
             const float* mask = nullptr, const float* mscale = nullptr) {
  ConvParams p;
  p.u_pre = u_pre;
  p.mask = mask;
  p.mscale = mscale;
  p.x = x; p.w = wgt; p.scale = scale; p.shift = shift; p.res = residual; p.y = y; p.partial = nullptr;
  p.H = h; p.W = w; p.C = c; p.K = k; p.R = r; p.S = s; p.stride = stride; p.pad = pad;
  p.Ho = (h + 2 * pad - r) / stride + 1;
  p.Wo = (w + 2 * pad - s) / stride + 1;
  const long M = (long)n * p.Ho * p.Wo;
  FRCNN_REQUIRE(M * (long)k < (1L << 31) && (long)n * h * w * c < (1L << 31), "conv2d: tensor too large for int32 indexing");
  p.M = (int)M;
  p.Ktot = r * s * c;
  p.ksteps = (p.Ktot + BK - 1) / BK;
  p.relu = relu;
  p.ys = out_stride; p.Hy = hy; p.Wy = wy;
  p.steps_per_split = p.ksteps; p.tiles_m = p.tiles_n = 0;
  p.gx = p.gw = p.gy = 0;
  p.wiH = p.wiW = p.wth = p.wtw = 0;
  p.epi_lds = g_epi_lds;
  {
    const size_t xb = (size_t)n * h * w * c * sizeof(float), wb = (size_t)k * p.Ktot * sizeof(float);
    p.xbytes = xb < ((size_t)1 << 31) ? (unsigned)xb : 0;
    p.wbytes = wb < ((size_t)1 << 31) ? (unsigned)wb : 0;
  }
  p.zero = zero_page_address();
  if (!p.zero) return frcnn::fail(FRCNN_ERR_LAUNCH, "conv2d: cannot resolve the zero page's device address");
  FRCNN_REQUIRE((long)k * p.Ktot < (1L << 31), "conv2d: filter too large for int32 indexing");
  const bool allow_split = out_stride == 1;
  Plan pl;
  bool have = false;
  if (split_k <= 0 && g_force_tm == 0) {   // cached plans always apply; new shapes are tuned only in autotune mode
    const ShapeKey key = shape_key(n, h, w, c, k, r, s, stride, pad, out_stride, residual != nullptr);
    have = lookup_plan(key, &pl);
    const bool wino = residual == nullptr && winograd_ok(r, s, stride, pad, c, k, out_stride);
    // a cached plan of the other form than this call may use (a residual operand, or a forced mode) is left in the cache
    // for the calls it was tuned for; this call runs the analytic plan
    bool keep_cache = false;
    if (have && ((pl.algo == 1 && (!wino || g_algo_mode == 1)) || (pl.algo == 0 && wino && g_algo_mode == 2))) {
      have = false;
      keep_cache = true;
    }
    if (!have && !keep_cache && g_autotune && tune_plan(p, M, k, scale, shift, residual, y, relu, ws, ws_bytes, stream, allow_split, wino, &pl)) {
      std::lock_guard<std::mutex> lock(g_plan_mutex);
      g_plan_cache[key] = pl;
      have = true;
    }
  }
  if (!have) {
    pl = choose_plan(p.M, k, p.ksteps, allow_split ? split_k : 1);
    if (g_algo_mode == 2 && split_k <= 0 && g_force_tm == 0 && residual == nullptr &&
        winograd_ok(r, s, stride, pad, c, k, out_stride)) {
      pl = Plan{M >= 2048 ? 2 : 5, 1, (c + BK - 1) / BK};   // forced Winograd without tuning: a mid-size GEMM tile
      pl.algo = 1;
      if (g_wino_fuse == 2 && (c % BK) == 0) { pl.cfg = 5; pl.fuse_in = 1; }
    }
  }
  if (!have && split_k <= 0 && pl.splits > 1 && (!ws || ws_bytes < plan_ws_bytes(pl, p, M, k))) {
    // the workspace was sized for this shape's cached plan, which does not apply to THIS call (a Winograd plan and a call
    // with a residual): run unsplit rather than fail
    pl = choose_plan(p.M, k, p.ksteps, 1);
  }
  {
    const size_t need = plan_ws_bytes(pl, p, M, k);
    if (need > 0 && (!ws || ws_bytes < need))
      return frcnn::fail(FRCNN_ERR_WS, "conv2d: workspace %zu < %zu bytes", ws_bytes, need);
  }
  return launch_plan(p, pl, M, k, scale, shift, residual, y, relu, ws, stream);
}
}  // namespace

extern "C" int frcnn_conv2d_fwd(const float* x, const float* wgt, const float* scale, const float* shift,
                                const float* residual, float* y, int n, int h, int w, int c, int k, int r, int s,
                                int stride, int pad, int relu, int split_k, void* ws, size_t ws_bytes,
                                void* stream_) {
  FRCNN_REQUIRE(x && wgt && y, "conv2d_fwd: null tensor");
  FRCNN_REQUIRE(conv_args_ok(n, h, w, c, k, r, s, stride, pad),
                "conv2d_fwd: bad shape n=%d h=%d w=%d c=%d k=%d r=%d s=%d stride=%d pad=%d (need c%%4==0)", n, h, w, c,
                k, r, s, stride, pad);
  if (g_prof_on) ++g_prof_call;
  return run_conv(x, wgt, scale, shift, residual, y, n, h, w, c, k, r, s, stride, pad, relu, split_k, ws, ws_bytes,
                  static_cast<hipStream_t>(stream_), 1, 0, 0);
}

extern "C" size_t frcnn_conv2d_winograd_filter_bytes(int k, int c) {
  if (k <= 0 || c <= 0 || (k % 4) || (c % 4)) return 0;
  return (size_t)16 * k * c * sizeof(float);
}

extern "C" int frcnn_conv2d_winograd_filter(const float* w_krsc, float* u, int k, int c, void* stream_) {
  FRCNN_REQUIRE(w_krsc && u && k > 0 && c > 0 && (k % 4) == 0 && (c % 4) == 0,
                "conv2d_winograd_filter: need a (k,3,3,c) filter with k%%4 == 0 and c%%4 == 0 (k=%d c=%d)", k, c);
  const size_t threads = (size_t)k * (c / 4);
  hipLaunchKernelGGL(wino_filter_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), w_krsc, u, k, c / 4);
  return frcnn::check_launch("wino_filter_kernel");
}

extern "C" int frcnn_conv2d_fwd_pre(const float* x, const float* wgt, const float* w_winograd, const float* scale,
                                    const float* shift, const float* residual, float* y, int n, int h, int w, int c,
                                    int k, int r, int s, int stride, int pad, int relu, int split_k, void* ws,
                                    size_t ws_bytes, void* stream_) {
  FRCNN_REQUIRE(x && wgt && y, "conv2d_fwd_pre: null tensor");
  FRCNN_REQUIRE(conv_args_ok(n, h, w, c, k, r, s, stride, pad),
                "conv2d_fwd_pre: bad shape n=%d h=%d w=%d c=%d k=%d r=%d s=%d stride=%d pad=%d (need c%%4==0)", n, h, w, c,
                k, r, s, stride, pad);
  FRCNN_REQUIRE(!w_winograd || winograd_ok(r, s, stride, pad, c, k, 1),
                "conv2d_fwd_pre: a Winograd filter only goes with a 3x3 / stride 1 / pad 1 layer, c%%4 == 0, k%%4 == 0");
  if (g_prof_on) ++g_prof_call;
  return run_conv(x, wgt, scale, shift, residual, y, n, h, w, c, k, r, s, stride, pad, relu, split_k, ws, ws_bytes,
                  static_cast<hipStream_t>(stream_), 1, 0, 0, w_winograd);
}

extern "C" int frcnn_conv2d_profile_begin(void) {
  for (ProfRec& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_prof.clear();
  g_prof_call = -1;
  g_prof_on = true;
  return FRCNN_OK;
}

extern "C" int frcnn_conv2d_profile_end(float* us, int* call, int* kind, int capacity) {
  g_prof_on = false;
  int n = 0;
  for (ProfRec& r : g_prof) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess && n < capacity && us) {
      us[n] = ms * 1e3f;
      call[n] = r.call;
      kind[n] = r.kind;
    }
    ++n;
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_prof.clear();
  return n;      // dispatches recorded (call again with a larger buffer if it exceeds the capacity: the data is gone)
}

// ------------------------------------------------------------------------------------------------
// Data gradient: dx = conv_transpose(dy, w).  For stride 1 this is a forward convolution of dy with the
// filter flipped in (r, s) and transposed in (k, c); a strided 1x1 scatters a 1x1 convolution onto the
// even pixels; a strided RxS first zero-inserts dy.  Replaces autograd's conv backward for the
// trainable part of lib/nets/resnet.py / lib/nets/fpn.py (lib/model/train_val.py:458 -> loss.backward()).
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void transpose_filter_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                              int K, int R, int S, int C) {
  const size_t total = (size_t)K * R * S * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // i enumerates the OUTPUT [c][r'][s'][k] so that writes are coalesced
    const int k = (int)(i % K);
    size_t t = i / K;
    const int s2 = (int)(t % S);
    t /= S;
    const int r2 = (int)(t % R);
    const int c = (int)(t / R);
    wt[i] = w[(((size_t)k * R + (R - 1 - r2)) * S + (S - 1 - s2)) * C + c];
  }
}

// dyd[n, ho*stride, wo*stride, :] = dy[n, ho, wo, :], zeros elsewhere (Hd x Wd map), 16 B per thread
__global__ __launch_bounds__(256) void dilate_kernel(const float* __restrict__ dy, float* __restrict__ dyd, int N,
                                                    int Ho, int Wo, int K4, int stride, int Hd, int Wd) {
  const size_t total = (size_t)N * Hd * Wd * K4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k4 = (int)(i % K4);
    size_t t = i / K4;
    const int wd = (int)(t % Wd);
    t /= Wd;
    const int hd = (int)(t % Hd);
    const int n = (int)(t / Hd);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (hd % stride == 0 && wd % stride == 0 && hd / stride < Ho && wd / stride < Wo)
      v = reinterpret_cast<const f32x4*>(dy)[(((size_t)n * Ho + hd / stride) * Wo + wd / stride) * K4 + k4];
    reinterpret_cast<f32x4*>(dyd)[i] = v;
  }
}

struct DgradGeom {
  int ho, wo, hd, wd, pad_t;
  bool dilate;
};
DgradGeom dgrad_geom(int h, int w, int r, int s, int stride, int pad) {
  DgradGeom g;
  g.ho = (h + 2 * pad - r) / stride + 1;
  g.wo = (w + 2 * pad - s) / stride + 1;
  g.pad_t = r - 1 - pad;
  g.dilate = stride > 1 && (r > 1 || s > 1);
  // zero-inserted map, extended by the rows/cols the strided forward pass never reached
  g.hd = (g.ho - 1) * stride + 1 + (h + 2 * pad - r) % stride;
  g.wd = (g.wo - 1) * stride + 1 + (w + 2 * pad - s) % stride;
  return g;
}
bool dgrad_args_ok(int n, int h, int w, int c, int k, int r, int s, int stride, int pad) {
  return conv_args_ok(n, h, w, c, k, r, s, stride, pad) && (k % 4) == 0 && r == s && r - 1 - pad >= 0;
}
}  // namespace

extern "C" int frcnn_conv2d_transpose_filter(const float* w_krsc, float* w_crsk_flipped, int k, int r, int s, int c,
                                             void* stream_) {
  FRCNN_REQUIRE(w_krsc && w_crsk_flipped && k > 0 && r > 0 && s > 0 && c > 0, "conv2d_transpose_filter: bad arguments");
  const size_t total = (size_t)k * r * s * c;
  hipLaunchKernelGGL(transpose_filter_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), w_krsc, w_crsk_flipped, k, r, s, c);
  return frcnn::check_launch("transpose_filter_kernel");
}

extern "C" size_t frcnn_conv2d_bwd_data_ws_bytes(int n, int h, int w, int c, int k, int r, int s, int stride,
                                                 int pad) {
  if (!dgrad_args_ok(n, h, w, c, k, r, s, stride, pad)) return 0;
  const DgradGeom g = dgrad_geom(h, w, r, s, stride, pad);
  if (stride > 1 && !g.dilate) return 0;  // strided 1x1: scattered output, no split-K
  if (!g.dilate) return frcnn_conv2d_fwd_ws_bytes(n, g.ho, g.wo, k, c, r, s, 1, g.pad_t, 0);
  const size_t dil = frcnn::align_up((size_t)n * g.hd * g.wd * k * sizeof(float), 256);
  return dil + frcnn_conv2d_fwd_ws_bytes(n, g.hd, g.wd, k, c, r, s, 1, g.pad_t, 0);
}

extern "C" int frcnn_conv2d_bwd_data_pre(const float* dy, const float* w_crsk_flipped, const float* w_winograd,
                                         const float* add, float* dx, int n, int h, int w, int c, int k, int r, int s,
                                         int stride, int pad, void* ws, size_t ws_bytes, void* stream_);

extern "C" int frcnn_conv2d_bwd_data(const float* dy, const float* w_crsk_flipped, const float* add, float* dx, int n,
                                     int h, int w, int c, int k, int r, int s, int stride, int pad, void* ws,
                                     size_t ws_bytes, void* stream_) {
  return frcnn_conv2d_bwd_data_pre(dy, w_crsk_flipped, nullptr, add, dx, n, h, w, c, k, r, s, stride, pad, ws, ws_bytes,
                                   stream_);
}

extern "C" int frcnn_conv2d_bwd_data_act(const float* dy, const float* w_crsk_flipped, const float* w_winograd,
                                         const float* add, const float* act_y, const float* act_scale, float* dx, int n,
                                         int h, int w, int c, int k, int r, int s, int stride, int pad, void* ws,
                                         size_t ws_bytes, void* stream_);

extern "C" int frcnn_conv2d_bwd_data_pre(const float* dy, const float* w_crsk_flipped, const float* w_winograd,
                                         const float* add, float* dx, int n, int h, int w, int c, int k, int r, int s,
                                         int stride, int pad, void* ws, size_t ws_bytes, void* stream_) {
  return frcnn_conv2d_bwd_data_act(dy, w_crsk_flipped, w_winograd, add, nullptr, nullptr, dx, n, h, w, c, k, r, s, stride,
                                   pad, ws, ws_bytes, stream_);
}

extern "C" int frcnn_conv2d_bwd_data_act(const float* dy, const float* w_crsk_flipped, const float* w_winograd,
                                         const float* add, const float* act_y, const float* act_scale, float* dx, int n,
                                         int h, int w, int c, int k, int r, int s, int stride, int pad, void* ws,
                                         size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FRCNN_REQUIRE(!act_scale || act_y, "conv2d_bwd_data_act: act_scale without act_y");
  FRCNN_REQUIRE(!act_y || stride == 1 || (r > 1 || s > 1),
                "conv2d_bwd_data_act: the strided 1x1 data gradient (scattered output) has no activation epilogue");
  FRCNN_REQUIRE(!w_winograd || (stride == 1 && !add && winograd_ok(r, s, 1, r - 1 - pad, k, c, 1)),
                "conv2d_bwd_data_pre: a Winograd filter only goes with a 3x3 / stride 1 / pad 1 layer without `add`");
  FRCNN_REQUIRE(dy && w_crsk_flipped && dx, "conv2d_bwd_data: null tensor");
  FRCNN_REQUIRE(dgrad_args_ok(n, h, w, c, k, r, s, stride, pad),
                "conv2d_bwd_data: bad shape n=%d h=%d w=%d c=%d k=%d r=%d s=%d stride=%d pad=%d (need c%%4==0, k%%4==0, "
                "r==s, pad<=r-1)", n, h, w, c, k, r, s, stride, pad);
  const DgradGeom g = dgrad_geom(h, w, r, s, stride, pad);
  const size_t need = frcnn_conv2d_bwd_data_ws_bytes(n, h, w, c, k, r, s, stride, pad);
  if (need > 0 && (!ws || ws_bytes < need))
    return frcnn::fail(FRCNN_ERR_WS, "conv2d_bwd_data: workspace %zu < %zu bytes", ws_bytes, need);
  if (stride == 1)  // dx (n,h,w,c) = conv(dy (n,ho,wo,k), w^T flipped), same-size output
    return run_conv(dy, w_crsk_flipped, nullptr, nullptr, add, dx, n, g.ho, g.wo, k, c, r, s, 1, g.pad_t, 0, 0, ws,
                    ws_bytes, stream, 1, 0, 0, w_winograd, act_y, act_scale);
  if (!g.dilate) {
    // strided 1x1: only pixels (ho*stride, wo*stride) receive a gradient; the rest is `add` (or zero)
    const size_t bytes = (size_t)n * h * w * c * sizeof(float);
    hipError_t e = add ? frcnn::copy_bytes(dx, add, bytes, stream) : frcnn::fill_bytes(dx, 0, bytes, stream);
    if (e != hipSuccess) return frcnn::fail(FRCNN_ERR_LAUNCH, "conv2d_bwd_data: init dx: %s", hipGetErrorString(e));
    return run_conv(dy, w_crsk_flipped, nullptr, nullptr, add, dx, n, g.ho, g.wo, k, c, 1, 1, 1, 0, 0, 1, nullptr, 0,
                    stream, stride, h, w);
  }
  float* dyd = static_cast<float*>(ws);
  const size_t dil = frcnn::align_up((size_t)n * g.hd * g.wd * k * sizeof(float), 256);
  const size_t total = (size_t)n * g.hd * g.wd * (k / 4);
  hipLaunchKernelGGL(dilate_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 8192)), dim3(256), 0, stream, dy,
                     dyd, n, g.ho, g.wo, k / 4, stride, g.hd, g.wd);
  int rc = frcnn::check_launch("dilate_kernel");
  if (rc != FRCNN_OK) return rc;
  return run_conv(dyd, w_crsk_flipped, nullptr, nullptr, add, dx, n, g.hd, g.wd, k, c, r, s, 1, g.pad_t, 0, 0,
                  static_cast<char*>(ws) + dil, ws_bytes - dil, stream, 1, 0, 0, nullptr, act_y, act_scale);
}

// ------------------------------------------------------------------------------------------------
// MaxPool 3x3 / stride 2 / pad 1, NHWC (lib/nets/resnet.py:156).  One thread = 4 channels of one
// output pixel; consecutive threads walk the channel dimension -> 16-byte coalesced accesses.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc(const float* __restrict__ x, float* __restrict__ y, int N,
                                                        int H, int W, int C4, int Ho, int Wo) {
  const size_t total = (size_t)N * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t pix = i / C4;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int hi = ho * 2 - 1 + dy;
      if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int wi = wo * 2 - 1 + dx;
        if ((unsigned)wi >= (unsigned)W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + hi) * W + wi) * C4 * 4 + c4 * 4);
        m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
      }
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = m;
  }
}

__global__ __launch_bounds__(256) void pad_channels_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          size_t pixels, int c, int c_pad) {
  const size_t total = pixels * c_pad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / c_pad;
    const int ch = (int)(i - pix * c_pad);
    y[i] = ch < c ? x[pix * c + ch] : 0.f;
  }
}
}  // namespace

extern "C" int frcnn_maxpool3x3s2_fwd(const float* x, float* y, int n, int h, int w, int c, void* stream_) {
  FRCNN_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "maxpool3x3s2_fwd: bad arguments (c%%4==0)");
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const size_t total = (size_t)n * ho * wo * (c / 4);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048 * 4);
  hipLaunchKernelGGL(maxpool3x3s2_nhwc, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), x, y, n, h, w,
                     c / 4, ho, wo);
  return frcnn::check_launch("maxpool3x3s2_nhwc");
}

// Backward of the 3x3/2 max-pool (autograd of nn.MaxPool2d in the trainable stem, cfg.RESNET.FIXED_BLOCKS == -1):
// gather form, deterministic.  An input pixel lies in at most 2 x 2 windows; it receives dy of a window when it is that
// window's FIRST maximum in row-major scan order (the index torch's forward stores).
namespace {
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_nhwc(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dx, int N, int H, int W, int C4, int Ho,
                                                            int Wo) {
  const size_t total = (size_t)N * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    size_t pix = i / C4;
    const int w = (int)(pix % W);
    pix /= W;
    const int h = (int)(pix % H);
    const int n = (int)(pix / H);
    const f32x4 me = *reinterpret_cast<const f32x4*>(x + i * 4);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    for (int ho = max(0, (h - 1 + 1) / 2); ho <= min(Ho - 1, (h + 1) / 2); ++ho)
      for (int wo = max(0, (w - 1 + 1) / 2); wo <= min(Wo - 1, (w + 1) / 2); ++wo) {
        // is (h, w) the first maximum of window (ho, wo)?  earlier = strictly before in row-major order
        bool first[4] = {true, true, true, true};
        for (int dyy = 0; dyy < 3; ++dyy) {
          const int hi = ho * 2 - 1 + dyy;
          if ((unsigned)hi >= (unsigned)H) continue;
          for (int dxx = 0; dxx < 3; ++dxx) {
            const int wi = wo * 2 - 1 + dxx;
            if ((unsigned)wi >= (unsigned)W || (hi == h && wi == w)) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + hi) * W + wi) * C4 * 4 + c4 * 4);
            const bool earlier = hi < h || (hi == h && wi < w);
            for (int e = 0; e < 4; ++e) first[e] = first[e] && (earlier ? v[e] < me[e] : v[e] <= me[e]);
          }
        }
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + ((((size_t)n * Ho + ho) * Wo + wo) * C4 + c4) * 4);
        for (int e = 0; e < 4; ++e) g[e] += first[e] ? d[e] : 0.f;
      }
    *reinterpret_cast<f32x4*>(dx + i * 4) = g;
  }
}
}  // namespace

extern "C" int frcnn_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int n, int h, int w, int c,
                                      void* stream_) {
  FRCNN_REQUIRE(x && dy && dx && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "maxpool3x3s2_bwd: bad arguments (c%%4==0)");
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const size_t total = (size_t)n * h * w * (c / 4);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048 * 4);
  hipLaunchKernelGGL(maxpool3x3s2_bwd_nhwc, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), x, dy, dx, n, h,
                     w, c / 4, ho, wo);
  return frcnn::check_launch("maxpool3x3s2_bwd_nhwc");
}

extern "C" int frcnn_pad_channels(const float* x, float* y, int64_t pixels, int c, int c_pad, void* stream_) {
  FRCNN_REQUIRE(x && y && pixels > 0 && c > 0 && c_pad >= c, "pad_channels: bad arguments");
  const size_t total = (size_t)pixels * c_pad;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 2048 * 4);
  hipLaunchKernelGGL(pad_channels_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), x, y,
                     (size_t)pixels, c, c_pad);
  return frcnn::check_launch("pad_channels_kernel");
}
