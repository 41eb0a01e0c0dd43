// BatchNorm2d with batch statistics (module in train() mode) over an NHWC activation viewed as [M rows][C channels]:
// forward (+ residual add + ReLU, + running-statistics update) and backward (d_gamma, d_beta, d_input, d_residual).
// The reference trains the LiDAR backbone this way: lib/nets/lidarnet.py:152-175 puts the BatchNorm layers of
// layer2/layer3 in train() mode and lib/nets/lidarnet.py:110 makes their affine parameters trainable.
//
// HBM-bound column reductions + elementwise passes.  Channels are the fast axis, so a wave reads 64 consecutive
// channels of one row (256 B) per load; rows are split over BN_GROUPS workgroups per 64-channel column block and the
// per-group partial sums are added in group order (deterministic).  Sums are carried in fp64: M is 10^3..10^5 rows
// and E[x^2] - E[x]^2 in fp32 would cancel.
// Built with -ffp-contract=off (separate roundings, like the reference's elementwise torch ops).
#include "common.h"

#include <algorithm>

using namespace frcnn;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BN_GROUPS = 64;

// What the pass behind the column sums needs (bn_fwd_final_kernel / bn_bwd_final_kernel).  With `counters` (one int per
// 64-channel column block, zero on entry and on exit) the LAST of a column block's BN_GROUPS workgroups runs that pass itself
// at the end of bn_partial_kernel: one launch fewer per BatchNorm and direction (168 of the LiDAR step's 1290 launches).  The
// partial sums then travel at agent scope (sc1: the workgroups of a column sit on different XCDs, one L2 each).
struct BnFinal {
  int* counters;
  long M;
  const float* gamma;
  const float* beta;
  float eps, momentum;
  float* running_mean;
  float* running_var;
  float* save_mean;
  float* save_invstd;   // bwd: input
  float* alpha;         // fwd: alpha / shift; bwd: coef (3 C floats)
  float* shift;
  float* dgamma;
  float* dbeta;
  int accumulate;       // bwd: dgamma / dbeta += instead of =
};

__device__ __forceinline__ void bn_fwd_final(double s, double q, int c, const BnFinal& f) {
  const double mean = s / (double)f.M;
  double var = q / (double)f.M - mean * mean;
  if (var < 0.0) var = 0.0;
  const float mean_f = (float)mean, var_f = (float)var;
  const float invstd = 1.0f / sqrtf(var_f + f.eps);
  f.save_mean[c] = mean_f;
  f.save_invstd[c] = invstd;
  f.alpha[c] = invstd * (f.gamma ? f.gamma[c] : 1.f);
  f.shift[c] = f.beta ? f.beta[c] : 0.f;
  if (f.running_mean) f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * mean_f;
  if (f.running_var) {
    const float unbiased = f.M > 1 ? (float)(var * (double)f.M / (double)(f.M - 1)) : var_f;
    f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * unbiased;
  }
}

__device__ __forceinline__ void bn_bwd_final(double s, double q, int c, int C, const BnFinal& f) {
  if (f.dbeta) f.dbeta[c] = f.accumulate ? f.dbeta[c] + (float)s : (float)s;
  if (f.dgamma) f.dgamma[c] = f.accumulate ? f.dgamma[c] + (float)q : (float)q;
  float* coef = f.alpha;
  coef[c] = (f.gamma ? f.gamma[c] : 1.f) * f.save_invstd[c];
  coef[C + c] = (float)(s / (double)f.M);
  coef[2 * C + c] = (float)(q / (double)f.M);
}

// part[g][c] = (sum_m a[m][c], sum_m b[m][c]) over the rows of group g, where
//   forward : a = y,  b = y*y
//   backward: a = g,  b = g * xhat     with g = relu ? (out > 0 ? dout : 0) : dout,  xhat = (y - mean) * invstd
template <bool BWD>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ y, const float* __restrict__ dout,
                                                        const float* __restrict__ out, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, int relu, long M, int C,
                                                        double* __restrict__ part, const BnFinal fin) {
  __shared__ double sa[4][64], sb[4][64];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  double a = 0.0, b = 0.0;
  if (c < C) {
    const float mu = BWD ? mean[c] : 0.f, is = BWD ? invstd[c] : 0.f;
    for (long m = blockIdx.y * 4 + wave; m < M; m += 4 * BN_GROUPS) {
      const size_t i = (size_t)m * C + c;
      if (BWD) {
        float g = dout[i];
        if (relu && !(out[i] > 0.f)) g = 0.f;
        const float xh = (y[i] - mu) * is;
        a += (double)g;
        b += (double)g * (double)xh;
      } else {
        const float v = y[i];
        a += (double)v;
        b += (double)v * (double)v;
      }
    }
  }
  sa[wave][lane] = a;
  sb[wave][lane] = b;
  __syncthreads();
  if (wave == 0 && c < C) {
    const double ta = ((sa[0][lane] + sa[1][lane]) + sa[2][lane]) + sa[3][lane];
    const double tb = ((sb[0][lane] + sb[1][lane]) + sb[2][lane]) + sb[3][lane];
    double* dst = part + ((size_t)blockIdx.y * C + c) * 2;
    if (fin.counters) {
      __hip_atomic_store(dst, ta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + 1, tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      dst[0] = ta;
      dst[1] = tb;
    }
  }
  if (!fin.counters) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the partial sums have reached the coherence point
  __syncthreads();
  if (threadIdx.x == 0) {
    int* ctr = fin.counters + blockIdx.x;
    const int old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old == BN_GROUPS - 1;
    if (s_last) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last || wave != 0 || c >= C) return;
  // the last workgroup of this column block: group order, like the stand-alone pass
  double s = 0.0, q = 0.0;
#pragma unroll 8
  for (int g = 0; g < BN_GROUPS; ++g) {
    const double* src = part + ((size_t)g * C + c) * 2;
    s += __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    q += __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (BWD) bn_bwd_final(s, q, c, C, fin);
  else bn_fwd_final(s, q, c, fin);
}

// mean / biased variance -> saved statistics, the affine coefficients of the apply pass, running statistics
// (torch.nn.functional.batch_norm(training=True): running_var takes the UNBIASED variance).
__global__ __launch_bounds__(256) void bn_fwd_final_kernel(const double* __restrict__ part, int C, const BnFinal fin) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int g = 0; g < BN_GROUPS; ++g) {
    s += part[((size_t)g * C + c) * 2 + 0];
    q += part[((size_t)g * C + c) * 2 + 1];
  }
  bn_fwd_final(s, q, c, fin);
}

// out = act((y - mean[c]) * alpha[c] + beta[c] (+ residual)); centred first, so a degenerate channel (variance << mean^2)
// does not cancel
__global__ __launch_bounds__(256) void bn_fwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                          const float* __restrict__ alpha,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ residual, int relu, size_t n4,
                                                          int C4, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
    const f32x4 a = reinterpret_cast<const f32x4*>(alpha)[i % C4];
    const f32x4 b = reinterpret_cast<const f32x4*>(shift)[i % C4];
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[i % C4];
    f32x4 o;
    for (int e = 0; e < 4; ++e) o[e] = (v[e] - mu[e]) * a[e] + b[e];
    if (residual) {
      const f32x4 r = reinterpret_cast<const f32x4*>(residual)[i];
      for (int e = 0; e < 4; ++e) o[e] = o[e] + r[e];
    }
    if (relu)
      for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : 0.f;
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

// d_beta = sum g, d_gamma = sum g*xhat; coefficients of  dy = a * (g - b - xhat * k)
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const double* __restrict__ part, int C, const BnFinal fin) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int g = 0; g < BN_GROUPS; ++g) {
    s += part[((size_t)g * C + c) * 2 + 0];
    q += part[((size_t)g * C + c) * 2 + 1];
  }
  bn_bwd_final(s, q, c, C, fin);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                          const float* __restrict__ y, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ coef, int relu, size_t n4, int C4,
                                                          float* __restrict__ dy, float* __restrict__ dres) {
  const int C = C4 * 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    f32x4 g = reinterpret_cast<const f32x4*>(dout)[i];
    if (relu) {
      const f32x4 o = reinterpret_cast<const f32x4*>(out)[i];
      for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
    }
    if (dres) reinterpret_cast<f32x4*>(dres)[i] = g;
    const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c4];
    const f32x4 is = reinterpret_cast<const f32x4*>(invstd)[c4];
    const f32x4 a = reinterpret_cast<const f32x4*>(coef)[c4];
    const f32x4 b = reinterpret_cast<const f32x4*>(coef + C)[c4];
    const f32x4 k = reinterpret_cast<const f32x4*>(coef + 2 * C)[c4];
    f32x4 d;
    for (int e = 0; e < 4; ++e) {
      const float xh = (v[e] - mu[e]) * is[e];
      d[e] = a[e] * ((g[e] - b[e]) - xh * k[e]);
    }
    reinterpret_cast<f32x4*>(dy)[i] = d;
  }
}

unsigned grid_for(size_t items, unsigned cap = 8192) { return (unsigned)std::min<size_t>((items + 255) / 256, cap); }

size_t ws_need(int c) { return (size_t)BN_GROUPS * c * 2 * sizeof(double) + (size_t)3 * c * sizeof(float); }

}  // namespace

extern "C" size_t frcnn_bn_train_ws_bytes(int c) { return c > 0 ? ws_need(c) : 0; }

extern "C" int frcnn_bn_train_counters(int c) { return c > 0 ? (c + 63) / 64 : 0; }

extern "C" int frcnn_bn_train_fwd(const float* y, int64_t rows, int c, const float* gamma, const float* beta, float eps,
                                  float momentum, float* running_mean, float* running_var, const float* residual,
                                  int relu, float* out, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes,
                                  int* counters, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FRCNN_REQUIRE(y && out && save_mean && save_invstd && rows > 0 && c > 0 && c % 4 == 0,
                "bn_train_fwd: bad arguments (c%%4==0)");
  if (!ws || ws_bytes < ws_need(c)) return fail(FRCNN_ERR_WS, "bn_train_fwd: workspace %zu < %zu bytes", ws_bytes, ws_need(c));
  double* part = static_cast<double*>(ws);
  float* alpha = reinterpret_cast<float*>(part + (size_t)BN_GROUPS * c * 2);
  float* shift = alpha + c;
  BnFinal fin;
  fin.counters = counters; fin.M = (long)rows; fin.gamma = gamma; fin.beta = beta; fin.eps = eps; fin.momentum = momentum;
  fin.running_mean = running_mean; fin.running_var = running_var; fin.save_mean = save_mean; fin.save_invstd = save_invstd;
  fin.alpha = alpha; fin.shift = shift; fin.dgamma = nullptr; fin.dbeta = nullptr; fin.accumulate = 0;
  hipLaunchKernelGGL(bn_partial_kernel<false>, dim3((c + 63) / 64, BN_GROUPS), dim3(256), 0, stream, y, nullptr, nullptr,
                     nullptr, nullptr, 0, (long)rows, c, part, fin);
  int rc = check_launch("bn_partial_kernel<fwd>");
  if (rc != FRCNN_OK) return rc;
  if (!counters) {
    hipLaunchKernelGGL(bn_fwd_final_kernel, dim3((c + 255) / 256), dim3(256), 0, stream, part, c, fin);
    rc = check_launch("bn_fwd_final_kernel");
    if (rc != FRCNN_OK) return rc;
  }
  const size_t n4 = (size_t)rows * (c / 4);
  hipLaunchKernelGGL(bn_fwd_apply_kernel, dim3(grid_for(n4)), dim3(256), 0, stream, y, save_mean, alpha, shift, residual,
                     relu, n4, c / 4, out);
  return check_launch("bn_fwd_apply_kernel");
}

extern "C" int frcnn_bn_train_bwd(const float* dout, const float* out, const float* y, int64_t rows, int c,
                                  const float* gamma, const float* save_mean, const float* save_invstd, int relu,
                                  float* dy, float* dres, float* dgamma, float* dbeta, int accumulate, void* ws,
                                  size_t ws_bytes, int* counters, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  FRCNN_REQUIRE(dout && y && save_mean && save_invstd && dy && rows > 0 && c > 0 && c % 4 == 0 && (!relu || out),
                "bn_train_bwd: bad arguments (c%%4==0, out needed for the ReLU mask)");
  if (!ws || ws_bytes < ws_need(c)) return fail(FRCNN_ERR_WS, "bn_train_bwd: workspace %zu < %zu bytes", ws_bytes, ws_need(c));
  double* part = static_cast<double*>(ws);
  float* coef = reinterpret_cast<float*>(part + (size_t)BN_GROUPS * c * 2);
  BnFinal fin;
  fin.counters = counters; fin.M = (long)rows; fin.gamma = gamma; fin.beta = nullptr; fin.eps = 0.f; fin.momentum = 0.f;
  fin.running_mean = nullptr; fin.running_var = nullptr; fin.save_mean = nullptr;
  fin.save_invstd = const_cast<float*>(save_invstd); fin.alpha = coef; fin.shift = nullptr; fin.dgamma = dgamma;
  fin.dbeta = dbeta; fin.accumulate = accumulate ? 1 : 0;
  hipLaunchKernelGGL(bn_partial_kernel<true>, dim3((c + 63) / 64, BN_GROUPS), dim3(256), 0, stream, y, dout, out,
                     save_mean, save_invstd, relu, (long)rows, c, part, fin);
  int rc = check_launch("bn_partial_kernel<bwd>");
  if (rc != FRCNN_OK) return rc;
  if (!counters) {
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((c + 255) / 256), dim3(256), 0, stream, part, c, fin);
    rc = check_launch("bn_bwd_final_kernel");
    if (rc != FRCNN_OK) return rc;
  }
  const size_t n4 = (size_t)rows * (c / 4);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(n4)), dim3(256), 0, stream, dout, out, y, save_mean, save_invstd,
                     coef, relu, n4, c / 4, dy, dres);
  return check_launch("bn_bwd_apply_kernel");
}
