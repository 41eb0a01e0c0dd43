// Ceiling probe for the conv kernel's inner loop on gfx950 (tuning aid, not part of libfrcnn_hip.so):
//   V0  bare v_mfma_f32_32x32x2_f32 loop, 4 independent accumulators
//   V1  V0 + the conv kernel's LDS fragment reads (4 ds_read_b128 per 16 MFMAs, pitch-36 rows)
//   V2  V1 + one workgroup barrier per 64 MFMAs
// for 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// VARIANT 3 = VARIANT 2 with signed, full-mantissa pseudo-random operands (what a conv layer feeds the pipe):
// DVFS holds a lower clock on such data than on the smooth positive ramp of variants 0-2.
template <int VARIANT, int NT>
__global__ __launch_bounds__(NT) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 128 * 36];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 2 * 128 * 36; i += NT) {
    if (VARIANT >= 3) {
      unsigned h = (i + 1) * 2654435761u;
      h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
      lds[i] = ((float)(int)(h & 0xFFFFFF) - 8388608.f) * (1.0f / 8388608.f) * (1.f + (float)((h >> 24) & 7));
    } else {
      lds[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    }
  }
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frag = (lane & 31) * 36 + 4 * (lane >> 5);
  const float* Ab = lds + (wave & 1) * 64 * 36 + frag;
  const float* Bb = lds + 128 * 36 + ((wave >> 1) & 1) * 64 * 36 + frag;
  f32x4 a[2] = {f32x4{1.f, 2.f, 3.f, 4.f} * (float)lane, f32x4{.5f, .25f, .125f, 1.f}};
  f32x4 b[2] = {f32x4{1.f, .5f, 2.f, 1.f}, f32x4{.1f, .2f, .3f, .4f} * (float)lane};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (VARIANT >= 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * 36 + kk * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * 36 + kk * 8);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j][q], a[i][q], acc[i][j], 0, 0, 0);
    }
    if (VARIANT >= 2) __syncthreads();
    if (VARIANT >= 3 && (it & 63) == 63)   // keep the accumulators bounded without leaving the MFMA-bound regime
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
          for (int r = 0; r < 16; ++r) acc[i][j][r] *= 0.001f;
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * NT + t] = s;
}

template <int VARIANT, int NT>
void run(const char* name, int blocks_per_cu, float* out) {
  const int iters = 20000, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<VARIANT, NT>), dim3(blocks), dim3(NT), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * (NT / 64) * iters * 64.0 * (32 * 32 * 2 * 2);
  printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 512 * sizeof(float));
  run<0, 256>("V0 bare, 1 wave/SIMD", 1, out);
  run<0, 512>("V0 bare, 2 waves/SIMD (512-thread WG)", 1, out);
  run<0, 256>("V0 bare, 2 WGs of 256 per CU", 2, out);
  run<1, 256>("V1 +LDS frag reads, 1 wave/SIMD", 1, out);
  run<1, 512>("V1 +LDS frag reads, 2 waves/SIMD", 1, out);
  run<1, 256>("V1 +LDS frag reads, 2 WGs of 256 per CU", 2, out);
  run<2, 256>("V2 +barrier/64 MFMA, 1 wave/SIMD", 1, out);
  run<2, 512>("V2 +barrier/64 MFMA, 2 waves/SIMD", 1, out);
  run<2, 256>("V2 +barrier/64 MFMA, 2 WGs of 256 per CU", 2, out);
  run<3, 512>("V3 random signed operands, 2 waves/SIMD", 1, out);
  run<3, 256>("V3 random signed operands, 2 WGs of 256 per CU", 2, out);
  run<3, 512>("V3 again (sustained)", 1, out);
  hipFree(out);
  return 0;
}
