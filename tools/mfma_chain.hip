// What ONE dependent accumulator chain per wave costs on gfx950 (tuning aid, not part of libfrcnn_hip.so): the 64x64 conv tile
// gives every wave a single 32x32 accumulator, so its v_mfma_f32_32x32x2_f32 instructions form one chain.
//   C1  32x32x2, 1 accumulator (chain)            C4  32x32x2, 4 independent accumulators (reference, = mfma_peak V0)
//   S4  16x16x4, 4 accumulators = the same 32x32 output per wave as C1, four independent chains of 40-cycle latency
//   L   + the conv kernel's LDS fragment reads (2 ds_read_b128 per 4 MFMAs)   B  + one workgroup barrier per 16 MFMAs
// each for 1, 2 and 4 workgroups of 256 threads per CU.   hipcc --offload-arch=gfx950 -O3 tools/mfma_chain.hip -o /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: C1, 1: C4, 2: S4.   EXTRA 0: bare, 1: + LDS reads, 2: + LDS reads + barrier per 16 (C1/S4: per K-step) MFMAs
// EXTRA 3 / 4 (C1 only): LDS reads + 4 / 8 independent v_fma_f32 behind every MFMA (does vector work hide in the MFMA's shadow?)
// EXTRA 5: LDS reads + 4 ds_write_b128 per 16 MFMAs;  EXTRA 6: LDS reads + 4 global_load_dwordx4 per 16 MFMAs (L2-resident)
// EXTRA 7: LDS reads + 16 s_add (scalar) behind every MFMA
template <int MODE, int EXTRA>
__global__ __launch_bounds__(256) void probe(float* out, int iters, const float* src) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 64 * 36];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 2 * 64 * 36; i += 256) {
    unsigned h = (i + 1) * 2654435761u;
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
    lds[i] = ((float)(int)(h & 0xFFFFFF) - 8388608.f) * (1.0f / 8388608.f);
  }
  __syncthreads();
  const int frag = (lane & 31) * 36 + 4 * (lane >> 5);
  const float* Ab = lds + (wave & 1) * 32 * 36 + frag;
  const float* Bb = lds + 64 * 36 + ((wave >> 1) & 1) * 32 * 36 + frag;
  f32x4 a = f32x4{1.f, 2.f, 3.f, 4.f} * (1e-3f * (float)lane), b = f32x4{1.f, .5f, 2.f, 1.f};
  f32x16 acc[4];
  f32x4 s4[4];
  float vx[8];
  for (int v = 0; v < 8; ++v) vx[v] = (float)(lane + v);
  unsigned sx = blockIdx.x;
  f32x4 g[4] = {};
  float* wb = lds + wave * 32 * 36 + (lane >> 3) * 36 + (lane & 7) * 4 + (EXTRA == 5 ? 0 : 0);
  for (int i = 0; i < 4; ++i) {
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    s4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (EXTRA >= 1) {
        a = *reinterpret_cast<const f32x4*>(Ab + kk * 8);
        b = *reinterpret_cast<const f32x4*>(Bb + kk * 8);
      }
      if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q], a[q], acc[0], 0, 0, 0);
          if (EXTRA == 3 || EXTRA == 4) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < (EXTRA == 3 ? 4 : 8); ++v) vx[v] = __builtin_fmaf(vx[v], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (EXTRA == 7) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < 16; ++v) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sx));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (EXTRA == 5 && kk == 1) {
#pragma unroll
          for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(wb + v * 8 * 36) = f32x4{vx[0], vx[1], vx[2], vx[3]};
        }
        if (EXTRA == 6 && kk == 2) {
#pragma unroll
          for (int v = 0; v < 4; ++v) g[v] = *reinterpret_cast<const f32x4*>(src + ((size_t)(blockIdx.x * 256 + t) * 4 + (size_t)((it * 4 + v) & 63) * 262144) % (16u << 20));
        }
      } else if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[q], a[q], acc[i], 0, 0, 0);
      } else {
        // 8 k per group: two 16x16x4 MFMAs per 16x16 sub-tile, four sub-tiles -> 8 MFMAs of 32 cycles = the C1 group's 256 cycles
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i) s4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[2 * h + (i & 1)], a[2 * h + (i >> 1)], s4[i], 0, 0, 0);
      }
    }
    if (EXTRA >= 2) __syncthreads();
    if ((it & 63) == 63) {
      for (int i = 0; i < 4; ++i) {
        for (int r = 0; r < 16; ++r) acc[i][r] *= 0.001f;
        s4[i] *= 0.001f;
      }
    }
  }
  float s = (float)sx;
  for (int v = 0; v < 8; ++v) s += vx[v];
  for (int i = 0; i < 4; ++i) {
    for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int r = 0; r < 4; ++r) s += s4[i][r] + g[i][r];
  }
  out[blockIdx.x * 256 + t] = s;
}

static float* g_src = nullptr;
template <int MODE, int EXTRA>
void run(const char* name, int blocks_per_cu, float* out) {
  const int iters = 8000, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, EXTRA>), dim3(blocks), dim3(256), 0, 0, out, iters, g_src);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // FLOP per wave per iteration: C1 16 MFMAs x 4096, C4 64 x 4096, S4 32 x 2048
  const double per_wave = MODE == 1 ? 64.0 * 4096 : 16.0 * 4096;
  const double flops = (double)blocks * 4 * iters * per_wave;
  printf("%-58s %d WG/CU %8.3f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  hipMalloc(&g_src, (size_t)(16u << 20) * sizeof(float) + 4096);
  hipMemset(g_src, 0, (size_t)(16u << 20) * sizeof(float) + 4096);
  for (int w = 1; w <= 4; w *= 2) {
    run<0, 3>("C1 + LDS reads + 4 v_fma behind every MFMA", w, out);
    run<0, 4>("C1 + LDS reads + 8 v_fma behind every MFMA", w, out);
    run<0, 7>("C1 + LDS reads + 16 s_add behind every MFMA", w, out);
    run<0, 5>("C1 + LDS reads + 4 ds_write_b128 per 16 MFMAs", w, out);
    run<0, 6>("C1 + LDS reads + 4 global_load_dwordx4 per 16 MFMAs", w, out);
    run<1, 0>("C4 32x32x2, 4 independent accumulators, bare", w, out);
    run<0, 0>("C1 32x32x2, ONE accumulator chain, bare", w, out);
    run<0, 1>("C1 + LDS fragment reads", w, out);
    run<0, 2>("C1 + LDS fragment reads + barrier per 16 MFMAs", w, out);
    run<2, 0>("S4 16x16x4, four 16x16 accumulators, bare", w, out);
    run<2, 1>("S4 + LDS fragment reads", w, out);
    run<2, 2>("S4 + LDS fragment reads + barrier per K-step", w, out);
  }
  hipFree(out);
  return 0;
}
