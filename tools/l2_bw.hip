// L2 -> CU read-bandwidth probe (tuning aid, not part of libfrcnn_hip.so): every wave streams 1 KB lines (16 B per lane)
// from a region small enough to stay in its XCD's 4 MB L2, G independent loads in flight per lane, the access pattern
// of the RoIAlign pooling kernel.  Reports aggregate GB/s for a few (region size, waves per CU) points.
//   hipcc --offload-arch=gfx950 -O3 tools/l2_bw.hip -o tools/bin/l2_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int G>
__global__ __launch_bounds__(256) void probe(const float4* __restrict__ buf, int lines_per_slice, int iters, float4* out) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  // workgroup b reads slice b % 8 only (blocks b and b + 8 share an XCD): each XCD's L2 holds one slice
  const float4* base = buf + (size_t)(blockIdx.x & 7) * lines_per_slice * 64;
  unsigned h = gw * 2654435761u + 12345u;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < iters; ++it) {
    float4 v[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
      h = h * 1664525u + 1013904223u;
      const unsigned line = __builtin_amdgcn_readfirstlane((h >> 8) % (unsigned)lines_per_slice);
      v[k] = base[(size_t)line * 64 + lane];
    }
#pragma unroll
    for (int k = 0; k < G; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
  }
  if (acc.x == 123.456f) out[gw] = acc;
}

int main() {
  const size_t max_bytes = (size_t)64 << 20;
  float4 *buf, *out;
  hipMalloc(&buf, max_bytes);
  hipMalloc(&out, 1 << 20);
  hipMemset(buf, 0, max_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 64;
  for (int slice_kb : {256, 1024, 2450, 8192}) {        // per-XCD working set
    for (int wg_per_cu : {2, 4, 5, 8}) {
      const int lines = slice_kb;                          // 1 KB lines
      const int nwg = 256 * wg_per_cu;
      for (int g : {4, 8, 16}) {
        auto launch = [&]() {
          if (g == 4) hipLaunchKernelGGL(probe<4>, dim3(nwg), dim3(256), 0, 0, buf, lines, iters * 2, out);
          else if (g == 8) hipLaunchKernelGGL(probe<8>, dim3(nwg), dim3(256), 0, 0, buf, lines, iters, out);
          else hipLaunchKernelGGL(probe<16>, dim3(nwg), dim3(256), 0, 0, buf, lines, iters / 2, out);
        };
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 5.0 * nwg * 4 * (double)iters * 8 * 1024.0;
        printf("slice %5d KB/XCD  %d WG/CU (%2d waves)  G=%-2d  %7.1f us/launch  %8.0f GB/s\n", slice_kb, wg_per_cu,
               wg_per_cu * 4, g, ms * 1e3 / 5, bytes / (ms * 1e-3) / 1e9);
      }
    }
  }
  return 0;
}
