// Counter-based random numbers shared by the target layers (sub-sampling keys) and the uncertainty heads (dropout masks,
// logit distortion): a value is a pure function of (seed, stream, index), so a launch is reproducible, independent of
// the grid shape, and the CPU oracle can replay every draw (oracle/frcnn_oracle.py: rand_key / uniform01 / normal01).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace frcnn {

__host__ __device__ __forceinline__ uint32_t hash32(uint32_t x) {  // lowbias32
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t rand_key(uint32_t seed, uint32_t stream, uint32_t i) {
  return hash32(hash32(seed ^ (stream * 0x9e3779b9U)) + i * 0x85ebca6bU);
}
// uniform in (0, 1): 24 random bits, centred in their cell
__device__ __forceinline__ float uniform01(uint32_t seed, uint32_t stream, uint32_t i) {
  return ((float)(rand_key(seed, stream, i) >> 8) + 0.5f) * (1.0f / 16777216.0f);
}
// standard normal by Box-Muller from two streams
__device__ __forceinline__ float normal01(uint32_t seed, uint32_t stream, uint32_t i) {
  const float u1 = uniform01(seed, 2 * stream, i), u2 = uniform01(seed, 2 * stream + 1, i);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853071795864f * u2);
}

}  // namespace frcnn
