#include "common.h"

#include <atomic>

namespace frcnn {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

namespace {
std::atomic<int> g_memops_mode{0};

// dst[0, bytes) = value: 16-byte stores over the aligned body, byte stores over the unaligned head and tail
__global__ __launch_bounds__(256) void fill_bytes_kernel(unsigned char* __restrict__ dst, unsigned value, size_t head,
                                                        size_t body16, size_t tail) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  const unsigned v4 = value | (value << 8) | (value << 16) | (value << 24);
  uint4* body = reinterpret_cast<uint4*>(dst + head);
  for (size_t i = tid; i < body16; i += nth) body[i] = make_uint4(v4, v4, v4, v4);
  if (tid < head) dst[tid] = (unsigned char)value;
  if (tid < tail) dst[head + body16 * 16 + tid] = (unsigned char)value;
}
// dst = src; both 16-byte aligned -> 16-byte copies with a byte tail, otherwise bytes
__global__ __launch_bounds__(256) void copy_bytes_kernel(unsigned char* __restrict__ dst, const unsigned char* __restrict__ src,
                                                        size_t body16, size_t tail_begin, size_t bytes) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  const uint4* s = reinterpret_cast<const uint4*>(src);
  uint4* d = reinterpret_cast<uint4*>(dst);
  for (size_t i = tid; i < body16; i += nth) d[i] = s[i];
  for (size_t i = tail_begin + tid; i < bytes; i += nth) dst[i] = src[i];
}
}  // namespace

hipError_t fill_bytes(void* dst, int value, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return hipSuccess;
  if (g_memops_mode.load() == 1) return hipMemsetAsync(dst, value, bytes, stream);
  unsigned char* p = static_cast<unsigned char*>(dst);
  size_t head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;
  if (head > bytes) head = bytes;
  const size_t body16 = (bytes - head) / 16, tail = bytes - head - body16 * 16;
  const size_t work = body16 > 16 ? body16 : 16;
  const unsigned grid = (unsigned)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048);
  hipLaunchKernelGGL(fill_bytes_kernel, dim3(grid), dim3(256), 0, stream, p, (unsigned)(value & 0xFF), head, body16, tail);
  return hipGetLastError();
}

hipError_t copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return hipSuccess;
  if (g_memops_mode.load() == 1) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
  const bool aligned = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
  const size_t body16 = aligned ? bytes / 16 : 0, tail_begin = body16 * 16;
  const size_t work = body16 > (bytes - tail_begin) ? body16 : (bytes - tail_begin);
  const unsigned grid = (unsigned)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048);
  hipLaunchKernelGGL(copy_bytes_kernel, dim3(grid ? grid : 1), dim3(256), 0, stream, static_cast<unsigned char*>(dst),
                     static_cast<const unsigned char*>(src), body16, tail_begin, bytes);
  return hipGetLastError();
}

}  // namespace frcnn

extern "C" unsigned frcnn_settings_signature(void) {
  unsigned long long words[5] = {(unsigned long long)frcnn::g_memops_mode.load(), frcnn::conv_settings_word(),
                                 frcnn::roi_settings_word(), frcnn::boxes_settings_word(), frcnn::wgrad_settings_word()};
  unsigned long long h = 1469598103934665603ull;                    // FNV-1a over the words' bytes
  for (unsigned long long w : words)
    for (int b = 0; b < 8; ++b) { h ^= (w >> (8 * b)) & 0xffu; h *= 1099511628211ull; }
  return (unsigned)(h ^ (h >> 32));
}

extern "C" int frcnn_set_memops_mode(int mode) {
  if (mode != 0 && mode != 1) return frcnn::fail(FRCNN_ERR_ARG, "set_memops_mode: 0 (kernels) or 1 (hipMemsetAsync / hipMemcpyAsync)");
  frcnn::g_memops_mode.store(mode);
  return FRCNN_OK;
}
extern "C" int frcnn_get_memops_mode(void) { return frcnn::g_memops_mode.load(); }

extern "C" int frcnn_version(void) { return 109; }
extern "C" const char* frcnn_last_error(void) { return frcnn::error_buffer(); }
