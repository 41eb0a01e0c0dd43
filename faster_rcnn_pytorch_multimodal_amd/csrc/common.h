// Shared host-side helpers of libfrcnn_hip.so (gfx950 only; no other backend is compiled in).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/frcnn_hip.h"

namespace frcnn {

// Per-thread last error text (returned by frcnn_last_error()).
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(FRCNN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return FRCNN_OK;
}

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Device memory initialisation / copies inside the library's launch sequences, as KERNEL launches (default) instead of
// hipMemsetAsync / hipMemcpyAsync: a stream capture then holds kernel nodes only.  ROCm 7's packet-captured replay of a
// single-chain hipGraph was measured to replay chains that mix kernel nodes with memset / memcpy nodes wrongly from the second
// replay on (tools/graph_replay_repro.hip, tools/train_graph_trace.py); with kernel nodes only the same
// step replays correctly.  frcnn_set_memops_mode(1) restores the runtime calls (A/B, reproducer).
hipError_t fill_bytes(void* dst, int value, size_t bytes, hipStream_t stream);
hipError_t copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t stream);

// frcnn_settings_signature(): a hash of the CURRENT VALUES of every process-wide switch that changes which kernels an entry
// point launches (frcnn_set_memops_mode, frcnn_conv2d_set_tile / _set_algo / _set_staging, frcnn_roi_align_set_variant,
// frcnn_filter_set_variant, frcnn_nms_set_suppress_at_equal, frcnn_conv2d_wgrad_set_variant).  A holder of captured graphs keys its captures by it: switching a
// setting away and back finds the old captures again.  Each translation unit reports its own switches.
unsigned long long conv_settings_word();
unsigned long long roi_settings_word();
unsigned long long boxes_settings_word();
unsigned long long wgrad_settings_word();

// frcnn_conv2d_set_autotune state (conv_igemm.hip), shared with the filter-gradient kernel's own plan cache
// (conv_wgrad.hip), which frcnn_conv2d_clear_plans empties as well.
bool autotune_enabled();
void clear_wgrad_plans();

}  // namespace frcnn

#define FRCNN_REQUIRE(cond, ...)                                   \
  do {                                                             \
    if (!(cond)) return ::frcnn::fail(FRCNN_ERR_ARG, __VA_ARGS__); \
  } while (0)
