// Stand-alone reproducer for the single-chain hipGraph replay fault described in DESIGN.md section 4.8
// (model/train_graph.py: a training step captured as ONE chain "adds wrong filter gradients from its second replay on"
// under ROCm 7's default replay of single-chain graphs, "graph packet capture"; correct with
// DEBUG_CLR_GRAPH_PACKET_CAPTURE=0).
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/graph_replay_repro tools/graph_replay_repro.hip
//   tools/bin/graph_replay_repro                       # default runtime path
//   DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 tools/bin/graph_replay_repro
//
// Every variant captures one stream into a graph (hipStreamCaptureModeGlobal, a linear chain), replays it REPLAYS times
// and compares the accumulator after EACH replay with the same sequence of eager launches.  The chain always ends with the
// in-place accumulation pattern of frcnn_conv2d_bwd_weight_acc (grad += f(scratch)), so a node that ran out of order, did
// not run, or read stale kernel arguments shows as a diverging accumulator.  Variants differ in ONE node kind:
//   kernels      only kernel nodes
//   memset_big   + hipMemsetAsync of the scratch buffer (9.6 MB, frcnn_conv2d_bwd_data's strided 1x1 form)
//   memset_odd   + hipMemsetAsync of 8 bytes and of an odd byte count (frcnn_anchor_target_layer's counters / flags)
//   memcpy_d2d   + hipMemcpyAsync device->device (frcnn_conv2d_bwd_data with `add`, counts read-back)
//   kernarg_tbl  the accumulate kernel takes a by-value table of 3 x 24 pointers (frcnn_conv2d_bwd_weight_acc_grouped)
//   ext_launch   kernels launched with hipExtLaunchKernelGGL (no events)
//   all          every node kind in one chain
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                    \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) {                                                                         \
      std::fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
      std::exit(2);                                                                                 \
    }                                                                                               \
  } while (0)

constexpr int GROUPS = 24;
struct Table {
  const float* x[GROUPS];
  const float* dy[GROUPS];
  float* grad[GROUPS];
};

// scratch[i * stride] += src[i] * scale   (the strided scatter of the 1x1 / stride 2 data gradient)
__global__ void scatter_kernel(const float* __restrict__ src, float* __restrict__ scratch, size_t n, int stride, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    scratch[i * stride] += src[i] * scale;
}
// counters[0] += number of positive scratch elements (atomics into a buffer that a memset zeroes first)
__global__ void count_kernel(const float* __restrict__ scratch, size_t n, int* __restrict__ counters, unsigned char* flags,
                             size_t nflags) {
  int local = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    local += scratch[i] > 0.f ? 1 : 0;
  if (local) atomicAdd(&counters[0], local);
  if (blockIdx.x == 0 && threadIdx.x < nflags && flags[threadIdx.x] != 0) atomicAdd(&counters[1], 1);   // flags must be 0
  if (blockIdx.x == 0 && threadIdx.x < nflags) flags[threadIdx.x] = 1;                                  // dirty them
}
// grad[i] += scratch[i] * w + counts   (in-place accumulation across replays)
__global__ void accumulate_kernel(const float* __restrict__ scratch, const int* __restrict__ counts, float* __restrict__ grad,
                                  size_t n, float w) {
  const float c = counts ? 1e-6f * (float)(counts[0] + 1000 * counts[1]) : 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    grad[i] += scratch[i] * w + c;
}
__global__ void accumulate_table_kernel(Table t, size_t n) {
  const int g = blockIdx.y;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    t.grad[g][i] += t.x[g][i] * t.dy[g][i];
}
__global__ void touch_kernel(float* __restrict__ x, size_t n, float a) {   // dirties the scratch buffer after use
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = a;
}

struct Bufs {
  float *src, *scratch, *grad, *tsrc, *tgrad;
  int *counters, *counts;
  unsigned char* flags;
  size_t n_src, n_scratch, n_flags, n_t;
  int stride;
};

enum { F_MEMSET_BIG = 1, F_MEMSET_ODD = 2, F_MEMCPY = 4, F_TABLE = 8, F_EXT = 16 };

static void chain(const Bufs& b, int flags, hipStream_t s, int step) {
  const dim3 grid(512), block(256);
  const float scale = 1.0f + 0.25f * 0.f * step;     // identical work per replay: a graph keeps its scalars
  if (flags & F_MEMSET_BIG) CHECK(hipMemsetAsync(b.scratch, 0, b.n_scratch * sizeof(float), s));
  else hipLaunchKernelGGL(touch_kernel, grid, block, 0, s, b.scratch, b.n_scratch, 0.f);
  if (flags & F_MEMSET_ODD) {
    CHECK(hipMemsetAsync(b.counters, 0, 2 * sizeof(int), s));
    CHECK(hipMemsetAsync(b.flags, 0, b.n_flags, s));                       // odd byte count
  }
  if (flags & F_EXT) hipExtLaunchKernelGGL(scatter_kernel, grid, block, 0, s, nullptr, nullptr, 0, b.src, b.scratch, b.n_src, b.stride, scale);
  else hipLaunchKernelGGL(scatter_kernel, grid, block, 0, s, b.src, b.scratch, b.n_src, b.stride, scale);
  const int* counts = nullptr;
  if (flags & F_MEMSET_ODD) {
    hipLaunchKernelGGL(count_kernel, grid, block, 0, s, b.scratch, b.n_scratch, b.counters, b.flags, b.n_flags);
    counts = b.counters;
    if (flags & F_MEMCPY) {
      CHECK(hipMemcpyAsync(b.counts, b.counters, 2 * sizeof(int), hipMemcpyDeviceToDevice, s));
      counts = b.counts;
    }
  } else if (flags & F_MEMCPY) {
    // dx = add (device -> device copy of the whole tensor), then the scatter adds onto it once more
    CHECK(hipMemcpyAsync(b.scratch, b.tsrc, b.n_scratch * sizeof(float), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(scatter_kernel, grid, block, 0, s, b.src, b.scratch, b.n_src, b.stride, scale);
  }
  if (flags & F_EXT) hipExtLaunchKernelGGL(accumulate_kernel, grid, block, 0, s, nullptr, nullptr, 0, b.scratch, counts, b.grad, b.n_scratch, 0.5f);
  else hipLaunchKernelGGL(accumulate_kernel, grid, block, 0, s, b.scratch, counts, b.grad, b.n_scratch, 0.5f);
  if (flags & F_TABLE) {
    Table t;
    for (int g = 0; g < GROUPS; ++g) {
      t.x[g] = b.tsrc + (size_t)g * b.n_t;
      t.dy[g] = b.scratch + (size_t)g * b.n_t;
      t.grad[g] = b.tgrad + (size_t)g * b.n_t;
    }
    hipLaunchKernelGGL(accumulate_table_kernel, dim3(64, GROUPS), block, 0, s, t, b.n_t);
  }
  hipLaunchKernelGGL(touch_kernel, grid, block, 0, s, b.scratch, b.n_scratch, 3.0f);   // leave the scratch dirty
}

static double checksum(const float* dev, size_t n, hipStream_t s) {
  std::vector<float> h(n);
  CHECK(hipMemcpyAsync(h.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost, s));
  CHECK(hipStreamSynchronize(s));
  double acc = 0;
  for (size_t i = 0; i < n; ++i) acc += (double)h[i] * (double)((i % 251) + 1);
  return acc;
}

int main(int argc, char** argv) {
  const int REPLAYS = 6;
  const char* env = std::getenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE");
  std::printf("DEBUG_CLR_GRAPH_PACKET_CAPTURE=%s\n", env ? env : "(unset: runtime default)");
  Bufs b;
  b.stride = 2;
  b.n_src = (size_t)75 * 125 * 128;            // dy of a strided 1x1 layer
  b.n_scratch = b.n_src * b.stride;            // 9.6 MB
  b.n_flags = 1021;                            // odd byte count
  b.n_t = b.n_scratch / GROUPS;
  CHECK(hipMalloc(&b.src, b.n_src * sizeof(float)));
  CHECK(hipMalloc(&b.scratch, b.n_scratch * sizeof(float)));
  CHECK(hipMalloc(&b.grad, b.n_scratch * sizeof(float)));
  CHECK(hipMalloc(&b.tsrc, b.n_scratch * sizeof(float)));
  CHECK(hipMalloc(&b.tgrad, b.n_scratch * sizeof(float)));
  CHECK(hipMalloc(&b.counters, 64));
  CHECK(hipMalloc(&b.counts, 64));
  CHECK(hipMalloc(&b.flags, 2048));
  std::vector<float> h(b.n_scratch);
  for (size_t i = 0; i < b.n_scratch; ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.3f;
  CHECK(hipMemcpy(b.src, h.data(), b.n_src * sizeof(float), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(b.tsrc, h.data(), b.n_scratch * sizeof(float), hipMemcpyHostToDevice));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

  struct Variant { const char* name; int flags; };
  const Variant variants[] = {{"kernels", 0}, {"memset_big", F_MEMSET_BIG}, {"memset_odd", F_MEMSET_ODD},
                              {"memcpy_d2d", F_MEMCPY}, {"memset_odd+memcpy", F_MEMSET_ODD | F_MEMCPY},
                              {"kernarg_tbl", F_TABLE}, {"ext_launch", F_EXT},
                              {"all", F_MEMSET_BIG | F_MEMSET_ODD | F_MEMCPY | F_TABLE | F_EXT}};
  int failures = 0;
  for (const Variant& v : variants) {
    if (argc > 1 && std::strcmp(argv[1], v.name) != 0) continue;
    // eager reference: REPLAYS passes, checksum after each
    std::vector<double> ref(REPLAYS), got(REPLAYS), reft(REPLAYS), gott(REPLAYS);
    CHECK(hipMemsetAsync(b.grad, 0, b.n_scratch * sizeof(float), s));
    CHECK(hipMemsetAsync(b.tgrad, 0, b.n_scratch * sizeof(float), s));
    CHECK(hipMemsetAsync(b.flags, 0, 2048, s));
    for (int r = 0; r < REPLAYS; ++r) {
      chain(b, v.flags, s, r);
      ref[r] = checksum(b.grad, b.n_scratch, s);
      reft[r] = checksum(b.tgrad, b.n_scratch, s);
    }
    // captured chain
    CHECK(hipMemsetAsync(b.grad, 0, b.n_scratch * sizeof(float), s));
    CHECK(hipMemsetAsync(b.tgrad, 0, b.n_scratch * sizeof(float), s));
    CHECK(hipMemsetAsync(b.flags, 0, 2048, s));
    CHECK(hipStreamSynchronize(s));
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    chain(b, v.flags, s, 0);
    CHECK(hipStreamEndCapture(s, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    size_t nodes = 0;
    CHECK(hipGraphGetNodes(graph, nullptr, &nodes));
    int first_bad = -1;
    for (int r = 0; r < REPLAYS; ++r) {
      CHECK(hipGraphLaunch(exec, s));
      got[r] = checksum(b.grad, b.n_scratch, s);
      gott[r] = checksum(b.tgrad, b.n_scratch, s);
      const bool ok = got[r] == ref[r] && gott[r] == reft[r];
      if (!ok && first_bad < 0) first_bad = r;
    }
    // back-to-back replays without a host read in between (the training loop's pattern)
    CHECK(hipMemsetAsync(b.grad, 0, b.n_scratch * sizeof(float), s));
    CHECK(hipMemsetAsync(b.tgrad, 0, b.n_scratch * sizeof(float), s));
    CHECK(hipMemsetAsync(b.flags, 0, 2048, s));
    for (int r = 0; r < REPLAYS; ++r) CHECK(hipGraphLaunch(exec, s));
    const double b2b = checksum(b.grad, b.n_scratch, s), b2bt = checksum(b.tgrad, b.n_scratch, s);
    const bool b2b_ok = b2b == ref[REPLAYS - 1] && b2bt == reft[REPLAYS - 1];
    std::printf("%-20s nodes %2zu  replay-by-replay: %s", v.name, nodes, first_bad < 0 ? "equal to eager" : "DIVERGES");
    if (first_bad >= 0)
      std::printf(" from replay %d (got %.9g / %.9g, eager %.9g / %.9g)", first_bad + 1, got[first_bad], gott[first_bad],
                  ref[first_bad], reft[first_bad]);
    std::printf("   back-to-back x%d: %s\n", REPLAYS, b2b_ok ? "equal" : "DIVERGES");
    failures += (first_bad >= 0) + !b2b_ok;
    CHECK(hipGraphExecDestroy(exec));
    CHECK(hipGraphDestroy(graph));
  }
  std::printf("%s\n", failures ? "FAULT REPRODUCED" : "no divergence in any variant");
  return failures ? 1 : 0;
}
