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

// long dependent chain: node i computes dst = src * p.a + p.b with a 256-byte by-value parameter block
struct Params {
  float a, b;
  float pad[62];
};
__global__ void axpb_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n, Params p) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i] * p.a + p.b + p.pad[61];
}

static int g_memset_every = 0;   // long chains: every k-th node is preceded by a hipMemsetAsync of its destination
__global__ void add_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

static void long_chain(float* const* ring, int nring, size_t n, float* grad, int nodes, hipStream_t s) {
  for (int i = 0; i < nodes; ++i) {
    if (g_memset_every && i % g_memset_every == g_memset_every - 1) {
      // dst = 0 (runtime memset node), dst += src: a memset that is skipped or lands elsewhere leaves the ring's old contents
      CHECK(hipMemsetAsync(ring[(i + 1) % nring], 0, n * sizeof(float), s));
      hipLaunchKernelGGL(add_kernel, dim3(512), dim3(256), 0, s, ring[i % nring], ring[(i + 1) % nring], n);
      continue;
    }
    Params p;
    std::memset(&p, 0, sizeof(p));
    p.a = (i % 3 == 0) ? 0.5f : ((i % 3 == 1) ? 1.5f : 1.25f);
    p.b = 0.001f * (float)(i % 17);
    hipLaunchKernelGGL(axpb_kernel, dim3(512), dim3(256), 0, s, ring[i % nring], ring[(i + 1) % nring], n, p);
  }
  hipLaunchKernelGGL(accumulate_kernel, dim3(512), dim3(256), 0, s, ring[nodes % nring], (const int*)nullptr, grad, n, 0.5f);
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

// Eager traffic between instantiation and the replays, like a training loop's: thousands of kernel launches and runtime
// memsets on another stream (they recycle whatever transient launch state the runtime keeps).
static bool g_churn = false;
static void churn(const Bufs& b) {
  if (!g_churn) return;
  static hipStream_t other = nullptr;
  if (!other) CHECK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
  for (int i = 0; i < 4000; ++i) {
    hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, other, b.tsrc + b.n_scratch - 64, (size_t)64, 0.25f);
    if (i % 4 == 0) CHECK(hipMemsetAsync(b.counts + 8, 0, 16 + (i % 7), other));
  }
  CHECK(hipStreamSynchronize(other));
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
  for (int i = 1; i < argc; ++i)
    if (std::strcmp(argv[i], "--churn") == 0) {
      g_churn = true;
      for (int j = i; j + 1 < argc; ++j) argv[j] = argv[j + 1];
      --argc;
      break;
    }
  std::printf("eager traffic between replays: %s\n", g_churn ? "yes (--churn)" : "no");
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
    churn(b);   // (its writes land before the reference pass, so both passes see the same operands)
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
      churn(b);
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
  // ---- long chains of dependent kernel nodes (the captured training step has ~1000 nodes, the inference frame ~170) ----
  {
    const int nring = 5;
    const size_t n = (size_t)1 << 20;
    float* ring[nring];
    for (int i = 0; i < nring; ++i) CHECK(hipMalloc(&ring[i], n * sizeof(float)));
    hipStream_t launch_stream;
    CHECK(hipStreamCreateWithFlags(&launch_stream, hipStreamNonBlocking));
    for (int pass = 0; pass < 2; ++pass)
    for (int nodes : {64, 256, 512, 1024, 2048, 4096}) {
      if (pass == 1 && nodes > 1024) continue;
      g_memset_every = pass == 0 ? 0 : 100;
      char name[32];
      std::snprintf(name, sizeof(name), pass == 0 ? "long_chain_%d" : "long_memset_%d", nodes);
      if (argc > 1 && std::strcmp(argv[1], name) != 0 && std::strcmp(argv[1], "long") != 0) continue;
      std::vector<double> ref(REPLAYS);
      CHECK(hipMemcpyAsync(ring[0], b.tsrc, n * sizeof(float), hipMemcpyDeviceToDevice, s));
      CHECK(hipMemsetAsync(b.grad, 0, n * sizeof(float), s));
      for (int r = 0; r < REPLAYS; ++r) {
        long_chain(ring, nring, n, b.grad, nodes, s);
        ref[r] = checksum(b.grad, n, s);
      }
      CHECK(hipMemcpyAsync(ring[0], b.tsrc, n * sizeof(float), hipMemcpyDeviceToDevice, s));
      CHECK(hipMemsetAsync(b.grad, 0, n * sizeof(float), s));
      CHECK(hipStreamSynchronize(s));
      hipGraph_t graph;
      hipGraphExec_t exec;
      CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
      long_chain(ring, nring, n, b.grad, nodes, s);
      CHECK(hipStreamEndCapture(s, &graph));
      // like torch.cuda.CUDAGraph: instantiated with AutoFreeOnLaunch, replayed on ANOTHER stream than the capture's
      CHECK(hipGraphInstantiateWithFlags(&exec, graph, hipGraphInstantiateFlagAutoFreeOnLaunch));
      int first_bad = -1;
      for (int r = 0; r < REPLAYS; ++r) {
        churn(b);
        CHECK(hipGraphLaunch(exec, launch_stream));
        if (checksum(b.grad, n, launch_stream) != ref[r] && first_bad < 0) first_bad = r;
      }
      std::printf("%-20s replay-by-replay: %s", name, first_bad < 0 ? "equal to eager\n" : "DIVERGES");
      if (first_bad >= 0) std::printf(" from replay %d\n", first_bad + 1);
      failures += first_bad >= 0;
      CHECK(hipGraphExecDestroy(exec));
      CHECK(hipGraphDestroy(graph));
    }
  }
  std::printf("%s\n", failures ? "FAULT REPRODUCED" : "no divergence in any variant");
  return failures ? 1 : 0;
}
