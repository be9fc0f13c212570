// launch_floor_probe.hip — what does ONE dependent kernel launch cost on MI355X, by launch shape?
//
// The engine's update is ~410 dependent launches; every one of them, a one-workgroup kernel included, shows 4.4 - 5.2 us in
// rocprofv3 (profiles/r02_l_kernel_stats.csv) against 2.46 us for an empty kernel launched eagerly (profiles/r01_e_gridsync_probe.txt)
// and 1.45 us for a trivial 256-workgroup kernel in MI355X_MICROARCH.md ("boundary").  This probe replays chains of NODES dependent
// launches of one kernel from a hipGraph (and eagerly) and reports microseconds per launch for each launch shape:
// kernel-argument bytes, dynamic LDS, threads per workgroup, grid size, scratch use, and what the kernel leaves in the caches
// (plain / write-through stores, reads).  For selected shapes the kernels also stamp s_memrealtime (100 MHz) at their first and
// last instruction, which separates the kernel's own duration from the gap to the next launch.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/launch_floor_probe.bin tools/launch_floor_probe.hip && tools/launch_floor_probe.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NODES = 384;

enum Body { EMPTY = 0, STORE_PLAIN = 1, STORE_WT = 2, READ = 3, SCRATCH = 4, STORE_NT = 5 };

template <int KARG>
struct Args {
  unsigned long long* stamps;  // [2 * NODES]: min start, max end (nullptr: no stamps)
  float* buf;
  int node, body, per_thread;  // per_thread: float4 accesses per thread
  int pad[(KARG - 32) / 4];
};

typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef int i32x4n __attribute__((ext_vector_type(4)));
__device__ void raw_store_f32x4(f32x4n data, i32x4n rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");

__device__ __forceinline__ void wt_store4(float* base, size_t idx, float4 v) {
  union { i32x4n v; struct { const float* p; unsigned n; unsigned f; } s; } u;
  u.s.p = base; u.s.n = 0x7FFFFFFFu; u.s.f = 0x00020000;
  raw_store_f32x4(f32x4n{v.x, v.y, v.z, v.w}, u.v, (int)(idx * 4), 0, 17);  // sc0 sc1
}

template <int KARG>
__global__ void __launch_bounds__(512) probe_kernel(Args<KARG> a) {
  extern __shared__ float dyn[];
  unsigned long long t0 = 0;
  if (a.stamps && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
  if (a.body == STORE_PLAIN) {
    for (int k = 0; k < a.per_thread; ++k) reinterpret_cast<float4*>(a.buf)[tid + k * nthr] = make_float4(1.f, 2.f, 3.f, (float)a.node);
  } else if (a.body == STORE_WT) {
    for (int k = 0; k < a.per_thread; ++k) wt_store4(a.buf, 4 * (tid + k * nthr), make_float4(1.f, 2.f, 3.f, (float)a.node));
  } else if (a.body == STORE_NT) {
    for (int k = 0; k < a.per_thread; ++k) {
      typedef float f32x4_st __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(f32x4_st{1.f, 2.f, 3.f, (float)a.node}, reinterpret_cast<f32x4_st*>(a.buf) + tid + k * nthr);
    }
  } else if (a.body == READ) {
    float s = 0.f;
    for (int k = 0; k < a.per_thread; ++k) { const float4 q = reinterpret_cast<const float4*>(a.buf)[tid + k * nthr]; s += q.x + q.y + q.z + q.w; }
    if (s == 123.456f) a.buf[0] = s;  // never true: keeps the loads alive
  } else if (a.body == SCRATCH) {
    float priv[64];
    for (int k = 0; k < 64; ++k) priv[k] = (float)(k + a.node);
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += priv[(a.node * 7 + k * 13 + threadIdx.x) & 63];  // dynamic index -> scratch
    if (s == 123.456f) a.buf[0] = s;
  }
  if (a.stamps && threadIdx.x == 0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    atomicMin(&a.stamps[2 * a.node], t0);
    atomicMax(&a.stamps[2 * a.node + 1], t1);
  }
  if (a.pad[0] == 0x7eadbeef) dyn[threadIdx.x] = 1.f;  // never true
}

struct Shape { const char* name; int karg, lds, threads, grid, body, per_thread; bool stamps; };

static hipStream_t g_stream;
static float* g_buf;
static unsigned long long* g_stamps;

template <int KARG>
static void enqueue(const Shape& s, int node) {
  Args<KARG> a;
  memset(&a, 0, sizeof(a));
  a.stamps = s.stamps ? g_stamps : nullptr; a.buf = g_buf; a.node = node; a.body = s.body; a.per_thread = s.per_thread;
  hipLaunchKernelGGL(probe_kernel<KARG>, dim3(s.grid), dim3(s.threads), s.lds, g_stream, a);
}

static void enqueue_any(const Shape& s, int node) {
  switch (s.karg) {
    case 64: enqueue<64>(s, node); break;
    case 512: enqueue<512>(s, node); break;
    case 2048: enqueue<2048>(s, node); break;
    case 4096: enqueue<4096>(s, node); break;
    default: printf("bad karg\n"); exit(1);
  }
}

static void set_lds_attr(int lds) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel<2048>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel<4096>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
}

// a chain = a repeating pattern of shapes, NODES launches in all
static double time_chain(const std::vector<Shape>& pattern, bool graph, int replays, double* body_us, double* gap_us) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  bool stamps = false;
  for (const Shape& s : pattern) stamps = stamps || s.stamps;
  auto reset_stamps = [&] {
    if (!stamps) return;
    std::vector<unsigned long long> init(2 * NODES);
    for (int i = 0; i < NODES; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0ull; }
    CK(hipMemcpy(g_stamps, init.data(), init.size() * 8, hipMemcpyHostToDevice));
  };
  reset_stamps();
  auto issue = [&] { for (int i = 0; i < NODES; ++i) enqueue_any(pattern[i % pattern.size()], i); };
  float ms = 0.f;
  if (graph) {
    hipGraph_t g; hipGraphExec_t x;
    CK(hipStreamBeginCapture(g_stream, hipStreamCaptureModeThreadLocal));
    issue();
    CK(hipStreamEndCapture(g_stream, &g));
    CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(x, g_stream));
    CK(hipStreamSynchronize(g_stream));
    reset_stamps();
    CK(hipEventRecord(e0, g_stream));
    for (int r = 0; r < replays; ++r) CK(hipGraphLaunch(x, g_stream));
    CK(hipEventRecord(e1, g_stream));
    CK(hipStreamSynchronize(g_stream));
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(x)); CK(hipGraphDestroy(g));
  } else {
    for (int w = 0; w < 3; ++w) issue();
    CK(hipStreamSynchronize(g_stream));
    reset_stamps();
    CK(hipEventRecord(e0, g_stream));
    for (int r = 0; r < replays; ++r) issue();
    CK(hipEventRecord(e1, g_stream));
    CK(hipStreamSynchronize(g_stream));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  CK(hipGetLastError());
  if (stamps && body_us && gap_us) {
    // stamps hold min start / max end over ALL replays of a node: only meaningful for ONE replay -> run one more, alone
    reset_stamps();
    if (graph) {
      hipGraph_t g; hipGraphExec_t x;
      CK(hipStreamBeginCapture(g_stream, hipStreamCaptureModeThreadLocal));
      issue();
      CK(hipStreamEndCapture(g_stream, &g));
      CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(x, g_stream));
      CK(hipStreamSynchronize(g_stream));
      reset_stamps();
      CK(hipGraphLaunch(x, g_stream));
      CK(hipStreamSynchronize(g_stream));
      CK(hipGraphExecDestroy(x)); CK(hipGraphDestroy(g));
    } else {
      issue();
      CK(hipStreamSynchronize(g_stream));
    }
    std::vector<unsigned long long> st(2 * NODES);
    CK(hipMemcpy(st.data(), g_stamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> body, gap;
    for (int i = 8; i < NODES; ++i) {
      body.push_back((double)(st[2 * i + 1] - st[2 * i]) * 0.01);
      gap.push_back((double)(st[2 * i] - st[2 * i - 1]) * 0.01);
    }
    std::sort(body.begin(), body.end()); std::sort(gap.begin(), gap.end());
    *body_us = body[body.size() / 2]; *gap_us = gap[gap.size() / 2];
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return (double)ms * 1e3 / ((double)NODES * replays);
}

int main(int argc, char** argv) {
  int prio_lo = 0, prio_hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
  const bool high = argc > 1 && !strcmp(argv[1], "high");
  if (high) CK(hipStreamCreateWithPriority(&g_stream, hipStreamNonBlocking, prio_hi));
  else CK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
  const size_t buf_bytes = 256u << 20;
  CK(hipMalloc(&g_buf, buf_bytes));
  CK(hipMemset(g_buf, 0, buf_bytes));
  CK(hipMalloc(&g_stamps, 2 * NODES * 8));
  set_lds_attr(96 * 1024);
  printf("# launch_floor_probe: %d dependent launches per chain, microseconds per launch (hipEvents around 10 graph replays / 10 eager passes)%s\n", NODES,
         high ? " [high-priority stream]" : "");
  printf("%-58s %8s %8s %10s %10s\n", "shape", "graph", "eager", "body(st)", "gap(st)");
  const std::vector<Shape> shapes = {
      {"grid 1 x 64, karg 64 B, LDS 0, empty", 64, 0, 64, 1, EMPTY, 0, false},
      {"grid 1 x 64, karg 64 B, LDS 0, empty, stamped", 64, 0, 64, 1, EMPTY, 0, true},
      {"grid 160 x 512, karg 64 B, empty", 64, 0, 512, 160, EMPTY, 0, false},
      {"grid 256 x 256, karg 64 B, empty", 64, 0, 256, 256, EMPTY, 0, false},
      {"grid 256 x 256, karg 64 B, empty, stamped", 64, 0, 256, 256, EMPTY, 0, true},
      {"grid 256 x 512, karg 64 B, empty", 64, 0, 512, 256, EMPTY, 0, false},
      {"grid 1024 x 256, karg 64 B, empty", 64, 0, 256, 1024, EMPTY, 0, false},
      {"grid 4096 x 256, karg 64 B, empty", 64, 0, 256, 4096, EMPTY, 0, false},
      {"grid 256 x 256, karg 512 B, empty", 512, 0, 256, 256, EMPTY, 0, false},
      {"grid 256 x 256, karg 2048 B, empty", 2048, 0, 256, 256, EMPTY, 0, false},
      {"grid 256 x 256, karg 4096 B, empty", 4096, 0, 256, 256, EMPTY, 0, false},
      {"grid 160 x 512, karg 512 B, LDS 52 KB, empty", 512, 52 * 1024, 512, 160, EMPTY, 0, false},
      {"grid 256 x 512, karg 64 B, LDS 52 KB, empty", 64, 52 * 1024, 512, 256, EMPTY, 0, false},
      {"grid 256 x 256, karg 64 B, scratch (64 floats/thread)", 64, 0, 256, 256, SCRATCH, 0, false},
      {"grid 256 x 256, plain stores 1 MB", 64, 0, 256, 256, STORE_PLAIN, 1, false},
      {"grid 256 x 256, plain stores 1 MB, stamped", 64, 0, 256, 256, STORE_PLAIN, 1, true},
      {"grid 256 x 256, plain stores 4 MB", 64, 0, 256, 256, STORE_PLAIN, 4, false},
      {"grid 256 x 256, plain stores 12 MB", 64, 0, 256, 256, STORE_PLAIN, 12, false},
      {"grid 256 x 256, nt stores 12 MB", 64, 0, 256, 256, STORE_NT, 12, false},
      {"grid 256 x 256, write-through stores 1 MB", 64, 0, 256, 256, STORE_WT, 1, false},
      {"grid 256 x 256, write-through stores 4 MB", 64, 0, 256, 256, STORE_WT, 4, false},
      {"grid 256 x 256, write-through stores 12 MB", 64, 0, 256, 256, STORE_WT, 12, false},
      {"grid 256 x 256, write-through stores 12 MB, stamped", 64, 0, 256, 256, STORE_WT, 12, true},
      {"grid 256 x 256, reads 1 MB", 64, 0, 256, 256, READ, 1, false},
      {"grid 256 x 256, reads 12 MB", 64, 0, 256, 256, READ, 12, false},
      {"grid 256 x 256, reads 12 MB, stamped", 64, 0, 256, 256, READ, 12, true},
  };
  for (const Shape& s : shapes) {
    double b = -1, g = -1;
    const double tg = time_chain({s}, true, 10, &b, &g);
    const double te = time_chain({s}, false, 10, nullptr, nullptr);
    if (s.stamps) printf("%-58s %8.2f %8.2f %10.2f %10.2f\n", s.name, tg, te, b, g);
    else printf("%-58s %8.2f %8.2f %10s %10s\n", s.name, tg, te, "-", "-");
    fflush(stdout);
  }
  // the engine's optimizer step in miniature: a writer of 12 MB (row pass), a reader of them that writes 1 MB (weight gradients), a
  // 1 MB read-modify-write (Adam): is a chain's time the sum of its kernels' own chains?
  const Shape A_wt = {"A", 512, 52 * 1024, 512, 160, STORE_WT, 10, true}, A_pl = {"A", 512, 52 * 1024, 512, 160, STORE_PLAIN, 10, true};
  const Shape B_rd = {"B", 2048, 0, 512, 256, READ, 6, true};
  const Shape C_wt = {"C", 64, 0, 256, 250, STORE_WT, 4, true}, C_pl = {"C", 64, 0, 256, 250, STORE_PLAIN, 4, true};
  const Shape T = {"T", 64, 0, 64, 1, EMPTY, 0, true};
  struct Chain { const char* name; std::vector<Shape> p; };
  const std::vector<Chain> chains = {
      {"[A: 160x512 wt 13 MB]", {A_wt}}, {"[A': the same, plain stores]", {A_pl}}, {"[B: 256x512 reads 12.6 MB, karg 2 KB]", {B_rd}},
      {"[C: 250x256 wt 4 MB]", {C_wt}}, {"[C': plain 4 MB]", {C_pl}}, {"[T: 1x64 empty]", {T}},
      {"[A, B, C] x 128", {A_wt, B_rd, C_wt}}, {"[A', B, C'] x 128", {A_pl, B_rd, C_pl}}, {"[A, T, B, T, C, T] x 64", {A_wt, T, B_rd, T, C_wt, T}},
      {"[A', T, B, T, C', T] x 64", {A_pl, T, B_rd, T, C_pl, T}},
  };
  printf("\n# chains (all stamped): us per launch from hipEvents; median stamped body / gap over the chain's launches\n");
  printf("%-58s %8s %8s %10s %10s\n", "chain", "graph", "eager", "body(st)", "gap(st)");
  for (const Chain& c : chains) {
    double b = -1, g = -1;
    const double tg = time_chain(c.p, true, 10, &b, &g);
    const double te = time_chain(c.p, false, 10, nullptr, nullptr);
    printf("%-58s %8.2f %8.2f %10.2f %10.2f\n", c.name, tg, te, b, g);
    fflush(stdout);
  }
  return 0;
}
