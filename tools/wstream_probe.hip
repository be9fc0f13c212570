// wstream_probe.hip — how fast can ONE CU take in its weight slabs from its XCD's L2, by request pattern?
//
// The bf16 training row pass (csrc/fused_bf16.h) is bound by this stream: every workgroup (16 rows x one network, 8 waves, one per CU,
// 160 of them) needs the 384 KB of bf16 weight fragments of its network - W1, W2, W2^T, 16 KB per wave and matrix - and nothing else in
// the kernel moves comparable bytes (profiles/r05_*_fused_phases_bf16.txt).  This probe replays exactly that access pattern without the
// arithmetic: GRID workgroups of 512 threads, workgroup b reads network b & 1; a wave reads its 3 x 8 stages of 2 x 1 KB (one 16-byte load
// per lane = 1 KB contiguous per instruction) with DEPTH stages requested ahead of the one it consumes (consumption = one v_add per
// register, which is what places the s_waitcnt), under a cache policy (0 default, 2 nt, 16 sc1, 17 sc0 sc1).  Between timed launches an
// optional "adam" kernel rewrites the 768 KB with write-through stores, as the optimizer does between two row passes (cold = 1).
// Reported: microseconds per launch (HIP events over a hipGraph of LAUNCHES dependent launches) and GB/s per CU = 384 KB / (launch - EMPTY launch).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/wstream_probe.bin tools/wstream_probe.hip && tools/wstream_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef int i32x4n __attribute__((ext_vector_type(4)));
__device__ f32x4n raw_load_f32x4(i32x4n rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void raw_store_f32x4(f32x4n data, i32x4n rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");

__device__ __forceinline__ i32x4n make_rsrc(const void* base, unsigned bytes) {
  union { i32x4n v; struct { const void* p; unsigned n; unsigned f; } s; } u;
  u.s.p = base; u.s.n = bytes; u.s.f = 0x00020000;
  return u.v;
}

constexpr int kStages = 24;           // 3 matrices x 8 stages of 32 k
constexpr int kNetBytes = 384 * 1024; // one network's fragments
constexpr int kWaves = 8;

// DEPTH stages in flight; POLICY = aux bits of the buffer load
template <int DEPTH, int POLICY>
__global__ void __launch_bounds__(512, 2) stream_kernel(const unsigned char* w, float* sink, int empty) {
  if (empty) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), net = blockIdx.x & 1;
  const i32x4n r = make_rsrc(w + (size_t)net * kNetBytes, kNetBytes);
  // stage S of this wave: block (S, wave) of 2 KB, tile-major: two loads of 1 KB contiguous
  f32x4n q[kStages][2];
  float acc = 0.f;
#pragma unroll
  for (int S = 0; S < kStages + DEPTH; ++S) {
    if (S < kStages) {
      q[S][0] = raw_load_f32x4(r, lane * 16, (S * kWaves + wave) * 2048, POLICY);
      q[S][1] = raw_load_f32x4(r, lane * 16 + 1024, (S * kWaves + wave) * 2048, POLICY);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (S >= DEPTH) {
      const int C = S - DEPTH;
      acc += q[C][0].x + q[C][0].y + q[C][0].z + q[C][0].w + q[C][1].x + q[C][1].y + q[C][1].z + q[C][1].w;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (acc == 12345.678f) sink[threadIdx.x] = acc;  // (never true: keeps the loads alive)
}

// BOTH networks' fragments in one workgroup (the "one workgroup per row tile, both networks, shared x tile" decomposition): 768 KB per CU
template <int DEPTH>
__global__ void __launch_bounds__(512, 2) stream2_kernel(const unsigned char* w, float* sink, int empty) {
  if (empty) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const i32x4n r = make_rsrc(w, 2 * kNetBytes);
  f32x4n q[2 * kStages][2];
  float acc = 0.f;
#pragma unroll
  for (int S = 0; S < 2 * kStages + DEPTH; ++S) {
    if (S < 2 * kStages) {
      q[S][0] = raw_load_f32x4(r, lane * 16, (S * kWaves + wave) * 2048, 0);
      q[S][1] = raw_load_f32x4(r, lane * 16 + 1024, (S * kWaves + wave) * 2048, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (S >= DEPTH) {
      const int C = S - DEPTH;
      acc += q[C][0].x + q[C][0].y + q[C][0].z + q[C][0].w + q[C][1].x + q[C][1].y + q[C][1].z + q[C][1].w;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (acc == 12345.678f) sink[threadIdx.x] = acc;
}

// PRIVATE copies (round 5, last): one copy of both networks' fragments per XCD, written by workgroups of THAT XCD (workgroup id mod 8 = XCD: the
// round-robin placement the engine's tile orders already rely on) with ordinary write-back stores, read only by workgroups of that XCD - is the
// stream then as fast behind the rewrite as it is warm?  (what an optimizer that runs redundantly on every XCD would buy the row pass)
template <int DEPTH>
__global__ void __launch_bounds__(512, 2) stream_private_kernel(const unsigned char* w, float* sink, int empty) {
  if (empty) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), xcd = blockIdx.x & 7, net = (blockIdx.x >> 3) & 1;
  const i32x4n r = make_rsrc(w + ((size_t)xcd * 2 + net) * kNetBytes, kNetBytes);
  f32x4n q[kStages][2];
  float acc = 0.f;
#pragma unroll
  for (int S = 0; S < kStages + DEPTH; ++S) {
    if (S < kStages) {
      q[S][0] = raw_load_f32x4(r, lane * 16, (S * kWaves + wave) * 2048, 0);
      q[S][1] = raw_load_f32x4(r, lane * 16 + 1024, (S * kWaves + wave) * 2048, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (S >= DEPTH) {
      const int C = S - DEPTH;
      acc += q[C][0].x + q[C][0].y + q[C][0].z + q[C][0].w + q[C][1].x + q[C][1].y + q[C][1].z + q[C][1].w;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (acc == 12345.678f) sink[threadIdx.x] = acc;
}
// every XCD rewrites ITS copy (block b: XCD b & 7, chunk b >> 3); policy 0 = write-back into the XCD's own L2
template <int POLICY>
__global__ void __launch_bounds__(256) adam_private(unsigned char* w, int n16, float v) {
  const int xcd = blockIdx.x & 7, i = (blockIdx.x >> 3) * 256 + threadIdx.x;
  if (i < n16) raw_store_f32x4(f32x4n{v, v, v, v}, make_rsrc(w + (size_t)xcd * 2 * kNetBytes, 2 * kNetBytes), i * 16, 0, POLICY);
}

// the optimizer's stand-in: rewrites all weights with write-through stores (values stay finite)
__global__ void __launch_bounds__(256) adam_like(unsigned char* w, int n16, float v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n16) raw_store_f32x4(f32x4n{v, v, v, v}, make_rsrc(w, 0x7FFFFFFFu), i * 16, 0, 17);
}

template <int DEPTH, int POLICY>
static double run(unsigned char* w, float* sink, int grid, int cold, int empty, hipStream_t s) {
  constexpr int LAUNCHES = 64;
  const int n16 = 2 * kNetBytes / 16;
  auto seq = [&] {
    for (int k = 0; k < LAUNCHES; ++k) {
      if (cold) hipLaunchKernelGGL(adam_like, dim3((n16 + 255) / 256), dim3(256), 0, s, w, n16, 0.001f * (float)(k & 7));
      hipLaunchKernelGGL((stream_kernel<DEPTH, POLICY>), dim3(grid), dim3(512), 0, s, w, sink, empty);
    }
  };
  seq();
  CK(hipStreamSynchronize(s));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  seq();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  for (int rep = 0; rep < 5; ++rep) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return 1e3 * ms / (5.0 * LAUNCHES);
}

template <int DEPTH, int POLICY>
static void report(const char* what, unsigned char* w, float* sink, int grid, hipStream_t s) {
  for (int cold = 0; cold < 2; ++cold) {
    const double base = run<DEPTH, POLICY>(w, sink, grid, cold, 1, s), t = run<DEPTH, POLICY>(w, sink, grid, cold, 0, s);
    printf("%-34s grid %3d %s  %6.2f us per step (empty: %5.2f)  -> stream %5.2f us = %6.1f GB/s per CU\n", what, grid, cold ? "after adam-like rewrite" : "warm                   ", t, base,
           t - base, 384.0 * 1024.0 / ((t - base) * 1e-6) / 1e9);
  }
}

template <int POLICY>
static double run_private(unsigned char* w8, float* sink, int cold, int empty, hipStream_t s) {
  constexpr int LAUNCHES = 64;
  const int n16 = 2 * kNetBytes / 16;
  auto seq = [&] {
    for (int k = 0; k < LAUNCHES; ++k) {
      if (cold) hipLaunchKernelGGL(adam_private<POLICY>, dim3(8 * ((n16 + 255) / 256)), dim3(256), 0, s, w8, n16, 0.001f * (float)(k & 7));
      hipLaunchKernelGGL((stream_private_kernel<8>), dim3(160), dim3(512), 0, s, w8, sink, empty);
    }
  };
  seq();
  CK(hipStreamSynchronize(s));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  seq();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  for (int rep = 0; rep < 5; ++rep) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return 1e3 * ms / (5.0 * LAUNCHES);
}

static double run2(unsigned char* w, float* sink, int grid, int cold, int empty, hipStream_t s) {
  constexpr int LAUNCHES = 64;
  const int n16 = 2 * kNetBytes / 16;
  auto seq = [&] {
    for (int k = 0; k < LAUNCHES; ++k) {
      if (cold) hipLaunchKernelGGL(adam_like, dim3((n16 + 255) / 256), dim3(256), 0, s, w, n16, 0.001f * (float)(k & 7));
      hipLaunchKernelGGL((stream2_kernel<4>), dim3(grid), dim3(512), 0, s, w, sink, empty);
    }
  };
  seq();
  CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s));
  for (int rep = 0; rep < 5; ++rep) seq();
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return 1e3 * ms / (5.0 * LAUNCHES);
}

int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  unsigned char* w; float* sink;
  CK(hipMalloc(&w, 2 * kNetBytes)); CK(hipMalloc(&sink, 4096));
  CK(hipMemset(w, 0, 2 * kNetBytes));
  printf("weight stream per CU: 8 waves x 24 stages x 2 KB = 384 KB per workgroup, one workgroup per CU\n");
  for (int cold = 0; cold < 2; ++cold) {  // (eager launches for both arms of this comparison)
    const double b = run2(w, sink, 80, cold, 1, s), t = run2(w, sink, 80, cold, 0, s);
    printf("BOTH networks per workgroup (768 KB per CU), depth 4, grid  80 %s  %6.2f us per step (empty: %5.2f)  -> stream %5.2f us = %6.1f GB/s per CU\n",
           cold ? "after adam-like rewrite" : "warm                   ", t, b, t - b, 768.0 * 1024.0 / ((t - b) * 1e-6) / 1e9);
  }
  {
    unsigned char* w8;
    CK(hipMalloc(&w8, 16 * (size_t)kNetBytes));
    CK(hipMemset(w8, 0, 16 * (size_t)kNetBytes));
    for (int cold = 0; cold < 2; ++cold) {
      const double b = run_private<0>(w8, sink, cold, 1, s), t = run_private<0>(w8, sink, cold, 0, s);
      printf("PRIVATE copy per XCD, depth 8, grid 160 %s  %6.2f us per step (empty: %5.2f)  -> stream %5.2f us = %6.1f GB/s per CU\n",
             cold ? "after a rewrite by the XCD's own workgroups (write-back)" : "warm                                                    ", t, b, t - b,
             384.0 * 1024.0 / ((t - b) * 1e-6) / 1e9);
    }
    {  // the same with the engine's write-through stores (sc0 sc1): does the writer's L2 keep the line?
      const double b = run_private<17>(w8, sink, 1, 1, s), t = run_private<17>(w8, sink, 1, 0, s);
      printf("PRIVATE copy per XCD, depth 8, grid 160 after a rewrite by the XCD's own workgroups (write-through sc0 sc1)  %6.2f us per step (empty: %5.2f)  -> stream %5.2f us = %6.1f GB/s per CU\n",
             t, b, t - b, 384.0 * 1024.0 / ((t - b) * 1e-6) / 1e9);
    }
    CK(hipFree(w8));
  }
  for (int grid : {160, 80, 256}) {
    report<1, 0>("depth 1, default", w, sink, grid, s);
    report<3, 0>("depth 3, default", w, sink, grid, s);
    report<8, 0>("depth 8, default", w, sink, grid, s);
    report<24, 0>("everything up front, default", w, sink, grid, s);
    report<8, 2>("depth 8, nt", w, sink, grid, s);
    report<8, 17>("depth 8, sc0 sc1", w, sink, grid, s);
  }
  return 0;
}
