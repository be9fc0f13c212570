// gridsync_probe.hip — cost and cross-XCD correctness of a device-wide barrier inside one kernel (256 workgroups, one per CU).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gridsync_probe tools/gridsync_probe.hip && /tmp/gridsync_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Bar { unsigned count; unsigned fail; };

// mode 0: atomics only (no cache maintenance: data must travel through coherent sc0 sc1 accesses)
// mode 1: release (wbl2) + acquire (inv) by ONE wave per workgroup      mode 2: acquire by every wave
template <int MODE>
__device__ __forceinline__ void grid_sync(Bar* b, unsigned& epoch, unsigned nblocks) {
  __syncthreads();
  if (threadIdx.x == 0) {
    epoch += nblocks;
    if (MODE >= 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // write back this XCD's L2 (sc1)
    __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    long long t0 = 0;
    while (__hip_atomic_load(&b->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      if ((++spins & 1023u) == 0) {  // every wave leaves: a peer that never arrives ends the kernel after ~1 s instead of hanging the GPU
        if (__hip_atomic_load(&b->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        const long long now = wall_clock64();
        if (t0 == 0) t0 = now;
        if (now - t0 > 100000000ll) { __hip_atomic_store(&b->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every wave: invalidate L1 / non-local L2 lines
}

// each round: workgroup g writes round*1000+g into buf[g*stride .. +n), barrier, reads the slice of workgroup (g+37)%G and checks it
template <int MODE>
__global__ void __launch_bounds__(512) probe(Bar* b, float* buf, int n, int rounds, unsigned* errors, int payload) {
  unsigned epoch = 0;
  const int g = blockIdx.x, G = gridDim.x;
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    if (payload)
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (MODE == 0) __hip_atomic_store(&buf[(size_t)g * n + i], (float)(r * 1000 + g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else buf[(size_t)g * n + i] = (float)(r * 1000 + g);
      }
    grid_sync<MODE>(b, epoch, G);
    if (payload) {
      const int o = (g + 37) % G;
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = MODE == 0 ? __hip_atomic_load(&buf[(size_t)o * n + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : buf[(size_t)o * n + i];
        bad += v != (float)(r * 1000 + o);
      }
    }
    grid_sync<MODE>(b, epoch, G);  // readers done before the next round overwrites
  }
  if (bad) atomicAdd(errors, bad);
}

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int G = prop.multiProcessorCount;
  printf("CUs %d\n", G);
  Bar* b; float* buf; unsigned* err;
  const int n = 4096;
  CK(hipMalloc(&b, sizeof(Bar))); CK(hipMalloc(&buf, (size_t)G * n * 4)); CK(hipMalloc(&err, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode)
    for (int payload = 0; payload < 2; ++payload)
      for (int threads : {256, 512}) {
        auto k = mode == 0 ? probe<0> : mode == 1 ? probe<1> : probe<2>;
        CK(hipMemset(b, 0, sizeof(Bar))); CK(hipMemset(err, 0, 4));
        const int rounds = 500;
        hipLaunchKernelGGL(k, dim3(G), dim3(threads), 0, 0, b, buf, n, 10, err, payload);  // warm
        CK(hipDeviceSynchronize());
        CK(hipMemset(b, 0, sizeof(Bar)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(G), dim3(threads), 0, 0, b, buf, n, rounds, err, payload);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned herr; Bar hb; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, b, sizeof(Bar), hipMemcpyDeviceToHost));
        printf("mode=%d payload=%d threads=%d: %.3f us per barrier (%d barriers), errors=%u timeout=%u\n", mode, payload, threads, ms * 1e3 / (2 * rounds), 2 * rounds, herr, hb.fail);
      }
  // launch floor for comparison: 200 dependent empty kernels
  CK(hipEventRecord(e0));
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, (int*)nullptr);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("empty kernel, eager back-to-back: %.3f us each\n", ms * 1e3 / 200);
  return 0;
}
