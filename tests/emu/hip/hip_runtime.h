// TEST INFRASTRUCTURE — a tiny SIMT emulator so that the HIP kernel *sources* under
// minppo_amd/csrc can be compiled with g++ and run on the CPU against the oracle
// (tests/test_emu_*.py).  It is never part of the product: minppo_amd loads only
// libminppo_hip.so and fails loudly without it.
//
// Model: a kernel launch runs its workgroups one after another; every work-item of a
// workgroup is a ucontext fiber; __syncthreads() yields to the scheduler, which resumes each
// live fiber once per barrier round.  Cross-lane primitives (emu wave_ops.h) exchange data
// through a scratch buffer around barriers, so they must be called in workgroup-uniform
// control flow (true for every kernel in csrc/).  Device pointers are host pointers.
#pragma once
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <ucontext.h>

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
typedef struct emu_stream* hipStream_t;
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
// code objects (mppo_model_attach_kernel): there is no device code on the emulator - loading one fails, and the caller says so
typedef struct emu_module* hipModule_t;
typedef struct emu_function* hipFunction_t;
typedef void* hipDeviceptr_t;
constexpr hipError_t hipErrorNotSupported = 801;
inline hipError_t hipModuleLoadData(hipModule_t*, const void*) { return hipErrorNotSupported; }
inline hipError_t hipModuleUnload(hipModule_t) { return hipSuccess; }
inline hipError_t hipModuleGetFunction(hipFunction_t*, hipModule_t, const char*) { return hipErrorNotSupported; }
inline hipError_t hipModuleGetGlobal(hipDeviceptr_t*, size_t*, hipModule_t, const char*) { return hipErrorNotSupported; }
inline hipError_t hipModuleLaunchKernel(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t, void**, void**) { return hipErrorNotSupported; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
// "device" allocations of the engine itself (the ranks' exchange buffers, csrc/k_peer.hip): POSIX shared memory, so that the rank
// PROCESSES of a CPU test can map each other's buffers the way hipIpc maps a peer GPU's (emu_stubs.cpp)
enum { hipDeviceMallocFinegrained = 1, hipDeviceMallocUncached = 3, hipIpcMemLazyEnablePeerAccess = 1 };
struct hipIpcMemHandle_t { char reserved[64]; };
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned flags);
hipError_t hipFree(void* p);
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p);
hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned flags);
hipError_t hipIpcCloseMemHandle(void* p);

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict

namespace emu {
extern dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
extern unsigned char* g_dyn_smem;
extern double g_xchg[4096];
extern const void* g_kernel_ptr;
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void barrier();
inline int tid() { return (int)(g_threadIdx.x + g_blockDim.x * (g_threadIdx.y + g_blockDim.y * g_threadIdx.z)); }
}  // namespace emu

#define threadIdx (::emu::g_threadIdx)
#define blockIdx (::emu::g_blockIdx)
#define blockDim (::emu::g_blockDim)
#define gridDim (::emu::g_gridDim)

inline void __syncthreads() { emu::barrier(); }

template <typename K, typename... Args>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
  emu::g_kernel_ptr = reinterpret_cast<const void*>(kernel);  // for diagnostics (dladdr)
  emu::launch(grid, block, shmem, [=]() { kernel(args...); });
}

// device math that <cmath> lacks
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
using std::isnan;
inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }

struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
inline float __expf(float x) { return expf(x); }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __fdividef(float a, float b) { return a / b; }
inline float2 make_float2(float x, float y) { return {x, y}; }
