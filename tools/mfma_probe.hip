// Microbenchmark: cycles per f32-input MFMA as a function of the number of independent accumulator chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0);
}
template <int NACC>
__global__ void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0);
}

template <typename K>
void run(const char* name, K kern, int nacc, int blocks, int threads, float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float cyc; hipMemcpy(&cyc, d + (1 << 20), 4, hipMemcpyDeviceToHost);
  printf("%-28s blocks %4d x %3d thr: %8.1f us total, %7.1f ns per MFMA per wave, s_memtime ticks/MFMA %.1f\n", name, blocks, threads, ms * 1e3,
         ms * 1e6 / (iters * nacc), cyc / (iters * nacc));
}

int main() {
  float* d; hipMalloc(&d, ((1 << 20) + 16) * 4);
  for (int rep = 0; rep < 2; ++rep) {
    run("32x32x2 f32, 1 acc", k32<1>, 1, 256, 256, d);
    run("32x32x2 f32, 2 acc", k32<2>, 2, 256, 256, d);
    run("32x32x2 f32, 4 acc", k32<4>, 4, 256, 256, d);
    run("32x32x2 f32, 1 acc, 1 wave/CU", k32<1>, 1, 256, 64, d);
    run("32x32x2 f32, 1 acc, 1 block", k32<1>, 1, 1, 64, d);
    run("16x16x4 f32, 1 acc", k16<1>, 1, 256, 256, d);
    run("16x16x4 f32, 2 acc", k16<2>, 2, 256, 256, d);
    run("16x16x4 f32, 4 acc", k16<4>, 4, 256, 256, d);
    run("16x16x4 f32, 8 acc", k16<8>, 8, 256, 256, d);
  }
  return 0;
}
