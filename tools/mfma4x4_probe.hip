// tools/mfma4x4_probe.hip - issue rate and semantics of v_mfma_f32_4x4x1_16b_f32 on gfx950 (stand-alone):
//   hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_probe.hip -o tools/mfma4x4_probe.bin && tools/mfma4x4_probe.bin
// 1. layout check: 16 blocks, block b: D[i][j] += A[i] * B[j] with A in lane 4b + i, B in lane 4b + j, D[i][j] in lane 4b + j, register i
// 2. rate: one wave per SIMD issues N MFMAs on 1 / 2 / 3 / 6 rotating accumulators; cycles per instruction, and the same for 16x16x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float* a, const float* b, float* d) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[threadIdx.x * 4 + r] = acc[r];
}

template <int NACC, bool BIG>
__global__ void rate_kernel(float* out, int iters, unsigned long long* ticks) {
  f32x4 acc[NACC];
  for (int q = 0; q < NACC; ++q) acc[q] = {0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.f + threadIdx.x * 0.002f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
      if (BIG) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
      else acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[q], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

template <int NACC, bool BIG>
static void run(const char* name, int waves_per_block) {
  float* out; unsigned long long* t;
  hipMalloc(&out, 4 * 64 * 8 * 1024); hipMalloc(&t, 8);
  const int iters = 4096;
  rate_kernel<NACC, BIG><<<256, 64 * waves_per_block>>>(out, iters, t);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  rate_kernel<NACC, BIG><<<256, 64 * waves_per_block>>>(out, iters, t);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * NACC;
  printf("%-22s acc=%d waves/block=%d: %.2f shader cycles per MFMA (wave 0), %.1f ns per MFMA and wave by the event clock\n", name, NACC, waves_per_block, (double)h / n, ms * 1e6 / n);
  hipFree(out); hipFree(t);
}

int main() {
  std::vector<float> a(64), b(64), d(256);
  for (int l = 0; l < 64; ++l) { a[l] = 1.f + l; b[l] = 100.f + l; }
  float *da, *db, *dd;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
  hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
  layout_kernel<<<1, 64>>>(da, db, dd);
  hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int blk = 0; blk < 16; ++blk)
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j)
        if (d[(4 * blk + j) * 4 + i] != a[4 * blk + i] * b[4 * blk + j]) ++bad;
  printf("layout D[blk][i][j] (lane 4 blk + j, register i) = A[lane 4 blk + i] * B[lane 4 blk + j]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
  run<1, false>("4x4x1", 1); run<2, false>("4x4x1", 1); run<3, false>("4x4x1", 1); run<6, false>("4x4x1", 1);
  run<3, false>("4x4x1", 4); run<3, false>("4x4x1", 8);
  run<1, true>("16x16x4", 1); run<2, true>("16x16x4", 1); run<2, true>("16x16x4", 4); run<2, true>("16x16x4", 8);
  return 0;
}
