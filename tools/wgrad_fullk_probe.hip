// Prototype microbenchmark: weight gradients with the FULL K range reduced inside one workgroup (no split-K slabs, no
// reduce kernel).  256 workgroups x 8 waves; a workgroup owns a 32x32 output tile, wave g owns K rows [160g, 160g+160) and
// streams its operands straight from a k-quad-blocked layout ([K/4][cols][4]: the four k's of a column are one float4) into
// MFMA registers; the 8 accumulators are summed through LDS.  Question: does this beat split-K GEMM (16.7 us) + reduce (5.0 us)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Prob { const float* A; const float* B; float* C; int M, N; };
struct Args { Prob p[4]; int K; float* sq; };

template <int DEPTH>
__global__ void __launch_bounds__(512, 2) fullk(Args a) {
  __shared__ float red[8][32 * 33];
  const int t = threadIdx.x, lane = t & 63, g = t >> 6, i = lane & 31, h = lane >> 5;
  const int wg = blockIdx.x, pi = wg >> 6, lt = wg & 63;
  const Prob p = a.p[pi];
  const int mt = lt >> 3, nt = lt & 7, m0 = mt * 32, n0 = nt * 32;
  const int q0 = g * (a.K / 32);  // first quad of this wave: K/8 rows = K/32 quads
  const int nst = a.K / 8 / 32;   // stages of 32 k (8 quads)
  const float4* A4 = reinterpret_cast<const float4*>(p.A);
  const float4* B4 = reinterpret_cast<const float4*>(p.B);
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 ra[DEPTH][4], rb[DEPTH][4];
  auto load = [&](int slot, int s) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int q = q0 + 8 * s + 2 * c + h;
      ra[slot][c] = A4[(size_t)q * p.M + m0 + i];
      rb[slot][c] = B4[(size_t)q * p.N + n0 + i];
    }
  };
  auto mm = [&](int slot) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[slot][c].x, rb[slot][c].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[slot][c].y, rb[slot][c].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[slot][c].z, rb[slot][c].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[slot][c].w, rb[slot][c].w, acc, 0, 0, 0);
    }
  };
  static_assert(DEPTH == 5, "prototype: the whole K range of a wave in flight");
  load(0, 0); load(1, 1); load(2, 2); load(3, 3); load(4, 4);
  __builtin_amdgcn_sched_barrier(0);
  mm(0); __builtin_amdgcn_sched_barrier(0); mm(1); __builtin_amdgcn_sched_barrier(0); mm(2); __builtin_amdgcn_sched_barrier(0); mm(3); __builtin_amdgcn_sched_barrier(0); mm(4);
  (void)nst;
  for (int r = 0; r < 16; ++r) red[g][((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + i] = acc[r];
  __syncthreads();
  float sq = 0.f;
  for (int e = t; e < 1024; e += 512) {
    const int row = e >> 5, col = e & 31;
    float v = 0.f;
    for (int k = 0; k < 8; ++k) v += red[k][row * 33 + col];
    p.C[(size_t)(m0 + row) * p.N + n0 + col] = v;
    sq += v * v;
  }
  for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m);
  __shared__ float sred[8];
  if (lane == 0) sred[g] = sq;
  __syncthreads();
  if (t == 0) { float s = 0.f; for (int k = 0; k < 8; ++k) s += sred[k]; a.sq[wg] = s; }
}

int main() {
  const int K = 1280, M = 256, N = 256;
  Args a{};
  a.K = K;
  std::vector<float> hA((size_t)K * M), hB((size_t)K * N);
  for (size_t j = 0; j < hA.size(); ++j) { hA[j] = (float)((j * 2654435761u) >> 20 & 255) / 256.f - 0.5f; hB[j] = (float)((j * 40503u) >> 7 & 255) / 256.f - 0.5f; }
  for (int k = 0; k < 4; ++k) {
    float *dA, *dB, *dC;
    CHECK(hipMalloc(&dA, hA.size() * 4)); CHECK(hipMalloc(&dB, hB.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    a.p[k] = Prob{dA, dB, dC, M, N};
  }
  CHECK(hipMalloc(&a.sq, 256 * 4));
  hipStream_t s; CHECK(hipStreamCreate(&s));
  for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(fullk<5>, dim3(256), dim3(512), 0, s, a);
  CHECK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, s));
  for (int it = 0; it < 500; ++it) hipLaunchKernelGGL(fullk<5>, dim3(256), dim3(512), 0, s, a);
  CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  // check one element on the host (quad layout: element (k, m) at ((k/4)*M + m)*4 + k%4)
  std::vector<float> hC((size_t)M * N);
  CHECK(hipMemcpy(hC.data(), a.p[0].C, hC.size() * 4, hipMemcpyDeviceToHost));
  double ref = 0; const int m = 37, n = 201;
  for (int k = 0; k < K; ++k) ref += (double)hA[((size_t)(k / 4) * M + m) * 4 + k % 4] * hB[((size_t)(k / 4) * N + n) * 4 + k % 4];
  printf("full-K wgrad prototype: %.2f us per launch (500 back-to-back launches, events); C[37][201] = %.5f (ref %.5f)\n", ms * 1e3 / 500, hC[m * N + n], ref);
  return 0;
}
