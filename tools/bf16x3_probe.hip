// bf16x3_probe.hip - what would the float row pass cost on the bf16 matrix cores with three-term operands?
//
// The arithmetic skeleton of that kernel without its row logic: 160 workgroups of 8 waves (one per CU, as the training row pass), each wave
// walks three "layers" of 8 k-stages; per stage it takes in the THREE bf16 fragment copies (hi, mid, lo) of its two 16-column tiles of the
// weight matrix - 6 x 1 KB contiguous loads, 6 KB per wave and stage, 1 152 KB per workgroup and launch - through a ring of DEPTH stages, and
// issues the six products hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid per tile on v_mfma_f32_16x16x32_bf16 (12 MFMAs per stage, 288 per wave and
// launch), with A operands held in registers (in the kernel they would come from three bf16 LDS tiles).  An optional "adam" launch rewrites the
// weights between launches (cold = 1), as the optimizer does.  Variants: MFMAS = 0 (stream only), LOADS = 0 (MFMAs only: registers are reused).
// Reported: microseconds per launch from a hipGraph of 64 launches, HIP events.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bf16x3_probe.bin tools/bf16x3_probe.hip && tools/bf16x3_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef int i32x4n __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8n __attribute__((ext_vector_type(8)));
__device__ f32x4n raw_load_f32x4(i32x4n rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void raw_store_f32x4(f32x4n data, i32x4n rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");

__device__ __forceinline__ i32x4n make_rsrc(const void* base, unsigned bytes) {
  union { i32x4n v; struct { const void* p; unsigned n; unsigned f; } s; } u;
  u.s.p = base; u.s.n = bytes; u.s.f = 0x00020000;
  return u.v;
}

constexpr int kLayers = 3, kStagesPerLayer = 8, kStages = kLayers * kStagesPerLayer, kWaves = 8, kCopies = 3;
constexpr int kStageBytes = 2048 * kCopies;                       // per wave and stage: 3 copies x 2 tiles x 1 KB
constexpr int kNetBytes = kStages * kWaves * kStageBytes;         // 1 152 KB per network

template <int DEPTH, bool LOADS, bool MFMAS>
__global__ void __launch_bounds__(512, 2) probe_kernel(const unsigned char* w, float* sink, int empty) {
  if (empty) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), net = blockIdx.x & 1;
  const i32x4n r = make_rsrc(w + (size_t)net * kNetBytes, kNetBytes);
  f32x4n q[DEPTH][6];
  f32x4n acc[2][3] = {};  // two column tiles x (three independent accumulation chains: the six products spread over them)
  // A operands (three terms) as raw bf16 bits in registers
  f32x4n a_bits[3];
  for (int c = 0; c < 3; ++c) a_bits[c] = f32x4n{1.0f + lane + c, 2.0f, 3.0f, 4.0f};
  auto load = [&](int S, int slot) {
#pragma unroll
    for (int k = 0; k < 6; ++k) q[slot][k] = LOADS ? raw_load_f32x4(r, lane * 16 + 1024 * k, (S * kWaves + wave) * kStageBytes, 0) : f32x4n{(float)S, 1.f, 2.f, 3.f};
  };
#pragma unroll
  for (int S = 0; S < DEPTH - 1; ++S) load(S, S);
#pragma unroll
  for (int S = 0; S < kStages; ++S) {
    if (S + DEPTH - 1 < kStages) load(S + DEPTH - 1, (S + DEPTH - 1) % DEPTH);
    const int slot = S % DEPTH;
    if (MFMAS) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8n bh = __builtin_bit_cast(bf16x8n, q[slot][3 * t + 0]), bm = __builtin_bit_cast(bf16x8n, q[slot][3 * t + 1]), bl = __builtin_bit_cast(bf16x8n, q[slot][3 * t + 2]);
        const bf16x8n ah = __builtin_bit_cast(bf16x8n, a_bits[0]), am = __builtin_bit_cast(bf16x8n, a_bits[1]), al = __builtin_bit_cast(bf16x8n, a_bits[2]);
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc[t][1], 0, 0, 0);
        acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc[t][2], 0, 0, 0);
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[t][1], 0, 0, 0);
        acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc[t][2], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k & 1][k % 3] += q[slot][k];
    }
  }
  float out = 0.f;
  for (int t = 0; t < 2; ++t) for (int c = 0; c < 3; ++c) out += acc[t][c].x + acc[t][c].y + acc[t][c].z + acc[t][c].w;
  if (out == 12345.678f) sink[threadIdx.x] = out;  // (never true: keeps everything alive)
}

__global__ void __launch_bounds__(256) adam_like(unsigned char* w, int n16, float v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n16) raw_store_f32x4(f32x4n{v, v, v, v}, make_rsrc(w, 0x7FFFFFFFu), i * 16, 0, 17);
}

template <int DEPTH, bool LOADS, bool MFMAS>
static double run(unsigned char* w, float* sink, int grid, int cold, int empty, hipStream_t s) {
  constexpr int LAUNCHES = 64;
  const int n16 = 2 * kNetBytes / 16;
  auto seq = [&] {
    for (int k = 0; k < LAUNCHES; ++k) {
      if (cold) hipLaunchKernelGGL(adam_like, dim3((n16 + 255) / 256), dim3(256), 0, s, w, n16, 0.001f * (float)(k & 7));
      hipLaunchKernelGGL((probe_kernel<DEPTH, LOADS, MFMAS>), dim3(grid), dim3(512), 0, s, w, sink, empty);
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

template <int DEPTH, bool LOADS, bool MFMAS>
static void report(const char* what, unsigned char* w, float* sink, hipStream_t s) {
  for (int cold = 0; cold < 2; ++cold) {
    const double base = run<DEPTH, LOADS, MFMAS>(w, sink, 160, cold, 1, s), t = run<DEPTH, LOADS, MFMAS>(w, sink, 160, cold, 0, s);
    printf("%-44s %s  %6.2f us per step (empty: %5.2f)  -> %5.2f us of kernel body\n", what, cold ? "after adam-like rewrite" : "warm                   ", t, base, t - base);
  }
}

int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  unsigned char* w; float* sink;
  CK(hipMalloc(&w, 2 * kNetBytes)); CK(hipMalloc(&sink, 4096));
  CK(hipMemset(w, 0, 2 * kNetBytes));
  printf("three-term bf16 skeleton of the float row pass: 160 workgroups x 8 waves, 24 stages x 6 KB per wave = 1 152 KB per workgroup, 288 MFMA 16x16x32 per wave\n");
  report<3, true, true>("stream + six products, ring depth 3", w, sink, s);
  report<4, true, true>("stream + six products, ring depth 4", w, sink, s);
  report<6, true, true>("stream + six products, ring depth 6", w, sink, s);
  report<3, true, false>("stream only, ring depth 3", w, sink, s);
  report<6, true, false>("stream only, ring depth 6", w, sink, s);
  report<3, false, true>("six products only (no loads)", w, sink, s);
  return 0;
}
