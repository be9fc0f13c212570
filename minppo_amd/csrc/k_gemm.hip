// k_gemm.hip — batched small GEMMs on the f32-input matrix cores (v_mfma_f32_32x32x2_f32).
//
// Every dense contraction of the actor / critic MLPs goes through this kernel: the forward
// layers of `MLP.__call__` (reference minppo/train.py:56-68), the activation-gradient products
// dH = dZ.W^T and the weight-gradient products dW = H^T.dZ that `jax.value_and_grad` derives
// from them (train.py:246).  Exact f32 arithmetic (the MFMA is an ordered fmaf chain), so the
// results are comparable with the f32 JAX reference at summation-order tolerance.
//
// Shape regime: M = 1280 (minibatch) or 4096 (rollout) rows, N = K = 256 and smaller.  A
// workgroup = 4 waves = one 64x64 tile of C (each wave one 32x32 MFMA accumulator); the two
// networks (and all six weight gradients) share one launch through blockIdx.z so that a launch
// carries 160-1100 workgroups.  Operand tiles are staged through LDS k-major (As[k][m],
// Bs[k][n], row stride 68 floats): the MFMA operand fetch is then one conflict-free ds_read_b32
// per operand per MFMA, negligible beside the 64-cycle f32 MFMA.  The f32 MFMA rate
// (157 TFLOP/s) is the roofline for these kernels; at these sizes the achieved fraction is set by
// tile count / wave occupancy rather than by memory.
#include <wave_ops.h>

#include "gemm.h"
#include "mppo_common.h"

namespace mppo {

constexpr int BM = 64, BN = 64, BK = 16, LDT = 68, GEMM_THREADS = 256;

// Tile loader, memory contiguous along k: element (r, k) at base[row(r)*ld + k]; LDS image T[k][r].
__device__ __forceinline__ void load_tile_kc(float* T, const float* base, int ld, const int* gather, int r0, int R, int k0, int kend, int t) {
  const int r = t >> 2, kq = (t & 3) * 4;
  const int gr = r0 + r;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (gr < R) {
    const long row = gather ? gather[gr] : gr;
    const float* ptr = base + row * (long)ld + k0 + kq;
    const bool vec = (k0 + kq + 3 < kend) && ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
    if (vec) {
      const float4 q = *reinterpret_cast<const float4*>(ptr);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
      for (int c = 0; c < 4; ++c) if (k0 + kq + c < kend) v[c] = ptr[c];
    }
  }
  for (int c = 0; c < 4; ++c) T[(kq + c) * LDT + r] = v[c];
}

// Tile loader, memory contiguous along r: element (r, k) at base[row(k)*ld + r]; LDS image T[k][r].
// Rmem = number of r-columns that exist in memory; logical column R-1 (>= Rmem) is all ones when ones_row.
__device__ __forceinline__ void load_tile_rc(float* T, const float* base, int ld, const int* gather, int r0, int R, int Rmem, bool ones_row,
                                             int k0, int kend, int t) {
  const int k = t >> 4, rq = (t & 15) * 4;
  const int gk = k0 + k;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (gk < kend) {
    const long row = gather ? gather[gk] : gk;
    const float* ptr = base + row * (long)ld + r0 + rq;
    const bool vec = (r0 + rq + 3 < Rmem) && ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
    if (vec) {
      q = *reinterpret_cast<const float4*>(ptr);
    } else {
      float v[4];
      for (int c = 0; c < 4; ++c) {
        const int rr = r0 + rq + c;
        v[c] = rr < Rmem ? ptr[c] : ((ones_row && rr == R - 1) ? 1.f : 0.f);
      }
      q = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  *reinterpret_cast<float4*>(T + k * LDT + rq) = q;
}

template <bool A_T, bool B_T, int EPI>
__global__ void __launch_bounds__(GEMM_THREADS) gemm_kernel(GemmBatch gb) {
  __shared__ __attribute__((aligned(16))) float As[BK * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDT];
  const int z = blockIdx.z;
  const int pi = z / gb.ksplit, ks = z - pi * gb.ksplit;
  const GemmProb p = gb.p[pi];
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (m0 >= p.M || n0 >= p.N) return;  // workgroup-uniform
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, hi = lane >> 5;
  int kper = (p.K + gb.ksplit - 1) / gb.ksplit;
  kper = (kper + BK - 1) / BK * BK;
  const int kb = ks * kper;
  const int ke = p.K < kb + kper ? p.K : kb + kper;
  const int MA = (A_T && p.ones_row) ? p.M - 1 : p.M;  // rows of op(A) that exist in memory

  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int k0 = kb; k0 < ke; k0 += BK) {
    if (A_T) load_tile_rc(As, p.A, p.lda, p.gather, m0, p.M, MA, p.ones_row != 0, k0, ke, t);
    else load_tile_kc(As, p.A, p.lda, p.gather, m0, p.M, k0, ke, t);
    if (B_T) load_tile_kc(Bs, p.B, p.ldb, nullptr, n0, p.N, k0, ke, t);
    else load_tile_rc(Bs, p.B, p.ldb, nullptr, n0, p.N, p.N, false, k0, ke, t);
    __syncthreads();
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = As[(kk + hi) * LDT + wr * 32 + l31];
      const float b = Bs[(kk + hi) * LDT + wc * 32 + l31];
      mfma_f32_32x32x2(a, b, acc);
    }
    __syncthreads();
  }

  float* C = p.C + (EPI == EPI_STORE ? (size_t)ks * gb.slab_stride : 0);
  const int col = n0 + wc * 32 + l31;
  if (col < p.N) {
    float bias = 0.f;
    if (EPI == EPI_BIAS_ACT && p.bias) bias = p.bias[col];
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (row < p.M) {
        float v = acc[r];
        if (EPI == EPI_BIAS_ACT) {
          v += bias;
          if (p.act == ACT_TANH) v = tanhf(v);
          else if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
        } else if (EPI == EPI_DACT) {
          const float hval = p.aux[(size_t)row * p.ldaux + col];
          if (p.act == ACT_TANH) v *= (1.f - hval * hval);
          else if (p.act == ACT_RELU) v = hval > 0.f ? v : 0.f;
        }
        C[(size_t)row * p.ldc + col] = v;
      }
    }
  }
}

template <bool A_T, bool B_T, int EPI>
static int32_t launch_t(const GemmBatch& gb, hipStream_t stream) {
  int maxM = 0, maxN = 0;
  for (int i = 0; i < gb.count; ++i) {
    maxM = gb.p[i].M > maxM ? gb.p[i].M : maxM;
    maxN = gb.p[i].N > maxN ? gb.p[i].N : maxN;
  }
  dim3 grid(cdiv(maxN, BN), cdiv(maxM, BM), gb.count * gb.ksplit);
  hipLaunchKernelGGL((gemm_kernel<A_T, B_T, EPI>), grid, dim3(GEMM_THREADS), 0, stream, gb);
  MPPO_CHECK_LAUNCH("gemm_kernel");
  return MPPO_OK;
}

int32_t gemm_launch(const GemmBatch& gb, int a_t, int b_t, int epi, int bf16, hipStream_t stream) {
  MPPO_REQUIRE(gb.count >= 1 && gb.count <= kGemmMaxProb, "gemm_launch: %d problems", gb.count);
  MPPO_REQUIRE(gb.ksplit >= 1 && (gb.ksplit == 1 || epi == EPI_STORE), "gemm_launch: split-K only with EPI_STORE");
  MPPO_REQUIRE(bf16 == 0, "gemm_launch: the bf16 MFMA path is not built into this library");
  for (int i = 0; i < gb.count; ++i) {
    const GemmProb& p = gb.p[i];
    MPPO_REQUIRE(p.A && p.B && p.C && p.M >= 1 && p.N >= 1 && p.K >= 1, "gemm_launch: problem %d malformed (M=%d N=%d K=%d)", i, p.M, p.N, p.K);
    MPPO_REQUIRE(epi != EPI_DACT || p.aux, "gemm_launch: EPI_DACT needs aux");
    MPPO_REQUIRE(!p.ones_row || a_t, "gemm_launch: ones_row needs a transposed A");
  }
  const int v = a_t * 2 + b_t;
  if (epi == EPI_BIAS_ACT && v == 0) return launch_t<false, false, EPI_BIAS_ACT>(gb, stream);
  if (epi == EPI_DACT && v == 1) return launch_t<false, true, EPI_DACT>(gb, stream);
  if (epi == EPI_STORE && v == 2) return launch_t<true, false, EPI_STORE>(gb, stream);
  return fail(MPPO_EINVAL, "gemm_launch: variant a_t=%d b_t=%d epi=%d is not instantiated", a_t, b_t, epi);
}

}  // namespace mppo

extern "C" int32_t mppo_gemm_batch(const mppo_gemm_desc_t* probs, int32_t count, int32_t variant, int32_t ksplit, size_t slab_stride, int32_t bf16,
                                   void* stream) {
  using namespace mppo;
  MPPO_REQUIRE(probs && count >= 1 && count <= kGemmMaxProb, "mppo_gemm_batch: 1..%d problems", kGemmMaxProb);
  MPPO_REQUIRE(variant >= 0 && variant <= 2, "mppo_gemm_batch: variant %d", variant);
  GemmBatch gb{};
  gb.count = count; gb.ksplit = ksplit < 1 ? 1 : ksplit; gb.slab_stride = slab_stride;
  for (int i = 0; i < count; ++i) {
    const mppo_gemm_desc_t& d = probs[i];
    GemmProb& p = gb.p[i];
    p.A = d.A; p.B = d.B; p.C = d.C; p.bias = d.bias; p.aux = d.aux; p.gather = d.gather; p.M = d.M; p.N = d.N; p.K = d.K;
    p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc; p.ldaux = d.ldaux; p.act = d.act; p.ones_row = d.ones_row;
  }
  static const int at[3] = {0, 0, 1}, bt[3] = {0, 1, 0}, ep[3] = {EPI_BIAS_ACT, EPI_DACT, EPI_STORE};
  return gemm_launch(gb, at[variant], bt[variant], ep[variant], bf16, static_cast<hipStream_t>(stream));
}
