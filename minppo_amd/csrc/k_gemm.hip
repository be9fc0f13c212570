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
// carries 160-300 workgroups.  There is NO LDS staging and NO barrier: the f32 MFMA consumes only 8 bytes
// of operands per lane per 64 cycles, so each wave streams its A and B fragments straight from L1/L2 into
// registers in MFMA layout (see "direct-to-register operand streams" below), one 32-deep k-set ahead.
// (Round 1 measured the LDS-staged version of this kernel: 40 % of its wave cycles sat in s_waitcnt /
// s_barrier around the LDS round trip — profiles/r01_b_pmc_gemm_lds.txt.)  The f32 MFMA rate
// (157 TFLOP/s) is the roofline for these kernels; at these sizes the achieved fraction is set by
// tile count / wave occupancy rather than by memory.
#include <wave_ops.h>

#include <cstdlib>
#include <cstring>

#include "gemm.h"
#include "mppo_common.h"

namespace mppo {

constexpr int BM = 64, BN = 64, KS = 32, GEMM_THREADS = 256;  // workgroup tile, k-set (16 MFMAs), 4 waves as 2 x 2

// ---- direct-to-register operand streams --------------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 takes, per lane l = (i = l&31, h = l>>5), ONE A value A[i][k_h] and ONE B value B[k_h][j=i]
// and sums the two k's of the two lane halves.  Any assignment of k's to (MFMA, h) works as long as A and B use the
// same one and every k is covered once.  We use, inside a set of 32 k's:  MFMA (g, c) of half h  <->  k = 8g + 4h + c
// (g = 0..3, c = 0..3), so that a lane's four values of a group are 4 CONSECUTIVE k's: one float4 load where memory is
// contiguous along k, four coalesced dword loads (128 B per half-wave) where it is contiguous along the row/column.
// Operands are read straight from L1/L2 into registers (8 B/clk per wave, far below the L1 rate), double-buffered one
// set ahead, so a wave never touches LDS and never meets a barrier.
struct OperandK {  // memory contiguous along k:  element (r, k) at base[row(r)*ld + k]   (row = lane's own row, clamped)
  const float* ptr;  // row(r)*ld + 4*h ; the k-set offset is added per load (pointer bump + immediates)
  bool ok, vec;
  __device__ __forceinline__ void init(const float* base, int ld, const int* gather, int r, int R, int h) {
    vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);  // problem-uniform: float4 loads are legal
    ok = r < R;
    const int rr = ok ? r : R - 1;
    const long row = gather ? gather[rr] : rr;
    ptr = base + row * (long)ld + 4 * h;
  }
  __device__ __forceinline__ void load(float (&v)[16], int k0) const {  // full set: k0 .. k0+31 all valid
    const float* q0 = ptr + k0;
    if (vec) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 q = *reinterpret_cast<const float4*>(q0 + 8 * g);
        v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
      }
    } else {  // odd action dimension / unpadded rows: same data with dword loads
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[4 * g + c] = q0[8 * g + c];
    }
  }
  __device__ __forceinline__ void load_tail(float (&v)[16], int k0, int kend, int h) const {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = k0 + 8 * g + 4 * h + c;
        v[4 * g + c] = (ok && k < kend) ? ptr[k0 + 8 * g + c] : 0.f;
      }
  }
};

struct OperandR {  // memory contiguous along r:  element (r, k) at base[k*ld + r]   (r = lane's own column, clamped)
  const float* base;  // wave-uniform
  int ld, lane_off, kend;  // lane_off = column + 4*h*ld : the only per-lane part of the address
  bool ok;
  __device__ __forceinline__ void init(const float* base_, int ld_, int r, int R, int kend_, int h) {
    ok = r < R;
    base = base_; ld = ld_; kend = kend_;
    lane_off = (ok ? r : R - 1) + 4 * h * ld_;
  }
  __device__ __forceinline__ void load(float (&v)[16], int k0) const {  // full set; uniform base + 32-bit lane offset
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float* up = base + (long)(k0 + 8 * g + c) * ld;  // scalar arithmetic
        v[4 * g + c] = up[lane_off];
      }
  }
  __device__ __forceinline__ void load_tail(float (&v)[16], int k0, int h) const {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = k0 + 8 * g + 4 * h + c;
        v[4 * g + c] = (ok && k < kend) ? base[(long)(k0 + 8 * g + c) * ld + lane_off] : 0.f;
      }
  }
};

__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); |abs error| < 2e-7 on the whole range, saturates cleanly for large |x|
  const float e = __expf(2.f * x);
  return 1.f - __fdividef(2.f, e + 1.f);
}

// One k-set (32 k's) of a wave's 32x32 tile.  f32: 16 x v_mfma_f32_32x32x2_f32.  BF16: the lane's four consecutive k's of a
// group (k = 8g + 4h + c) are exactly the operand of v_mfma_f32_32x32x8_bf16, so the same registers feed 4 MFMAs after
// rounding to bf16 (round to nearest even), f32 accumulate.
template <bool BF16>
__device__ __forceinline__ void mfma_set(const float (&a)[16], const float (&b)[16], f32x16& acc) {
  if (BF16) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      mfma_bf16_32x32x8(pack_bf16x4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]), pack_bf16x4(b[4 * g], b[4 * g + 1], b[4 * g + 2], b[4 * g + 3]), acc);
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) mfma_f32_32x32x2(a[i], b[i], acc);
  }
}

// Direct variant for the k-contiguous-A products (forward NN, backward NT).
template <bool B_T, int EPI, bool BF16>
__global__ void __launch_bounds__(GEMM_THREADS) gemm_kernel(GemmBatch gb) {
  const int z = blockIdx.z;
  const GemmProb p = gb.p[z];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, hi = lane >> 5;
  const int m0 = blockIdx.y * BM + wr * 32, n0 = blockIdx.x * BN + wc * 32;  // this wave's 32 x 32 tile
  if (m0 >= p.M || n0 >= p.N) return;                                        // wave-uniform; no barriers below
  const int K = p.K;
  OperandK ak, bk;
  OperandR br;
  ak.init(p.A, p.lda, p.gather, m0 + l31, p.M, hi);
  if (B_T) bk.init(p.B, p.ldb, nullptr, n0 + l31, p.N, hi);
  else br.init(p.B, p.ldb, n0 + l31, p.N, K, hi);
  // forward only: the waves of the first column tile also write their (gathered) A rows to a contiguous copy, so that
  // the weight-gradient product later reads the minibatch without a gather (p.a_copy: [M, lda], same row stride)
  float* acopy = (EPI == EPI_BIAS_ACT && p.a_copy && n0 == 0 && ak.ok) ? p.a_copy + (size_t)(m0 + l31) * p.lda + 4 * hi : nullptr;
  auto copy_out = [&](const float (&v)[16], int k0) {
    if (acopy) {
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(acopy + k0 + 8 * g) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
    }
  };

  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nfull = K / KS, tail = K - nfull * KS;
  float a0[16], b0[16], a1[16], b1[16];
  if (nfull > 0) {
    ak.load(a0, 0);
    if (B_T) bk.load(b0, 0); else br.load(b0, 0);
    // steady state, copy-free ping-pong.  Invariant at the loop top: (a0,b0) hold set sidx-1, loaded but not yet consumed.
    // Each half is one basic block: the loads of the next set are interleaved with the 16 MFMAs of the current one and
    // have a whole further MFMA block (>= 1024 cycles) before their first use.
    int sidx = 1;
    for (; sidx + 1 < nfull; sidx += 2) {
      ak.load(a1, sidx * KS);
      if (B_T) bk.load(b1, sidx * KS); else br.load(b1, sidx * KS);
      mfma_set<BF16>(a0, b0, acc);
      if (!BF16) { if (B_T) { MPPO_INTERLEAVE_MFMA16(1, 1) } else { MPPO_INTERLEAVE_MFMA16(2, 2) } }
      copy_out(a0, (sidx - 1) * KS);
      ak.load(a0, (sidx + 1) * KS);
      if (B_T) bk.load(b0, (sidx + 1) * KS); else br.load(b0, (sidx + 1) * KS);
      mfma_set<BF16>(a1, b1, acc);
      if (!BF16) { if (B_T) { MPPO_INTERLEAVE_MFMA16(1, 1) } else { MPPO_INTERLEAVE_MFMA16(2, 2) } }
      copy_out(a1, sidx * KS);
    }
    if (sidx < nfull) {  // one more set to fetch
      ak.load(a1, sidx * KS);
      if (B_T) bk.load(b1, sidx * KS); else br.load(b1, sidx * KS);
      mfma_set<BF16>(a0, b0, acc);
      if (!BF16) { if (B_T) { MPPO_INTERLEAVE_MFMA16(1, 1) } else { MPPO_INTERLEAVE_MFMA16(2, 2) } }
      copy_out(a0, (sidx - 1) * KS);
      mfma_set<BF16>(a1, b1, acc);
      copy_out(a1, sidx * KS);
    } else {
      mfma_set<BF16>(a0, b0, acc);
      copy_out(a0, (sidx - 1) * KS);
    }
  }
  if (tail > 0) {
    ak.load_tail(a0, nfull * KS, K, hi);
    if (B_T) bk.load_tail(b0, nfull * KS, K, hi); else br.load_tail(b0, nfull * KS, hi);
    if (BF16) {
      mfma_set<true>(a0, b0, acc);  // zeros beyond the tail
    } else {
      const int nm = (tail + 7) / 8 * 4;  // MFMAs that can carry data (groups of 8 k's)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i < nm) mfma_f32_32x32x2(a0[i], b0[i], acc);
    }
    if (acopy) {  // tail columns (zeros past K keep the copy's padding clean up to the row stride)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int k = nfull * KS + 8 * g + 4 * hi + c;
          if (k < p.lda) acopy[nfull * KS + 8 * g + c] = a0[4 * g + c];
        }
    }
  }

  const int col = n0 + l31;
  if (col < p.N) {
    float bias = 0.f;
    if (EPI == EPI_BIAS_ACT && p.bias) bias = p.bias[col];
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (row < p.M) {
        float v = acc[r];
        if (EPI == EPI_BIAS_ACT) {
          v += bias;
          if (p.act == ACT_TANH) v = fast_tanh(v);
          else if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
        } else if (EPI == EPI_DACT) {
          const float hval = p.aux[(size_t)row * p.ldaux + col];
          if (p.act == ACT_TANH) v *= (1.f - hval * hval);
          else if (p.act == ACT_RELU) v = hval > 0.f ? v : 0.f;
        }
        p.C[(size_t)row * p.ldc + col] = v;
      }
    }
  }
}

// Direct variant of the weight-gradient product C[M,N] = A^T . B with A stored [K,M] and B stored [K,N] (K = samples,
// no gather: the minibatch rows were laid out contiguously by the forward pass).  Both operands are "r-contiguous":
// 16 coalesced dword loads per operand and k-set, uniform base + 32-bit lane offset.  Split-K over blockIdx.z.
template <bool BF16>
__global__ void __launch_bounds__(GEMM_THREADS) gemm_tn_kernel(GemmBatch gb) {
  const int z = blockIdx.z;
  const int pi = z / gb.ksplit, ks = z - pi * gb.ksplit;
  const GemmProb p = gb.p[pi];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, hi = lane >> 5;
  const int m0 = blockIdx.y * BM + wr * 32, n0 = blockIdx.x * BN + wc * 32;
  if (m0 >= p.M || n0 >= p.N) return;
  int kper = (p.K + gb.ksplit - 1) / gb.ksplit;
  kper = (kper + KS - 1) / KS * KS;
  const int kb = ks * kper;
  const int ke = p.K < kb + kper ? p.K : kb + kper;
  OperandR ar, br;
  ar.init(p.A, p.lda, m0 + l31, p.M, ke, hi);
  br.init(p.B, p.ldb, n0 + l31, p.N, ke, hi);
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float colsum = 0.f;  // sum over k of B(k, n) = bias gradient; waves of the first row tile only
  const bool do_colsum = p.bias_out && m0 == 0;
  const int nfull = ke > kb ? (ke - kb) / KS : 0, tail = ke > kb ? (ke - kb) - nfull * KS : 0;
  float a0[16], b0[16], a1[16], b1[16];
  if (nfull > 0) {
    ar.load(a0, kb); br.load(b0, kb);
    int sidx = 1;
    for (; sidx + 1 < nfull; sidx += 2) {
      ar.load(a1, kb + sidx * KS); br.load(b1, kb + sidx * KS);
      mfma_set<BF16>(a0, b0, acc);
#pragma unroll
      for (int i = 0; i < 16; ++i) colsum += b0[i];
      if (!BF16) { MPPO_INTERLEAVE_MFMA16(3, 2) }
      ar.load(a0, kb + (sidx + 1) * KS); br.load(b0, kb + (sidx + 1) * KS);
      mfma_set<BF16>(a1, b1, acc);
#pragma unroll
      for (int i = 0; i < 16; ++i) colsum += b1[i];
      if (!BF16) { MPPO_INTERLEAVE_MFMA16(3, 2) }
    }
    if (sidx < nfull) {
      ar.load(a1, kb + sidx * KS); br.load(b1, kb + sidx * KS);
      mfma_set<BF16>(a0, b0, acc);
#pragma unroll
      for (int i = 0; i < 16; ++i) colsum += b0[i];
      if (!BF16) { MPPO_INTERLEAVE_MFMA16(3, 2) }
      mfma_set<BF16>(a1, b1, acc);
#pragma unroll
      for (int i = 0; i < 16; ++i) colsum += b1[i];
    } else {
      mfma_set<BF16>(a0, b0, acc);
#pragma unroll
      for (int i = 0; i < 16; ++i) colsum += b0[i];
    }
  }
  if (tail > 0) {
    ar.load_tail(a0, kb + nfull * KS, hi); br.load_tail(b0, kb + nfull * KS, hi);
    mfma_set<BF16>(a0, b0, acc);
#pragma unroll
    for (int i = 0; i < 16; ++i) colsum += b0[i];
  }
  float* C = p.C + (size_t)ks * gb.slab_stride;
  const int col = n0 + l31;
  if (col < p.N) {
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (row < p.M) C[(size_t)row * p.ldc + col] = acc[r];
    }
  }
  if (do_colsum) {
    colsum += __shfl_xor(colsum, 32);  // the two lane halves hold the two k-halves of every group
    if (hi == 0 && col < p.N) p.bias_out[(size_t)ks * gb.slab_stride + col] = colsum;
  }
}

template <bool B_T, int EPI, bool BF16>
static int32_t launch_t(const GemmBatch& gb, hipStream_t stream) {
  int maxM = 0, maxN = 0;
  for (int i = 0; i < gb.count; ++i) {
    maxM = gb.p[i].M > maxM ? gb.p[i].M : maxM;
    maxN = gb.p[i].N > maxN ? gb.p[i].N : maxN;
  }
  dim3 grid(cdiv(maxN, BN), cdiv(maxM, BM), gb.count);
  hipLaunchKernelGGL((gemm_kernel<B_T, EPI, BF16>), grid, dim3(GEMM_THREADS), 0, stream, gb);
  MPPO_CHECK_LAUNCH("gemm_kernel");
  return MPPO_OK;
}

int32_t gemm_launch(const GemmBatch& gb, int a_t, int b_t, int epi, int bf16, hipStream_t stream) {
  MPPO_REQUIRE(gb.count >= 1 && gb.count <= kGemmMaxProb, "gemm_launch: %d problems", gb.count);
  MPPO_REQUIRE(gb.ksplit >= 1 && (gb.ksplit == 1 || epi == EPI_STORE), "gemm_launch: split-K only with EPI_STORE");
  for (int i = 0; i < gb.count; ++i) {
    const GemmProb& p = gb.p[i];
    MPPO_REQUIRE(p.A && p.B && p.C && p.M >= 1 && p.N >= 1 && p.K >= 1, "gemm_launch: problem %d malformed (M=%d N=%d K=%d)", i, p.M, p.N, p.K);
    MPPO_REQUIRE(epi != EPI_DACT || p.aux, "gemm_launch: EPI_DACT needs aux");
    MPPO_REQUIRE(!p.bias_out || epi == EPI_STORE, "gemm_launch: bias_out only with EPI_STORE");
  }
  const int v = a_t * 2 + b_t;
  // implementation per variant: direct-to-register everywhere (gathered weight-gradient operands fall back to LDS).
  // MPPO_GEMM_IMPL = "ddd" / "lll" / ... overrides per variant (d = direct, l = LDS) for A/B measurements.
  static const char* impl = MPPO_EXPERIMENT_ENV("MPPO_GEMM_IMPL");
  const char choice = (impl && (int)strlen(impl) > v) ? impl[v] : 'd';
  // bf16-in / f32-accumulate exists for the direct forward and weight-gradient kernels only (the fused row pass covers the rest)
  if (choice == 'l') { MPPO_REQUIRE(!bf16, "gemm_launch: no bf16 variant of the LDS-staged kernels"); return gemm_launch_lds(gb, a_t, b_t, epi, stream); }
  if (epi == EPI_BIAS_ACT && v == 0) return bf16 ? launch_t<false, EPI_BIAS_ACT, true>(gb, stream) : launch_t<false, EPI_BIAS_ACT, false>(gb, stream);
  if (epi == EPI_DACT && v == 1) { MPPO_REQUIRE(!bf16, "gemm_launch: no bf16 variant of the stand-alone backward product"); return launch_t<true, EPI_DACT, false>(gb, stream); }
  if (epi == EPI_STORE && v == 2) {
    bool gathered = false;
    for (int i = 0; i < gb.count; ++i) gathered = gathered || gb.p[i].gather;
    if (gathered) { MPPO_REQUIRE(!bf16, "gemm_launch: no bf16 variant of the gathered weight-gradient product"); return gemm_launch_lds(gb, a_t, b_t, epi, stream); }  // rows by index: LDS-staged kernel
    int maxM = 0, maxN = 0;
    for (int i = 0; i < gb.count; ++i) { maxM = gb.p[i].M > maxM ? gb.p[i].M : maxM; maxN = gb.p[i].N > maxN ? gb.p[i].N : maxN; }
    const dim3 grid(cdiv(maxN, BN), cdiv(maxM, BM), gb.count * gb.ksplit);
    if (bf16) hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(GEMM_THREADS), 0, stream, gb);
    else hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, dim3(GEMM_THREADS), 0, stream, gb);
    MPPO_CHECK_LAUNCH("gemm_tn_kernel");
    return MPPO_OK;
  }
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
    p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc; p.ldaux = d.ldaux; p.act = d.act; p.bias_out = d.bias_out;
  }
  static const int at[3] = {0, 0, 1}, bt[3] = {0, 1, 0}, ep[3] = {EPI_BIAS_ACT, EPI_DACT, EPI_STORE};
  return gemm_launch(gb, at[variant], bt[variant], ep[variant], bf16, static_cast<hipStream_t>(stream));
}
