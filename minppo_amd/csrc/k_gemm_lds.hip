// k_gemm_lds.hip — the LDS-staged variant of the batched small GEMM (64x64x32 tiles, global -> registers -> LDS one k-tile
// ahead, one barrier per k-tile).  Kept beside the direct-to-register variant of k_gemm.hip: for the weight-gradient
// products (both operands sample-major, long K) coalesced float4 row loads + an LDS transpose beat 32 strided dword
// loads per k-set; gemm_launch (k_gemm.hip) picks the implementation per variant.
#include <wave_ops.h>

#include "gemm.h"
#include "mppo_common.h"

namespace mppo {
namespace lds {

constexpr int BM = 64, BN = 64, BK = 32, LDT = 68, GEMM_THREADS = 256;
constexpr int TILE_F = BK * LDT;  // floats of one staged operand tile

// ---- global -> register staging (issued one k-tile ahead), register -> LDS (k-major image T[k][r]) ----
// Two loaders per memory orientation: `fast` is branch-free (16-byte aligned rows, unconditional float4 loads, row /
// column indices clamped into the allocation so that out-of-tile lanes re-read valid data which the epilogue never
// stores); `slow` predicates every element and serves ragged shapes and the partial last k-tile.

// "kc": memory contiguous along k, element (r,k) at base[row(r)*ld + k].  A thread owns row r = t>>2 and the two
// k-quads kq = 4*(t&3) and kq+16 (so that the transposed LDS stores of a wave hit each bank at most twice).
struct StageKC {
  float v[8];
  const float* ptr;  // row(r) base + kq, fixed for the whole K loop
  bool row_ok;
  __device__ __forceinline__ void init(const float* base, int ld, const int* gather, int r0, int R, int t) {
    const int r = t >> 2, kq = (t & 3) * 4;
    int gr = r0 + r;
    row_ok = gr < R;
    gr = row_ok ? gr : R - 1;
    const long row = gather ? gather[gr] : gr;
    ptr = base + row * (long)ld + kq;
  }
  __device__ __forceinline__ void load_fast(int k0) {
    const float4 a = *reinterpret_cast<const float4*>(ptr + k0);
    const float4 b = *reinterpret_cast<const float4*>(ptr + k0 + 16);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  __device__ __forceinline__ void load_slow(int k0, int kend, int t) {
    const int kq = (t & 3) * 4;
    for (int hq = 0; hq < 2; ++hq)
      for (int c = 0; c < 4; ++c) {
        const int kk = k0 + kq + 16 * hq + c;
        v[4 * hq + c] = (row_ok && kk < kend) ? ptr[k0 + 16 * hq + c] : 0.f;
      }
  }
  __device__ __forceinline__ void store(float* T, int t) const {
    const int r = t >> 2, kq = (t & 3) * 4;
    for (int hq = 0; hq < 2; ++hq)
      for (int c = 0; c < 4; ++c) T[(kq + 16 * hq + c) * LDT + r] = v[4 * hq + c];
  }
};

// "rc": memory contiguous along r, element (r,k) at base[row(k)*ld + r].  A thread owns the r-quad rq = 4*(t&15) of the
// two k-rows k = t>>4 and k+16.  With a gather the row indices of the NEXT tile are fetched one tile early.
struct StageRC {
  float4 q[2];
  const float* colptr;  // base + clamped column
  const int* gather;
  int ld, col, Rmem, kend;
  int nxt[2];
  __device__ __forceinline__ void init(const float* base, int ld_, const int* gather_, int r0, int Rmem_, int k0, int kend_, int t, bool fast) {
    ld = ld_; gather = gather_; Rmem = Rmem_; kend = kend_;
    col = r0 + (t & 15) * 4;
    const int cmax = (ld & ~3) - 4;
    colptr = base + (fast ? (col < cmax ? col : cmax) : col);
    if (gather) prefetch_idx(k0, t);
  }
  __device__ __forceinline__ void prefetch_idx(int k0, int t) {
    for (int hq = 0; hq < 2; ++hq) {
      int gk = k0 + (t >> 4) + 16 * hq;
      gk = gk < kend ? gk : kend - 1;
      nxt[hq] = gather[gk];
    }
  }
  __device__ __forceinline__ void load_fast(int k0, int t) {  // every k-row of the tile is < kend
    for (int hq = 0; hq < 2; ++hq) {
      const long row = gather ? nxt[hq] : k0 + (t >> 4) + 16 * hq;
      q[hq] = *reinterpret_cast<const float4*>(colptr + row * (long)ld);
    }
    if (gather) prefetch_idx(k0 + BK, t);
  }
  __device__ __forceinline__ void load_slow(int k0, int t) {
    for (int hq = 0; hq < 2; ++hq) {
      const int gk = k0 + (t >> 4) + 16 * hq;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (gk < kend) {
        const long row = gather ? gather[gk] : gk;
        const float* ptr = colptr + row * (long)ld;  // colptr == base + col on the slow path
        for (int c = 0; c < 4; ++c) if (col + c < Rmem) v[c] = ptr[c];
      }
      q[hq] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  __device__ __forceinline__ void store(float* T, int t) const {
    const int k = t >> 4, rq = (t & 15) * 4;
    *reinterpret_cast<float4*>(T + k * LDT + rq) = q[0];
    *reinterpret_cast<float4*>(T + (k + 16) * LDT + rq) = q[1];
  }
};

__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); |abs error| < 2e-7 on the whole range, saturates cleanly for large |x|
  const float e = __expf(2.f * x);
  return 1.f - __fdividef(2.f, e + 1.f);
}

// Pipeline: the global loads of k-tile i+1 are in flight while the 16 MFMAs per wave of k-tile i run; the MFMA operands
// of a k-tile are all fetched from LDS into registers before the MFMA chain starts; one workgroup barrier per k-tile.
template <bool A_T, bool B_T, int EPI, bool FAST>
__global__ void __launch_bounds__(GEMM_THREADS) gemm_kernel(GemmBatch gb) {
  __shared__ __attribute__((aligned(16))) float As[2 * TILE_F];
  __shared__ __attribute__((aligned(16))) float Bs[2 * TILE_F];
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

  StageKC a_kc, b_kc;
  StageRC a_rc, b_rc;
  if (A_T) a_rc.init(p.A, p.lda, p.gather, m0, p.M, kb, ke, t, FAST);
  else a_kc.init(p.A, p.lda, p.gather, m0, p.M, t);
  if (B_T) b_kc.init(p.B, p.ldb, nullptr, n0, p.N, t);
  else b_rc.init(p.B, p.ldb, nullptr, n0, p.N, kb, ke, t, FAST);
  auto load_tiles = [&](int k0) {
    const bool full = FAST && (k0 + BK <= ke);  // workgroup-uniform
    if (full) {
      if (A_T) a_rc.load_fast(k0, t); else a_kc.load_fast(k0);
      if (B_T) b_kc.load_fast(k0); else b_rc.load_fast(k0, t);
    } else {
      if (A_T) a_rc.load_slow(k0, t); else a_kc.load_slow(k0, ke, t);
      if (B_T) b_kc.load_slow(k0, ke, t); else b_rc.load_slow(k0, t);
    }
  };
  auto store_tiles = [&](int b) {
    if (A_T) a_rc.store(As + b * TILE_F, t); else a_kc.store(As + b * TILE_F, t);
    if (B_T) b_kc.store(Bs + b * TILE_F, t); else b_rc.store(Bs + b * TILE_F, t);
  };

  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float colsum = 0.f;  // EPI_STORE with bias_out: column sums of the B tile (= bias gradient), m-tile 0 only
  const bool do_colsum = (EPI == EPI_STORE) && p.bias_out && blockIdx.y == 0 && t < BN;

  int buf = 0;
  if (kb < ke) {
    load_tiles(kb);
    store_tiles(0);
  }
  __syncthreads();
  for (int k0 = kb; k0 < ke; k0 += BK) {
    const bool more = k0 + BK < ke;
    if (more) load_tiles(k0 + BK);
    const float* Ab = As + buf * TILE_F + wr * 32 + l31 + hi * LDT;
    const float* Bb = Bs + buf * TILE_F + wc * 32 + l31 + hi * LDT;
    float av[BK / 2], bv[BK / 2];
#pragma unroll
    for (int i = 0; i < BK / 2; ++i) { av[i] = Ab[2 * i * LDT]; bv[i] = Bb[2 * i * LDT]; }
    const int nk2 = ((ke - k0 < BK ? ke - k0 : BK) + 1) >> 1;  // MFMAs that carry data (k-rows past ke are zeros)
    if (nk2 == BK / 2) {
#pragma unroll
      for (int i = 0; i < BK / 2; ++i) mfma_f32_32x32x2(av[i], bv[i], acc);
    } else {
#pragma unroll
      for (int i = 0; i < BK / 2; ++i)
        if (i < nk2) mfma_f32_32x32x2(av[i], bv[i], acc);  // partial last k-tile only
    }
    if (do_colsum) {
      const float* Bc = Bs + buf * TILE_F + t;
      float cs = 0.f;
#pragma unroll
      for (int kk = 0; kk < BK; ++kk) cs += Bc[kk * LDT];  // rows past ke were staged as zeros
      colsum += cs;
    }
    if (more) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
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
          if (p.act == ACT_TANH) v = fast_tanh(v);
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
  if (do_colsum && n0 + t < p.N) p.bias_out[(size_t)ks * gb.slab_stride + n0 + t] = colsum;
}

template <bool A_T, bool B_T, int EPI>
static int32_t launch_lds_t(const GemmBatch& gb, hipStream_t stream) {
  int maxM = 0, maxN = 0;
  bool fast = true;  // every problem has 16-byte aligned operands with row strides that are multiples of 4 floats
  for (int i = 0; i < gb.count; ++i) {
    const GemmProb& p = gb.p[i];
    maxM = p.M > maxM ? p.M : maxM;
    maxN = p.N > maxN ? p.N : maxN;
    const bool al = ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B)) & 15) == 0 && (p.lda & 3) == 0 && (p.ldb & 3) == 0 && p.lda >= 4 &&
                    p.ldb >= 4;
    fast = fast && al;
  }
  dim3 grid(cdiv(maxN, BN), cdiv(maxM, BM), gb.count * gb.ksplit);
  if (fast) hipLaunchKernelGGL((gemm_kernel<A_T, B_T, EPI, true>), grid, dim3(GEMM_THREADS), 0, stream, gb);
  else hipLaunchKernelGGL((gemm_kernel<A_T, B_T, EPI, false>), grid, dim3(GEMM_THREADS), 0, stream, gb);
  MPPO_CHECK_LAUNCH("gemm_kernel");
  return MPPO_OK;
}


}  // namespace lds

int32_t gemm_launch_lds(const GemmBatch& gb, int a_t, int b_t, int epi, hipStream_t stream) {
  const int v = a_t * 2 + b_t;
  if (epi == EPI_BIAS_ACT && v == 0) return lds::launch_lds_t<false, false, EPI_BIAS_ACT>(gb, stream);
  if (epi == EPI_DACT && v == 1) return lds::launch_lds_t<false, true, EPI_DACT>(gb, stream);
  if (epi == EPI_STORE && v == 2) return lds::launch_lds_t<true, false, EPI_STORE>(gb, stream);
  return fail(MPPO_EINVAL, "gemm_launch_lds: variant a_t=%d b_t=%d epi=%d is not instantiated", a_t, b_t, epi);
}

}  // namespace mppo
