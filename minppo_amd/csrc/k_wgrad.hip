// k_wgrad.hip — all weight and bias gradients of one minibatch as split-K partial sums, one launch.
//
// What `jax.value_and_grad(_loss_fn)` (reference minppo/train.py:246-247) derives for the six Dense layers:
//   dW = H_prev^T . dZ   (contraction over the minibatch rows, K = mb)      db = column sums of dZ
// for actor and critic.  The K chunks are summed by grad_reduce_kernel (k_ppo.hip), which follows.
//
// Shape regime: six problems C[M,N] = A^T.B with A stored [K, M] (h1 / h2 / the gathered observations), B stored [K, N]
// (dZ1 / dZ2 / dOut), M, N <= 256, K = 1280 .. 2560.  Too little work per output tile to fill 256 CUs without splitting K,
// so a workgroup = one 64x64 output tile x one K chunk (K / ksplit rows):
//   * operands are staged through LDS in their natural [k][m] layout, one 32-row k-set per stage, with range-checked
//     buffer_load_dwordx4 (4 per thread and stage; round 1's direct-to-register kernel issued 32 dword loads per stage and
//     fetched every operand element twice per workgroup).  The first five stages (= the whole K chunk at mb = 1280) are
//     requested before anything else: the activations were written with streaming stores by the row pass, so the first
//     touch comes from memory, and there is ONE such latency per workgroup instead of one per stage;
//   * 4 waves, each a 32x32 accumulator on v_mfma_f32_32x32x2_f32 (bf16: 32x32x8), operands read from LDS rows
//     (lanes = consecutive columns: conflict-free, ds_read2st64_b32);
//   * K chunk <-> XCD: workgroup ids that are equal modulo 8 run on the same XCD (round-robin dispatch), and with ksplit = 8
//     chunk c of EVERY tile is given to XCD c, so each operand row is fetched from memory by exactly one L2 and the other
//     tiles' re-reads hit that L2 (a performance mapping only; nothing depends on it for correctness).
// Measured and rejected in round 2: summing the chunks of a tile inside this launch by the tile's last workgroup
// to arrive (agent-scope write-through partials + arrival counter).  Correct (600-launch bitwise stress test), but the one
// workgroup per tile that pulls 8 x 16 KB back through the memory side needs 7.4 us - more than the launch of the separate
// reduce kernel (5.0 us), whose 256 workgroups share that work.
#include <wave_ops.h>

#include <cstdlib>

#include "mppo_common.h"
#include "ppo_layout.h"
#include "wgrad.h"

namespace mppo {

constexpr int WT = 64;        // output tile edge
constexpr int WKS = 32;       // k rows per stage
constexpr int WTHREADS = 256;

constexpr int WRING = 5;  // stages in flight (registers); the K chunk is a whole number of rings (wgrad_plan)

template <bool BF16>
__global__ void __launch_bounds__(WTHREADS) wgrad_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[2][2][WKS][WT];  // [operand][buffer][k][column]: 32 KB
  float (*sA)[WKS][WT] = smem[0];
  float (*sB)[WKS][WT] = smem[1];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // workgroup -> (tile, K chunk): chunk = id % ksplit; with ksplit = 8 that is the XCD the workgroup runs on
  const int wg = blockIdx.x;
  const int tile = wg / a.ksplit, split = wg % a.ksplit;
  if (tile >= a.ntiles) return;
  int pi = 0;
  while (pi + 1 < a.count && tile >= a.p[pi + 1].tile0) ++pi;
  const WgradProb p = a.p[pi];
  const int lt = tile - p.tile0, mt = lt / p.tiles_n, nt = lt - mt * p.tiles_n;
  const int m0 = mt * WT, n0 = nt * WT;
  const int kb = split * a.kchunk, ke = a.K < kb + a.kchunk ? a.K : kb + a.kchunk;
  const int nst = a.kchunk / WKS;  // a multiple of WRING; stages past `ke` read zeros (range-checked buffer loads)

  // staging: thread t moves float4 (rows t/16 and t/16 + 16 of the stage, columns 4*(t%16) ..) of both operands.
  // Range-checked buffer loads: rows at or past `ke` return 0 without a branch; a column overhang (tile wider than the
  // operand's readable row) gets a lane offset past the buffer's extent, which reads as 0 too.  WRING stages are in flight at any time:
  // the activations were written with streaming stores by the row pass, so the first touch comes from memory (~2 us).
  const int lr = t >> 4, lc = (t & 15) * 4;
  const int ext = ke > kb ? ke - kb : 0;
  const BufView bufA = make_buf(p.A + (size_t)kb * p.lda, (unsigned)ext * (unsigned)p.lda * 4u);
  const BufView bufB = make_buf(p.B + (size_t)kb * p.ldb, (unsigned)ext * (unsigned)p.ldb * 4u);
  const int ca = m0 + lc, cb = n0 + lc;
  constexpr int kOutOfRange = 0x40000000;  // lane offset past any buffer extent: the range check returns 0, no mask arithmetic
  const int offa = ca < p.acols ? (lr * p.lda + ca) * 4 : kOutOfRange, offb = cb < p.bcols ? (lr * p.ldb + cb) * 4 : kOutOfRange;
  const int last = nst - 1;
  float4 ra[WRING][2], rb[WRING][2];
  if (a.dbg & 8) for (int i = 0; i < WRING; ++i) for (int h = 0; h < 2; ++h) { ra[i][h] = make_float4(0.f, 0.f, 0.f, 0.f); rb[i][h] = ra[i][h]; }
#define WG_LOAD(SLOT, STAGE)                                                          \
  do {                                                                                \
    const int _st = (STAGE) < last ? (STAGE) : last;                                  \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                   \
      if (!(a.dbg & 8)) {                                                           \
      ra[SLOT][h] = buf_load_f4(bufA, offa, (_st * WKS + 16 * h) * p.lda * 4);        \
      rb[SLOT][h] = buf_load_f4(bufB, offb, (_st * WKS + 16 * h) * p.ldb * 4); }      \
    }                                                                                 \
  } while (0)
#define WG_LSTORE(SLOT, BUF)                                                                                                      \
  do {                                                                                                                            \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                                               \
      *reinterpret_cast<float4*>(&sA[BUF][lr + 16 * h][lc]) = ra[SLOT][h];                                                        \
      *reinterpret_cast<float4*>(&sB[BUF][lr + 16 * h][lc]) = rb[SLOT][h];                                                        \
    }                                                                                                                             \
  } while (0)
  const int wr = wave >> 1, wc = wave & 1, i31 = lane & 31, hi = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float colsum = 0.f;  // bias gradient: sum over k of B(k, n); waves of the first row band of the first row tile only
  const bool do_colsum = p.off_b >= 0 && mt == 0 && wr == 0;
  // a wave whose 32x32 block lies entirely outside the problem (the head layers' second column block, the observation
  // layer's last row block) leaves the matrix pipe to the other waves of the SIMD; it still stages operands
#ifdef MPPO_EMU
  const bool live = true;  // the emulator's MFMA shim synchronises the whole workgroup
#else
  const bool live = m0 + wr * 32 < p.M && n0 + wc * 32 < p.N;
#endif
  auto compute = [&](int buf) {
    if ((a.dbg & 1) || !live) return;
    const float* pa = &sA[buf][0][wr * 32 + i31];
    const float* pb = &sB[buf][0][wc * 32 + i31];
    if (BF16) {
      // v_mfma_f32_32x32x8_bf16: lane (i, h) supplies k = 8g + 4h + c, c = 0..3
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float av[4], bv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { av[c] = pa[(8 * g + 4 * hi + c) * WT]; bv[c] = pb[(8 * g + 4 * hi + c) * WT]; }
        mfma_bf16_32x32x8(pack_bf16x4(av[0], av[1], av[2], av[3]), pack_bf16x4(bv[0], bv[1], bv[2], bv[3]), acc);
        colsum += (bv[0] + bv[1]) + (bv[2] + bv[3]);
      }
    } else {
      // v_mfma_f32_32x32x2_f32: lane (i, h) supplies k = 2j + h of MFMA j
      float av[16], bv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) { av[j] = pa[(2 * j + hi) * WT]; bv[j] = pb[(2 * j + hi) * WT]; }
#pragma unroll
      for (int j = 0; j < 16; ++j) mfma_f32_32x32x2(av[j], bv[j], acc);
#pragma unroll
      for (int j = 0; j < 16; ++j) colsum += bv[j];
    }
  };
  // prologue: the whole first ring is requested at once
  WG_LOAD(0, 0); WG_LOAD(1, 1); WG_LOAD(2, 2); WG_LOAD(3, 3); WG_LOAD(4, 4);
  WG_LSTORE(0, 0);
  if (nst > WRING) WG_LOAD(0, 5);
  __syncthreads();
  int par = 0;  // LDS buffer of the ring's first stage (a ring has an odd number of stages: the parity flips every round)
  for (int s = 0; s < nst; s += WRING) {
    const bool more = s + 2 * WRING <= nst;  // stage s + u + 6 exists: a further ring follows (uniform; stage u + 1's slot is free once staged)
    compute(par);     WG_LSTORE(1, par ^ 1); if (more) WG_LOAD(1, s + 6);  __syncthreads();
    compute(par ^ 1); WG_LSTORE(2, par);     if (more) WG_LOAD(2, s + 7);  __syncthreads();
    compute(par);     WG_LSTORE(3, par ^ 1); if (more) WG_LOAD(3, s + 8);  __syncthreads();
    compute(par ^ 1); WG_LSTORE(4, par);     if (more) WG_LOAD(4, s + 9);  __syncthreads();
    compute(par);     WG_LSTORE(0, par ^ 1); if (s + 2 * WRING < nst) WG_LOAD(0, s + 10); __syncthreads();
    par ^= 1;
  }
#undef WG_LOAD
#undef WG_LSTORE

  if (a.dbg & 2) return;
  // ---- partial tile -> slab `split` through LDS (the operand buffers are dead): the MFMA accumulator layout holds a
  // column per lane, i.e. 128-byte row fragments; staged as a [64][68] tile, every store instruction of a wave writes four
  // complete 256-byte rows.  The bias gradient's partial rides along. ----
  float* slab = a.slabs + (size_t)split * a.slab_stride;
  float* ct = &smem[0][0][0][0];  // 64 x 68 floats = 17 KB of the 32 KB
  constexpr int CTS = WT + 4;
  static_assert(sizeof(smem) >= WT * CTS * sizeof(float), "the output tile is staged over the operand buffers");
#pragma unroll
  for (int r = 0; r < 16; ++r) ct[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * CTS + wc * 32 + i31] = acc[r];
  colsum += __shfl_xor(colsum, 32);  // the two lane halves hold the odd / even k's
  const int col = n0 + wc * 32 + i31;
  if (do_colsum && hi == 0 && col < p.N) slab[p.off_b + col] = colsum;
  __syncthreads();
  const int tc = n0 + lc;
  const bool vec = (p.N & 3) == 0 && (p.off_w & 3) == 0;  // rows of this gradient tensor are 16-byte aligned
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rl = lr + 16 * j, row = m0 + rl;
    if (row < p.M && tc < p.N) {
      const float4 v = *reinterpret_cast<const float4*>(ct + rl * CTS + lc);
      float* dst = slab + p.off_w + (size_t)row * p.N + tc;
      if (vec) {
        *reinterpret_cast<float4*>(dst) = v;
      } else {  // the head layers (N = A, N = 1)
        dst[0] = v.x;
        if (tc + 1 < p.N) dst[1] = v.y;
        if (tc + 2 < p.N) dst[2] = v.z;
        if (tc + 3 < p.N) dst[3] = v.w;
      }
    }
  }
}

bool wgrad_supported(const WgradArgs& a) {
  if (a.count < 1 || a.count > kWgradMaxProb) return false;
  for (int i = 0; i < a.count; ++i) {
    const WgradProb& p = a.p[i];
    if ((p.lda & 3) || (p.ldb & 3) || (p.acols & 3) || (p.bcols & 3) || p.acols < p.M || p.bcols < p.N || p.acols > p.lda || p.bcols > p.ldb || (reinterpret_cast<uintptr_t>(p.A) & 15) || (reinterpret_cast<uintptr_t>(p.B) & 15)) return false;
  }
  return true;
}

int32_t wgrad_plan(WgradArgs& a, int K) {
  int tiles = 0;
  for (int i = 0; i < a.count; ++i) {
    WgradProb& p = a.p[i];
    p.tiles_m = cdiv(p.M, WT); p.tiles_n = cdiv(p.N, WT); p.tile0 = tiles;
    tiles += p.tiles_m * p.tiles_n;
  }
  a.ntiles = tiles;
  a.K = K;
  const int ring = WKS * WRING;  // the kernel streams whole rings of stages
  a.kchunk = cdiv(cdiv(K, a.ksplit), ring) * ring;
  return MPPO_OK;
}

int32_t wgrad_launch(const WgradArgs& a_in, bool bf16, hipStream_t stream) {
  WgradArgs a = a_in;
  static const int dbg = [] { const char* e = getenv("MPPO_WGRAD_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
  MPPO_REQUIRE(wgrad_supported(a), "wgrad_launch: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  MPPO_REQUIRE(a.ksplit >= 1 && a.ksplit <= kGradKSplitMax && a.kchunk % (WKS * WRING) == 0, "wgrad_launch: ksplit %d / kchunk %d", a.ksplit, a.kchunk);
  const dim3 grid(a.ntiles * a.ksplit);
  if (bf16) hipLaunchKernelGGL(wgrad_kernel<true>, grid, dim3(WTHREADS), 0, stream, a);
  else hipLaunchKernelGGL(wgrad_kernel<false>, grid, dim3(WTHREADS), 0, stream, a);
  MPPO_CHECK_LAUNCH("wgrad_kernel");
  return MPPO_OK;
}

}  // namespace mppo
