// k_fused.hip — one launch for everything of a minibatch step that is local to a row:
//   hidden layer 1 -> hidden layer 2 -> output layer -> clipped-PPO loss terms -> d(loss)/d(outputs) -> dZ2 -> dZ1
// (reference minppo/train.py:222-243 forward + the row-local half of the backward pass that train.py:246 derives).
//
// One workgroup = 16 minibatch rows (gathered by permutation index) of ONE network (blockIdx.y: 0 actor, 1 critic);
// H/32 waves, wave w owns hidden columns [32w, 32w+32) as two 16x16 MFMA tiles (v_mfma_f32_16x16x4_f32).  The 16-row
// activation tiles (x, h1, h2, dZ2) live in LDS (row stride +4 floats: conflict-free ds_read_b128 of 4 consecutive k);
// weights stream from L2.  At mb = 1280 that is 160 workgroups of 8 waves: two waves per SIMD, whose dependent MFMA
// chains interleave on the matrix pipe.  Replaces four launches (two GEMMs, the head/loss kernel, one backward GEMM) and
// their ~4 us fixed cost each; what needs a reduction over rows (weight gradients) stays in the split-K GEMM that follows.
// Outputs to HBM/L2: h1, h2, dZ2, dZ1, dOut and the gathered rows (xmb) - the operands of the weight-gradient kernel - in
// K-QUAD layout [rows/4][cols][4] (ppo_layout.h: the four consecutive minibatch rows of a column are one float4, which is both
// what a lane of this kernel holds in its accumulators and what a lane of k_wgrad.hip feeds to four MFMAs); loss partials.
// Template arguments: BF16 (bf16-in / f32-accumulate MFMA, BASELINE configs[3]); ROLLOUT (forward + pi.sample + log_prob + value
// on N rows, reference train.py:157-160,182: stops after the heads, writes no activations); OT (16-wide output tiles: A <= 16 / 32).
// Weight stream: range-checked buffer loads through a branch-free three-deep register ring (see BStage / GemmPipe); the
// output-layer weights live in registers.  LDS per workgroup at O = 225, H = 256: 52 KB.
#include <wave_ops.h>

#include <cstdlib>

#include "mppo_common.h"
#include "ppo_layout.h"
#include "wgrad.h"

namespace mppo {

// Phase timers (profiling builds only, -DMPPO_FUSED_TIMERS: tools/fused_phases.py): wave 0 of the first training workgroup stamps
// s_memtime at the phase boundaries into a device array that mppo_debug_fused_timers() copies out.
#ifdef MPPO_FUSED_TIMERS
__device__ unsigned long long g_fused_t[24 + 64];
#define FT(k) do { if (!ROLLOUT && blockIdx.x == 40 && blockIdx.y == 0 && threadIdx.x == 0) g_fused_t[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define FTW(k) do { if (!ROLLOUT && blockIdx.x == 40 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) g_fused_t[(k) + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FT(k) do { } while (0)
#define FTW(k) do { } while (0)
#endif

constexpr float kLog2PiF = 1.8378770664093453f;
constexpr int FRT = 16;  // rows per workgroup
#ifndef MPPO_ROLLOUT_WAVES
#define MPPO_ROLLOUT_WAVES 4
#endif
#ifndef MPPO_TRAIN_WAVES
#define MPPO_TRAIN_WAVES 2
#endif

__device__ __forceinline__ float fused_tanh(float x) {
  const float e = __expf(2.f * x);
  return 1.f - __fdividef(2.f, e + 1.f);
}

struct FusedArgs {
  int mb, O, OP, A, AP, DP, H, use_tanh;
  const float* params;
  ParamLayout L;
  mppo_batch_t b;
  const int* idx;
  const float* adv_stat;
  float inv_count;
  mppo_loss_cfg_t lc;
  float *h1[2], *h2[2], *dz2[2], *dz1[2];
  float *dout, *xmb, *partial;
  const float* w2t[2];  // W2^T shadow copies (W2T = true instantiations)
  const unsigned short* frag[2];  // bf16 fragment-order copies per network: W1 (KP x H) | W2 | W2^T (BF16 && W2T instantiations)
  // rollout mode (ROLLOUT = true): forward + pi.sample + pi.log_prob (train.py:157-160); rows are not gathered
  const float* noise;
  float *action, *log_prob, *value, *mean_out;
  int net0;  // first network of the launch: 0 = actor + critic, 1 = critic only (bootstrap value, train.py:182)
  // engine minibatch loop (PRE = true instantiations, ppo_layout.h XPre): this step's rows, already gathered, in k-quad layout; the
  // workgroups with blockIdx.y >= 2 gather the NEXT step's rows (idx_next) into xnext while the others compute.  They are
  // the LAST rows of the grid: workgroups are dispatched in linear order, so the gather fills the CUs the row tiles leave idle
  // (96 of 256 at the headline shape) and never delays a row tile (it did, by 2 us, as columns of the grid at 8192 environments)
  const float* xpre;
  float* xnext;
  const int* idx_next;
  int skip;  // timing experiments only (MPPO_FUSED_SKIP bit mask): 1 L1, 2 L2, 4 heads, 8 dZ2, 16 dZ1, 32 activation stores, 64 gather
};

// One 16-row x 32-column slab of A_tile . op(W) on the 16x16x4 f32 MFMA (A_tile in LDS with row stride AS, K a multiple
// of 32, zero beyond Kvalid).  The wave's 32 columns are two INTERLEAVED 16-column tiles: tile tau holds columns
// n0 + 2j + tau (j = lane&15), so that one float2 load feeds both tiles.
//   NT = false: W stored [K][H] (forward):   lane (j, kq) loads W[k][n0+2j .. +1] for its 8 k's of the stage
//   NT = true : W stored [H][K] (backward):  B(k,n) = W[n*K + k], lane loads float4 W[n][k..k+3] for both of its columns
// Stage = 32 k = 16 MFMAs; stage S+1's B fragment is fetched while stage S computes (copy-free ping-pong), so with two
// waves per SIMD a load has about 1000 cycles before its first use.  k inside a stage: 32S + 16g + 4kq + c.
// Loads are BRANCH-FREE on purpose: a conditional around a prefetch splits the loop body into basic blocks, and the
// compiler's s_waitcnt insertion then has to assume the shorter of the two histories at the join - it waited for vmcnt(0)
// before a stage whose operands had been requested two stages earlier, i.e. the ring degenerated to depth one.
template <bool NT>
struct BStage {
  float x0[8], x1[8];
  // NT = false: `wb` is a range-checked view of W[0 .. Kvalid) x H: rows of the zero-padded K tail read as 0
  __device__ __forceinline__ void load(const float* W, const BufView& wb, int H, int K, int S, int n0, int j, int kq) {
    if (NT) {
      // buffer loads like the forward pipe: the per-lane part of the address is fixed for the whole GEMM, the stage advances a
      // scalar offset (flat loads recomputed a 64-bit address per load: VALU work in the MFMA stream)
      const int lane_off = ((n0 + 2 * j) * K + 4 * kq) * 4;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#ifdef MPPO_FUSED_NT_FLAT
        const float4 q0 = *reinterpret_cast<const float4*>(W + (size_t)(n0 + 2 * j) * K + 32 * S + 16 * g + 4 * kq);
        const float4 q1 = *reinterpret_cast<const float4*>(W + (size_t)(n0 + 2 * j + 1) * K + 32 * S + 16 * g + 4 * kq);
#else
        const float4 q0 = buf_load_f4(wb, lane_off, (32 * S + 16 * g) * 4);
        const float4 q1 = buf_load_f4(wb, lane_off + K * 4, (32 * S + 16 * g) * 4);
#endif
        x0[4 * g] = q0.x; x0[4 * g + 1] = q0.y; x0[4 * g + 2] = q0.z; x0[4 * g + 3] = q0.w;
        x1[4 * g] = q1.x; x1[4 * g + 1] = q1.y; x1[4 * g + 2] = q1.z; x1[4 * g + 3] = q1.w;
      }
    } else {
      const int lane_off = (4 * kq * H + n0 + 2 * j) * 4;  // bytes; the only per-lane part of the address
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float2 q = buf_load_f2(wb, lane_off, (32 * S + 16 * g + c) * H * 4);
          x0[4 * g + c] = q.x; x1[4 * g + c] = q.y;
        }
    }
  }
};

// BF16: a lane's four consecutive k's (the float4 it already holds) are one operand of v_mfma_f32_16x16x16_bf16: the same
// registers, rounded to bf16, feed 4 MFMAs instead of 16 (f32 accumulate).
template <bool BF16>
__device__ __forceinline__ void stage_mfma(const float* arow, int S, const float (&x0)[8], const float (&x1)[8], f32x4& acc0, f32x4& acc1) {
  const float4 a0 = *reinterpret_cast<const float4*>(arow + 32 * S);
  const float4 a1 = *reinterpret_cast<const float4*>(arow + 32 * S + 16);
  if (BF16) {
    const bf16x4 A0 = pack_bf16x4(a0.x, a0.y, a0.z, a0.w), A1 = pack_bf16x4(a1.x, a1.y, a1.z, a1.w);
    mfma_bf16_16x16x16(A0, pack_bf16x4(x0[0], x0[1], x0[2], x0[3]), acc0);
    mfma_bf16_16x16x16(A0, pack_bf16x4(x1[0], x1[1], x1[2], x1[3]), acc1);
    mfma_bf16_16x16x16(A1, pack_bf16x4(x0[4], x0[5], x0[6], x0[7]), acc0);
    mfma_bf16_16x16x16(A1, pack_bf16x4(x1[4], x1[5], x1[6], x1[7]), acc1);
    return;
  }
  mfma_f32_16x16x4(a0.x, x0[0], acc0); mfma_f32_16x16x4(a0.x, x1[0], acc1);
  mfma_f32_16x16x4(a0.y, x0[1], acc0); mfma_f32_16x16x4(a0.y, x1[1], acc1);
  mfma_f32_16x16x4(a0.z, x0[2], acc0); mfma_f32_16x16x4(a0.z, x1[2], acc1);
  mfma_f32_16x16x4(a0.w, x0[3], acc0); mfma_f32_16x16x4(a0.w, x1[3], acc1);
  mfma_f32_16x16x4(a1.x, x0[4], acc0); mfma_f32_16x16x4(a1.x, x1[4], acc1);
  mfma_f32_16x16x4(a1.y, x0[5], acc0); mfma_f32_16x16x4(a1.y, x1[5], acc1);
  mfma_f32_16x16x4(a1.z, x0[6], acc0); mfma_f32_16x16x4(a1.z, x1[6], acc1);
  mfma_f32_16x16x4(a1.w, x0[7], acc0); mfma_f32_16x16x4(a1.w, x1[7], acc1);
}

// Three-deep register ring over the K stages: stage S+2 is fetched while stage S computes, so a weight fragment has two
// full MFMA blocks (about 2000 cycles with two waves per SIMD) to arrive from L2.  The first two stages are requested by
// prefetch(), which the kernel calls one phase EARLY (weights depend on nothing computed here): the fill latency of each
// GEMM phase hides under the barrier / epilogue / loss code of the phase before it.
template <bool NT>
struct GemmPipe {
  BStage<NT> b0, b1;
  BufView wb;
  __device__ __forceinline__ void prefetch(int K, int Kvalid, const float* W, int H, int n0, int lane) {
    const int j = lane & 15, kq = lane >> 4;
    const int last = K / 32 - 1;
    wb = make_buf(W, (unsigned)Kvalid * (unsigned)H * 4u);
    b0.load(W, wb, H, K, 0, n0, j, kq);
    b1.load(W, wb, H, K, last < 1 ? last : 1, n0, j, kq);
  }
  // invariant at the loop top: b0 = stage S, b1 = stage S+1.  Stage indices past the end are clamped to the last stage
  // (a redundant, harmless load) instead of being skipped: straight-line code, exact vmcnt bookkeeping.
  template <bool BF16>
  __device__ __forceinline__ void run(const float* At, int AS, int K, const float* W, int H, int n0, int lane, f32x4& acc0, f32x4& acc1) {
    const int j = lane & 15, kq = lane >> 4;
    const float* arow = At + j * AS + 4 * kq;
    const int nst = K / 32, last = nst - 1;
    BStage<NT> b2;
    int S = 0;
    // MPPO_SCHED_FENCE: the machine scheduler otherwise sinks a stage's loads down to their first use (it minimises
    // register pressure), which is exactly the latency exposure the ring exists to avoid
    for (; S + 2 < nst; S += 3) {
      b2.load(W, wb, H, K, S + 2, n0, j, kq);
      MPPO_SCHED_FENCE();
      stage_mfma<BF16>(arow, S, b0.x0, b0.x1, acc0, acc1);
      MPPO_SCHED_FENCE();
      b0.load(W, wb, H, K, S + 3 < last ? S + 3 : last, n0, j, kq);
      MPPO_SCHED_FENCE();
      stage_mfma<BF16>(arow, S + 1, b1.x0, b1.x1, acc0, acc1);
      MPPO_SCHED_FENCE();
      b1.load(W, wb, H, K, S + 4 < last ? S + 4 : last, n0, j, kq);
      MPPO_SCHED_FENCE();
      stage_mfma<BF16>(arow, S + 2, b2.x0, b2.x1, acc0, acc1);
      MPPO_SCHED_FENCE();
    }
    if (S < nst) stage_mfma<BF16>(arow, S, b0.x0, b0.x1, acc0, acc1);
    if (S + 1 < nst) stage_mfma<BF16>(arow, S + 1, b1.x0, b1.x1, acc0, acc1);
  }
  // Two 16-row tiles per workgroup (RT = 2: rows j and j + 16 of a 32-row activation tile): every weight stage is fetched ONCE and
  // multiplied into both tiles - twice the MFMAs per byte streamed, the same ring otherwise.
  __device__ __forceinline__ void run2(const float* At, int AS, int K, const float* W, int H, int n0, int lane, f32x4 (&acc)[2][2]) {
    const int j = lane & 15, kq = lane >> 4;
    const float* arow = At + j * AS + 4 * kq;
    const float* arow2 = arow + 16 * AS;
    const int nst = K / 32, last = nst - 1;
    BStage<NT> b2;
    int S = 0;
    for (; S + 2 < nst; S += 3) {
      b2.load(W, wb, H, K, S + 2, n0, j, kq);
      MPPO_SCHED_FENCE();
      stage_mfma<false>(arow, S, b0.x0, b0.x1, acc[0][0], acc[0][1]);
      stage_mfma<false>(arow2, S, b0.x0, b0.x1, acc[1][0], acc[1][1]);
      MPPO_SCHED_FENCE();
      b0.load(W, wb, H, K, S + 3 < last ? S + 3 : last, n0, j, kq);
      MPPO_SCHED_FENCE();
      stage_mfma<false>(arow, S + 1, b1.x0, b1.x1, acc[0][0], acc[0][1]);
      stage_mfma<false>(arow2, S + 1, b1.x0, b1.x1, acc[1][0], acc[1][1]);
      MPPO_SCHED_FENCE();
      b1.load(W, wb, H, K, S + 4 < last ? S + 4 : last, n0, j, kq);
      MPPO_SCHED_FENCE();
      stage_mfma<false>(arow, S + 2, b2.x0, b2.x1, acc[0][0], acc[0][1]);
      stage_mfma<false>(arow2, S + 2, b2.x0, b2.x1, acc[1][0], acc[1][1]);
      MPPO_SCHED_FENCE();
    }
    if (S < nst) { stage_mfma<false>(arow, S, b0.x0, b0.x1, acc[0][0], acc[0][1]); stage_mfma<false>(arow2, S, b0.x0, b0.x1, acc[1][0], acc[1][1]); }
    if (S + 1 < nst) { stage_mfma<false>(arow, S + 1, b1.x0, b1.x1, acc[0][0], acc[0][1]); stage_mfma<false>(arow2, S + 1, b1.x0, b1.x1, acc[1][0], acc[1][1]); }
  }
};

// The same ring for a bf16 network with shadow copies: the B operand comes from the fragment-order bf16 copy of the weight
// (ppo_layout.h frag_index): a lane's eight values of a column tile and stage are 16 contiguous bytes (two 16-byte buffer loads instead of eight
// 8-byte ones, half the bytes, and no float -> bf16 conversion of weights in the loop).  `W` = fragment base of the matrix.
struct FragStage {
  float4 q0, q1;  // raw bits: q0 = tile 0 (k 0..3 of group 0 | group 1), q1 = tile 1
  __device__ __forceinline__ void load(const BufView& wb, int S, int nwaves, int wave, int lane) {
    const int uni = ((S * nwaves + wave_uniform(wave)) * 64) * 32;  // bytes: (stage, wave slab) block of 2 KB; the wave index in a scalar register (else: a waterfall loop around every load)
    q0 = buf_load_f4(wb, lane * (2 * kFragLaneElems), uni);
    q1 = buf_load_f4(wb, lane * (2 * kFragLaneElems) + 2 * kFragTileElems, uni);
  }
};
__device__ __forceinline__ void frag_mfma(const float* arow, int S, const FragStage& b, f32x4& acc0, f32x4& acc1) {
  const float4 a0 = *reinterpret_cast<const float4*>(arow + 32 * S);
  const float4 a1 = *reinterpret_cast<const float4*>(arow + 32 * S + 16);
  const bf16x4 A0 = pack_bf16x4(a0.x, a0.y, a0.z, a0.w), A1 = pack_bf16x4(a1.x, a1.y, a1.z, a1.w);
  const bf16x4 t0g0 = bf16x4_from_bits(b.q0.x, b.q0.y), t0g1 = bf16x4_from_bits(b.q0.z, b.q0.w);
  const bf16x4 t1g0 = bf16x4_from_bits(b.q1.x, b.q1.y), t1g1 = bf16x4_from_bits(b.q1.z, b.q1.w);
  mfma_bf16_16x16x16(A0, t0g0, acc0);
  mfma_bf16_16x16x16(A0, t1g0, acc1);
  mfma_bf16_16x16x16(A1, t0g1, acc0);
  mfma_bf16_16x16x16(A1, t1g1, acc1);
}
struct FragPipe {
  FragStage b0, b1;
  BufView wb;
  int nw;
  // same interface as GemmPipe (K = padded depth, H = columns, n0 = 32 * wave); Kvalid is implicit (the copy is zero-padded)
  __device__ __forceinline__ void prefetch(int K, int, const float* W, int H, int n0, int lane) {
    const int last = K / 32 - 1;
    nw = H / 32;
    wb = make_buf(W, (unsigned)K * (unsigned)H * 2u);
    b0.load(wb, 0, nw, n0 >> 5, lane);
    b1.load(wb, last < 1 ? last : 1, nw, n0 >> 5, lane);
  }
  template <bool BF16>
  __device__ __forceinline__ void run(const float* At, int AS, int K, const float*, int, int n0, int lane, f32x4& acc0, f32x4& acc1) {
    const int j = lane & 15, kq = lane >> 4, wv = n0 >> 5;
    const float* arow = At + j * AS + 4 * kq;
    const int nst = K / 32, last = nst - 1;
    FragStage b2;
    int S = 0;
    for (; S + 2 < nst; S += 3) {
      b2.load(wb, S + 2, nw, wv, lane);
      MPPO_SCHED_FENCE();
      frag_mfma(arow, S, b0, acc0, acc1);
      MPPO_SCHED_FENCE();
      b0.load(wb, S + 3 < last ? S + 3 : last, nw, wv, lane);
      MPPO_SCHED_FENCE();
      frag_mfma(arow, S + 1, b1, acc0, acc1);
      MPPO_SCHED_FENCE();
      b1.load(wb, S + 4 < last ? S + 4 : last, nw, wv, lane);
      MPPO_SCHED_FENCE();
      frag_mfma(arow, S + 2, b2, acc0, acc1);
      MPPO_SCHED_FENCE();
    }
    if (S < nst) frag_mfma(arow, S, b0, acc0, acc1);
    if (S + 1 < nst) frag_mfma(arow, S + 1, b1, acc0, acc1);
  }
  __device__ __forceinline__ void run2(const float*, int, int, const float*, int, int, int, f32x4 (&)[2][2]) {}  // (32-row tiles are a float-network form)
};
template <bool FRAG, bool NT> struct PipeSel { typedef GemmPipe<NT> type; };
template <bool NT> struct PipeSel<true, NT> { typedef FragPipe type; };

template <int LRW>
__device__ __forceinline__ float row32_sum(float x) {
  x = group16_sum(x);
  x += __shfl_xor(x, 16);
  return x;
}

// K-quad operands of the weight-gradient kernel.  Float network: one float4 per (quad of rows, column).  bf16 network (BF16 = true):
// the weight-gradient MFMAs round their operands to bf16 anyway, so the row pass stores them ROUNDED - 8 bytes per quad, the same
// index (quad_index) in units of bf16: half the bytes written here and read there, no conversion in the consumer, same results.
template <bool BF16>
__device__ __forceinline__ void store_quad(float* base, size_t qi, const float (&q)[4]) {
  if (BF16) stream_store(base + (qi >> 1), bf16x4_bits(pack_bf16x4(q[0], q[1], q[2], q[3])));
  else wt_store(base, qi, make_float4(q[0], q[1], q[2], q[3]));
}
// two adjacent columns (c0 even): 32 bytes as two float4, or 16 bytes of bf16
// WT: write-through (sc0 sc1) stores - the bytes leave this XCD's L2 while the kernel is still computing instead of waiting, dirty, for the
// end-of-kernel flush, and the weight-gradient launch finds them in memory: 13.9 -> 12.6 us there.  The row pass itself pays 0.9 us for it
// (a wave retires when its write-through stores are acknowledged); net 6.59 -> 6.555 ms per update.  All four operand stores or none:
// write-through for h1 / h2 / dZ2 and a streaming store for the last one (dZ1) measured 6.82 ms.
// float networks: SPLIT-PAIR column order inside a quad row (wgrad.h) - the even column of the lane's pair at qi, the odd one at qi2, half
// a quad row further: each of the two store instructions then writes whole lines across the lanes
template <bool BF16, bool WT = true>
__device__ __forceinline__ void store_quad2(float* base, size_t qi, size_t qi2, const float (&q0)[4], const float (&q1)[4]) {
  if (BF16) {
    const float2 a = bf16x4_bits(pack_bf16x4(q0[0], q0[1], q0[2], q0[3])), b = bf16x4_bits(pack_bf16x4(q1[0], q1[1], q1[2], q1[3]));
    if (WT) wt_store(base, qi >> 1, make_float4(a.x, a.y, b.x, b.y));
    else stream_store(base + (qi >> 1), make_float4(a.x, a.y, b.x, b.y));
  } else if (WT) {
    wt_store(base, qi, make_float4(q0[0], q0[1], q0[2], q0[3]));
    wt_store(base, qi2, make_float4(q1[0], q1[1], q1[2], q1[3]));
  } else {
    stream_store(base + qi, make_float4(q0[0], q0[1], q0[2], q0[3]));
    stream_store(base + qi2, make_float4(q1[0], q1[1], q1[2], q1[3]));
  }
}
// positions of the column pair (c0, c0 + 1), c0 even, of quad row `row` in an operand of `cols` columns: plain order for a bf16 network
// (one 16-byte store holds both), split-pair order for a float network
template <bool BF16>
__device__ __forceinline__ void pair_index(size_t row, size_t c0, size_t cols, size_t& qi, size_t& qi2) {
  if (BF16) { qi = quad_index(row, c0, cols); qi2 = qi + 4; }
  else { qi = ((row >> 2) * cols + (c0 >> 1)) * 4; qi2 = qi + (cols >> 1) * 4; }
}

// Gather role of a PRE launch: 16 rows of the NEXT minibatch (half = blockIdx.y picks two of the tile's four quads) from the
// trajectory into the k-quad buffer; rows past the minibatch are written as zeros (the buffer is an operand of the weight-gradient
// product and of the first layer).  Depends on the permutation only - it runs beside the workgroups that compute this step.
template <bool BF16>
__device__ __forceinline__ void gather_rows_tile(const FusedArgs& a, int tile, int half) {
  const int OP = a.OP, row0 = tile * FRT;
  for (int e = threadIdx.x; e < 2 * OP; e += blockDim.x) {
    const int qd = 2 * half + e / OP, c = e % OP;
    long ix[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = row0 + 4 * qd + j;
      ix[j] = a.idx_next[r < a.mb ? r : a.mb - 1];
    }
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x = a.b.obs[ix[j] * a.b.obs_ld + c];
      v[j] = row0 + 4 * qd + j < a.mb ? x : 0.f;
    }
    store_quad<BF16>(a.xnext, quad_index(row0 + 4 * qd, c, OP), v);
  }
  // the same rows' scalars of the loss, behind the observation quads (ppo_layout.h xquad_floats): float4 = one column of four rows
  if (a.b.action) {
    const int A = a.A, SC = A + 4;
    float* sq = a.xnext + xquad_obs_floats(OP, a.mb);
    for (int e = threadIdx.x; e < 2 * SC; e += blockDim.x) {
      const int qd = 2 * half + e / SC, c = e % SC;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = row0 + 4 * qd + j;
        const long ix = a.idx_next[r < a.mb ? r : a.mb - 1];
        const float x = c < A ? a.b.action[ix * a.b.act_ld + c] : c == A ? a.b.log_prob[ix] : c == A + 1 ? a.b.adv[ix] : c == A + 2 ? a.b.value[ix] : a.b.target[ix];
        v[j] = r < a.mb ? x : 0.f;
      }
      *reinterpret_cast<float4*>(sq + ((size_t)((row0 >> 2) + qd) * SC + c) * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// rollout launches have 2 x N/16 workgroups (512 at N = 4096): two per CU must be co-resident = 4 waves per SIMD (the second
// __launch_bounds__ argument is HIP's minimum waves per execution unit), i.e. at most 128 VGPRs
// OT = 16-wide output tiles of the head GEMM: 1 for A <= 16, 2 for A <= 32 (BASELINE configs[4]: 20 actuators)
// W2T: the backward product reads the transposed shadow copy of W2 (GradBufs::w2t) through the forward-style pipe: 23.9 -> 21.9 us
// PRE: engine minibatch loop - the x tile comes from the pre-gathered k-quad buffer (one trip to memory instead of index -> row), the
// gathered rows are not written again, and grid rows 2 and 3 gather the next step's rows
// RT: 16-row tiles per workgroup.  2 = a 32-row tile whose two halves share every weight stage (GemmPipe::run2): for minibatches whose
// 16-row tiling has more workgroups than the chip has CUs (BASELINE configs[4]: 2 x 160 = 320 on 256 - two rounds, the second one a
// quarter full); 2 x 80 workgroups of 32 rows are ONE round.  Float networks on the engine's pre-gathered path only.
template <bool BF16, bool ROLLOUT, int OT, bool W2T = false, bool PRE = false, int RT = 1>
__global__ void __launch_bounds__(512, ROLLOUT ? MPPO_ROLLOUT_WAVES : MPPO_TRAIN_WAVES) fused_mlp_kernel(FusedArgs a) {
  static_assert(RT == 1 || (RT == 2 && PRE && !BF16 && !ROLLOUT), "32-row tiles: float training row pass on pre-gathered rows");
  constexpr int SD = 16 * OT;  // row stride of the per-row output-space tiles
  constexpr int ROWS = FRT * RT;
  FT(0);
  FTW(56);
  // The role is decided from the launch geometry alone (grid rows 0, 1: the two networks; rows 2, 3: gather), not from a kernel
  // argument: a scalar load ahead of this branch would put one more (cold) trip to the argument segment in front of every row tile.
  // <ROLLOUT, PRE> together name the gather-ONLY launch (fused_gather_rows): every workgroup gathers.
  if (PRE && (ROLLOUT || blockIdx.y >= 2)) {  // uniform per workgroup
    if (RT == 2) {  // grid row 2 / 3: the first / second 16-row tile of this 32-row block, both halves
      gather_rows_tile<BF16>(a, 2 * (int)blockIdx.x + (int)blockIdx.y - 2, 0);
      gather_rows_tile<BF16>(a, 2 * (int)blockIdx.x + (int)blockIdx.y - 2, 1);
      return;
    }
    gather_rows_tile<BF16>(a, (int)blockIdx.x, ROLLOUT ? (int)blockIdx.y : (int)blockIdx.y - 2);
    return;
  }
  MPPO_DYN_SMEM(smem_raw);
  float* sm = reinterpret_cast<float*>(smem_raw);
  const int H = a.H, O = a.O, OP = a.OP, A = a.A, AP = a.AP;
  const int net = blockIdx.y + (ROLLOUT ? a.net0 : 0);  // 0 actor, 1 critic
  const int KP = (O + 31) & ~31;  // first-layer K padded to whole 32-k stages (x tile zero-padded)
  const int XS = KP + 4, HS = H + 4;
  const int R0 = ROWS * (XS > HS ? XS : HS);
  float* xt = sm;            // [ROWS][XS]  then dZ2 tile [ROWS][HS]
  float* h1t = sm + R0;      // [ROWS][HS]
  float* h2t = h1t + ROWS * HS;
  float* s_do = h2t + ROWS * HS;        // [ROWS][SD]  d mean (cols < A, zero beyond) | critic: col 0 = d value
  float* s_red = s_do + ROWS * SD;      // [ROWS][SD]  d log_std terms
  float* s_l = s_red + ROWS * SD;       // [ROWS]      per-row loss term
  float* s_hp = xt;                     // [H/32 waves][RT][OT][64 lanes][4]  partial head tiles: over the x tile, dead between layer 1 and dZ2
  const int t = threadIdx.x, nthr = blockDim.x, lane = t & 63, wave = t >> 6;
  const int row0 = blockIdx.x * ROWS;
  const bool tanh_act = net == 0 && a.use_tanh;
  const float* W1 = a.params + (net ? a.L.c_w1 : a.L.a_w1);
  const float* B1 = a.params + (net ? a.L.c_b1 : a.L.a_b1);
  const float* W2 = a.params + (net ? a.L.c_w2 : a.L.a_w2);
  const float* B2 = a.params + (net ? a.L.c_b2 : a.L.a_b2);
  const float* W3 = a.params + (net ? a.L.c_w3 : a.L.a_w3);
  const float* B3 = a.params + (net ? a.L.c_b3 : a.L.a_b3);
  const int nout = net ? 1 : A;

  const int n0 = 32 * wave;
  const int cj = lane & 15, rq = lane >> 4;
  // PRE: this step's observation tile is a contiguous block of the pre-gathered buffer: it depends on nothing but the kernel arguments and
  // is the longest chain in front of the first MFMA (memory -> registers -> LDS -> barrier), so it is requested before anything else
  float4 xq0 = make_float4(0.f, 0.f, 0.f, 0.f), xq1 = xq0;
  const int nxq = 4 * RT * OP;  // float4 elements of the k-quad tile, element e = (quad e / OP, column e % OP); contiguous in memory
  const float* xtile = PRE ? a.xpre + (size_t)(row0 >> 2) * OP * (BF16 ? 2 : 4) : nullptr;  // (bf16 network: 8-byte quads, see store_quad)
  auto load_xq = [&](int e) {
    if (BF16) { const float2 b = *reinterpret_cast<const float2*>(xtile + 2 * e); return bf16x4_unpack(b.x, b.y); }
    return *reinterpret_cast<const float4*>(xtile + 4 * e);
  };
  // (32-row tiles: up to eight elements per thread, all requested here - a loop of load-then-store pays one memory latency per element)
  constexpr int NXE = RT == 2 ? 6 : 1;
  float4 xqe[NXE];
  if (PRE) {
    if (t < nxq) xq0 = load_xq(t);
    if (t + nthr < nxq) xq1 = load_xq(t + nthr);
    if (RT == 2) {
#pragma unroll
      for (int k = 0; k < NXE; ++k) xqe[k] = load_xq(t + (2 + k) * nthr < nxq ? t + (2 + k) * nthr : nxq - 1);
    }
  }
  // ---- everything that depends on nothing computed here is requested now and consumed phases later: the first two
  // weight stages of layer 1, the biases, and the per-row scalars of the loss (index -> action / log_prob / advantage /
  // value / target: a dependent HBM chain of ~3 us that would otherwise sit between the head GEMM and the loss) ----
  constexpr bool FRAG = BF16 && W2T;  // weights from the bf16 fragment-order shadow copies
  const float* F1 = FRAG ? reinterpret_cast<const float*>(a.frag[net]) : W1;
  const float* F2 = FRAG ? reinterpret_cast<const float*>(a.frag[net] + (size_t)KP * H) : W2;
  typename PipeSel<FRAG, false>::type pipe1;
  pipe1.prefetch(KP, O, F1, H, n0, lane);
  const float2 bz1 = *reinterpret_cast<const float2*>(B1 + n0 + 2 * cj), bz2 = *reinterpret_cast<const float2*>(B2 + n0 + 2 * cj);
  const float adv_mean = ROLLOUT ? 0.f : a.adv_stat[0], adv_rstd = ROLLOUT ? 1.f : a.adv_stat[1];
  float ls[OT], b3v[OT];  // output o = cj + 16*ot of this lane
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) {
    const int o = cj + 16 * ot;
    ls[ot] = (net == 0 && o < A) ? a.params[a.L.log_std + o] : 0.f;
    b3v[ot] = o < nout ? B3[o] : 0.f;
  }
  // output-layer weights of this lane, straight into registers (they used to be staged in LDS: 20 KB at A = 20, which kept a
  // second workgroup off the CU): w3p = B operands of the head GEMM (k = 32*wave + 16g + 4rq + c, output o), w3q = B operands
  // of dZ2 = dOut . W3^T (hidden columns c0, c0 + 1, output ai = 4m + rq)
  float w3p[OT][8], w3q[4 * OT][2];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int gc = 0; gc < 8; ++gc) {
      const int o = cj + 16 * ot, k = 32 * wave + 16 * (gc >> 2) + 4 * rq + (gc & 3);
      w3p[ot][gc] = o < nout ? W3[k * nout + o] : 0.f;
    }
  if (!ROLLOUT) {
#pragma unroll
    for (int m = 0; m < 4 * OT; ++m) {
      const int ai = 4 * m + rq, c0 = n0 + 2 * cj;
      w3q[m][0] = ai < nout ? W3[c0 * nout + ai] : 0.f;
      w3q[m][1] = ai < nout ? W3[(c0 + 1) * nout + ai] : 0.f;
    }
  }
  FT(1);
  // ---- the index -> row chain.  Everything gathered by permutation index costs two dependent trips to memory (the index,
  // then what it points at).  All indices this thread needs - its four loss rows and the two rows whose observation chunks
  // it stages - are requested FIRST and together, then everything that depends on them in one batch; every load is
  // unconditional on a clamped address and masked afterwards (straight-line code: the compiler cannot hoist a load over a
  // branch, and round 1's per-row `if (on)` blocks serialised into eight consecutive memory latencies before the first MFMA).
  const int nx = FRT * (KP / 4);                 // float4 chunks of the x tile
  const int e0 = t, e1 = t + nthr;                // this thread's chunks (nthr >= nx / 2 for every supported H: checked on the host)
  const int xr0 = e0 / (KP / 4), xc0 = (e0 % (KP / 4)) * 4, xr1 = (e1 < nx ? e1 : e0) / (KP / 4), xc1 = ((e1 < nx ? e1 : e0) % (KP / 4)) * 4;
  const int gi0 = row0 + xr0 < a.mb ? row0 + xr0 : a.mb - 1, gi1 = row0 + xr1 < a.mb ? row0 + xr1 : a.mb - 1;
  const bool gather = !ROLLOUT && a.idx && !(a.skip & 64);
  long prow[4], xrow0 = gi0, xrow1 = gi1;
  bool pon[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = row0 + 4 * rq + r;
    pon[r] = i < a.mb;
    const int ic = pon[r] ? i : a.mb - 1;
    prow[r] = (gather && !PRE) ? (long)a.idx[ic] : (long)ic;  // (PRE: the loss scalars come pre-gathered, nothing here is addressed by index)
  }
  if (gather && !PRE) { xrow0 = a.idx[gi0]; xrow1 = a.idx[gi1]; }
  // dependent batch: the observation chunks (needed first), then the per-row scalars of the loss (needed four phases later)
  if (!PRE) {
    if (e0 < nx && xc0 < OP) xq0 = *reinterpret_cast<const float4*>(a.b.obs + xrow0 * a.b.obs_ld + xc0);
    if (e1 < nx && xc1 < OP) xq1 = *reinterpret_cast<const float4*>(a.b.obs + xrow1 * a.b.obs_ld + xc1);
  }
  // actor: action[o], old log_prob, advantage | critic: old value, target, -  (rollout: noise[o]).  RAW loads on clamped addresses, no
  // masking here: every use below is guarded by the same row / column conditions.  (A select on the loaded value in this place makes
  // it the value's first use, and the s_waitcnt goes where the first use is: with the network branch inside the row loop that was one
  // wait per row and array, eight memory latencies in series in front of the first MFMA - read off the ISA.)
  float pf0[RT][OT][4], pf1[RT][4], pf2[RT][4];  // (per 16-row tile of the workgroup)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pf1[rt][r] = pf2[rt][r] = 0.f;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) pf0[rt][ot][r] = 0.f;
    }
  if (ROLLOUT) {
    // (the rollout launch - four waves per SIMD, two workgroups per CU - measures 2 us FASTER with the masking select here, i.e. with the
    // wait for its noise in the prologue: 24.5 against 26.6 us; it keeps the masked form)
    if (net == 0 && a.noise) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const int o = cj + 16 * ot;
          const float v = a.noise[(size_t)prow[r] * A + (o < A ? o : A - 1)];
          pf0[0][ot][r] = (pon[r] && o < A) ? v : 0.f;
        }
    }
  } else if (PRE) {
    // pre-gathered by the previous launch (gather_rows_tile): the lane's four rows of a column are ONE float4 of the scalar quads behind
    // the observation quads - contiguous per row tile, no index, no dependent trip (rows past the minibatch read as 0; every use is masked)
    const int SC = A + 4;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
    const float* sq = a.xpre + xquad_obs_floats(OP, a.mb) + ((size_t)((row0 >> 2) + 4 * rt + rq) * SC) * 4;
    if (net == 0) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const int o = cj + 16 * ot;
        const float4 q = *reinterpret_cast<const float4*>(sq + 4 * (o < A ? o : A - 1));
        pf0[rt][ot][0] = q.x; pf0[rt][ot][1] = q.y; pf0[rt][ot][2] = q.z; pf0[rt][ot][3] = q.w;
      }
      const float4 q1 = *reinterpret_cast<const float4*>(sq + 4 * A), q2 = *reinterpret_cast<const float4*>(sq + 4 * (A + 1));
      pf1[rt][0] = q1.x; pf1[rt][1] = q1.y; pf1[rt][2] = q1.z; pf1[rt][3] = q1.w;
      pf2[rt][0] = q2.x; pf2[rt][1] = q2.y; pf2[rt][2] = q2.z; pf2[rt][3] = q2.w;
    } else {
      const float4 q0 = *reinterpret_cast<const float4*>(sq + 4 * (A + 2)), q1 = *reinterpret_cast<const float4*>(sq + 4 * (A + 3));
      pf0[rt][0][0] = q0.x; pf0[rt][0][1] = q0.y; pf0[rt][0][2] = q0.z; pf0[rt][0][3] = q0.w;
      pf1[rt][0] = q1.x; pf1[rt][1] = q1.y; pf1[rt][2] = q1.z; pf1[rt][3] = q1.w;
    }
    }
  } else if (net == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const int o = cj + 16 * ot;
        pf0[0][ot][r] = a.b.action[prow[r] * a.b.act_ld + (o < A ? o : A - 1)];
      }
      pf1[0][r] = a.b.log_prob[prow[r]];
      pf2[0][r] = a.b.adv[prow[r]];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pf0[0][0][r] = a.b.value[prow[r]];
      pf1[0][r] = a.b.target[prow[r]];
    }
  }

  FT(2);
  FTW(64);
  // ---- P0: gathered observation rows -> LDS (zero-padded to KP columns) ----
  if (PRE) {
    // a k-quad element is one column of four consecutive rows: four LDS words a row stride apart (consecutive lanes hold consecutive columns)
    auto put = [&](int e, const float4& q) {
      int qd = (int)(e >= OP) + (int)(e >= 2 * OP) + (int)(e >= 3 * OP);  // e / OP for e < 4 RT OP, without the division
      if (RT == 2) qd += (int)(e >= 4 * OP) + (int)(e >= 5 * OP) + (int)(e >= 6 * OP) + (int)(e >= 7 * OP);
      const int c = e - qd * OP;
      float* d = xt + (4 * qd) * XS + c;
      d[0] = q.x; d[XS] = q.y; d[2 * XS] = q.z; d[3 * XS] = q.w;
    };
    if (e0 < nxq) put(e0, xq0);
    if (e1 < nxq) put(e1, xq1);
    if (RT == 2) {
#pragma unroll
      for (int k = 0; k < NXE; ++k)
        if (t + (2 + k) * nthr < nxq) put(t + (2 + k) * nthr, xqe[k]);
    }
    for (int e = t + (RT == 2 ? 2 + NXE : 2) * nthr; e < nxq; e += nthr) put(e, load_xq(e));
    for (int r = t >> 5; r < ROWS; r += nthr >> 5)  // K padding of the first layer: fewer than 32 columns per row, one lane each
      if (OP + (t & 31) < KP) xt[r * XS + OP + (t & 31)] = 0.f;
  } else {
  if (e0 < nx) *reinterpret_cast<float4*>(xt + xr0 * XS + xc0) = xq0;
  if (e1 < nx) *reinterpret_cast<float4*>(xt + xr1 * XS + xc1) = xq1;
  }
  for (int e = t + 2 * nthr; e < nx && !PRE; e += nthr) {  // (only for widths with more than two chunks per thread)
    const int r = e / (KP / 4), c4 = (e % (KP / 4)) * 4;
    const int gi = row0 + r < a.mb ? row0 + r : a.mb - 1;
    const long row = gather ? a.idx[gi] : gi;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < OP) q = *reinterpret_cast<const float4*>(a.b.obs + row * a.b.obs_ld + c4);
    *reinterpret_cast<float4*>(xt + r * XS + c4) = q;
  }
  FTW(72);
  __syncthreads();
  FT(3);
  FTW(80);
  // ---- P1 / P2: hidden layers ----
  typename PipeSel<FRAG, false>::type pipe2;
  typename PipeSel<FRAG, !W2T>::type pipe5;
  const float* W2back = FRAG ? reinterpret_cast<const float*>(a.frag[net] + (size_t)KP * H + (size_t)H * H) : W2T ? a.w2t[net] : W2;
  for (int layer = 0; layer < 2; ++layer) {
    f32x4 acc0, acc1;
    f32x4 accr[2][2];  // (RT = 2: [row tile][column tile])
    for (int r = 0; r < 4; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; accr[0][0][r] = accr[0][1][r] = accr[1][0][r] = accr[1][1][r] = 0.f; }
#ifdef MPPO_FUSED_TIMERS
    if (!ROLLOUT && blockIdx.x == 40 && blockIdx.y == 0 && lane == 0) g_fused_t[24 + 16 * layer + wave] = __builtin_amdgcn_s_memtime();
#endif
    if (layer == 0) {
      if (RT == 2) pipe1.run2(xt, XS, KP, F1, H, n0, lane, accr);
      else if (!(a.skip & 1)) pipe1.template run<BF16>(xt, XS, KP, F1, H, n0, lane, acc0, acc1);
      if (!ROLLOUT) pipe2.prefetch(H, H, F2, H, n0, lane);  // arrives during the epilogue + barrier below
    } else {
      // rollout: 4 waves per SIMD hide the fill latency, and the 128-VGPR budget has no room for a cross-phase prefetch
      if (ROLLOUT) pipe2.prefetch(H, H, F2, H, n0, lane);
      if (RT == 2) pipe2.run2(h1t, HS, H, F2, H, n0, lane, accr);
      else if (!(a.skip & 2)) pipe2.template run<BF16>(h1t, HS, H, F2, H, n0, lane, acc0, acc1);
      if (!ROLLOUT) pipe5.prefetch(H, H, W2back, H, n0, lane);  // W2^T fragments of the backward product: hidden under heads / loss / dZ2
    }
    FT(4 + 2 * layer);
#ifdef MPPO_FUSED_TIMERS
    if (!ROLLOUT && blockIdx.x == 40 && blockIdx.y == 0 && lane == 0) g_fused_t[24 + 16 * layer + 8 + wave] = __builtin_amdgcn_s_memtime();
#endif
    if (layer == 0 && !ROLLOUT && net == 0 && !PRE) {
      // the gathered rows, k-quad layout [mb/4][OP][4], for the first layer's weight gradient (the actor workgroup writes them).
      // Here, after this wave's share of the first GEMM and before the barrier, the copy costs the early waves nothing: they
      // would wait for the SIMD's second wave anyway (the x tile stays intact until the head partials overwrite it in P3).
      for (int e = t; e < 4 * OP; e += nthr) {
        const int qd = e / OP, c = e - qd * OP;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = row0 + 4 * qd + j < a.mb ? xt[(4 * qd + j) * XS + c] : 0.f;
        store_quad<BF16>(a.xmb, quad_index(row0 + 4 * qd, c, OP), v);
      }
    }
    float* ht = layer == 0 ? h1t : h2t;
    float* hg = layer == 0 ? a.h1[net] : a.h2[net];
    const int c0 = n0 + 2 * cj;  // the wave's two interleaved column tiles: c0, c0 + 1
    const float2 bz = layer == 0 ? bz1 : bz2;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
    float q0[4], q1[4];  // the lane's 4 rows x 2 columns: one k-quad of column c0 and one of column c0 + 1
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * rt + 4 * rq + r;
      float v0 = (RT == 2 ? accr[rt][0][r] : acc0[r]) + bz.x, v1 = (RT == 2 ? accr[rt][1][r] : acc1[r]) + bz.y;
      if (tanh_act) { v0 = fused_tanh(v0); v1 = fused_tanh(v1); } else { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      *reinterpret_cast<float2*>(ht + rr * HS + c0) = make_float2(v0, v1);
      const bool on = row0 + rr < a.mb;  // rows past the minibatch are zero in the quad buffers (they are contracted over)
      q0[r] = on ? v0 : 0.f; q1[r] = on ? v1 : 0.f;
    }
    if (!ROLLOUT && !(a.skip & 32)) {  // two adjacent columns: 32 bytes per lane (bf16: 16)
      size_t qi, qi2;
      pair_index<BF16>(row0 + 16 * rt + 4 * rq, c0, H, qi, qi2);
      store_quad2<BF16>(hg, qi, qi2, q0, q1);
    }
    }
    __syncthreads();
    FT(5 + 2 * layer);
  }

  FT(8);
  // ---- P3: output layer on the matrix cores: OT 16x16 tiles (rows x outputs), K = H split over the waves ----
  // wave w multiplies h2[:, 32w .. 32w+32) by W3[32w .. 32w+32, :]; the H/32 partial tiles are summed through LDS.
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    f32x4 hp[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
      for (int r = 0; r < 4; ++r) hp[ot][r] = 0.f;
    if (!(a.skip & 4)) {
      const float* arow = h2t + (cj + 16 * rt) * HS + 4 * rq + 32 * wave;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const float4 av = *reinterpret_cast<const float4*>(arow + 16 * g);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          mfma_f32_16x16x4(av.x, w3p[ot][4 * g + 0], hp[ot]); mfma_f32_16x16x4(av.y, w3p[ot][4 * g + 1], hp[ot]);
          mfma_f32_16x16x4(av.z, w3p[ot][4 * g + 2], hp[ot]); mfma_f32_16x16x4(av.w, w3p[ot][4 * g + 3], hp[ot]);
        }
      }
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
      *reinterpret_cast<float4*>(s_hp + (((wave * RT + rt) * OT + ot) * 64 + lane) * 4) = make_float4(hp[ot][0], hp[ot][1], hp[ot][2], hp[ot][3]);
  }
  __syncthreads();
  FT(9);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    // every wave adds the partial tiles (same order: identical values everywhere); lane (cj = lane&15, q = lane>>4) holds
    // out[ot][r] = output cj + 16*ot of row 4q + r.  A DPP row of 16 lanes therefore spans all outputs of a row: row
    // reductions are group16 sums (of the per-lane sum over ot), one per accumulator register.  Only wave 0 stores.
    float out[OT][4];
    const int nw = nthr >> 6;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      out[ot][0] = out[ot][1] = out[ot][2] = out[ot][3] = 0.f;
      for (int w = 0; w < nw; ++w) {
        const float4 q = *reinterpret_cast<const float4*>(s_hp + (((w * RT + rt) * OT + ot) * 64 + lane) * 4);
        out[ot][0] += q.x; out[ot][1] += q.y; out[ot][2] += q.z; out[ot][3] += q.w;
      }
    }
    const bool st = wave == 0;
    float sum_ls_l = 0.f, b3_l = 0.f;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) { sum_ls_l += ls[ot]; b3_l += b3v[ot]; }
    if (ROLLOUT) {
      // pi.sample + pi.log_prob (train.py:158-160) / value (train.py:157,182); same arithmetic as head_kernel<.,false>
      if (net == 0) {
        const float sum_ls = group16_sum(sum_ls_l);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = row0 + 4 * rq + r;
          const bool on = i < a.mb;
          float z2 = 0.f;
#pragma unroll
          for (int ot = 0; ot < OT; ++ot) {
            const int o = cj + 16 * ot;
            if (on && o < A) {
              const float mean = out[ot][r] + b3v[ot];
              const float act = mean + __expf(ls[ot]) * pf0[0][ot][r];
              const float z = (act - mean) * __expf(-ls[ot]);
              z2 += z * z;
              if (st) {
                a.action[(size_t)i * A + o] = act;
                if (a.mean_out) a.mean_out[(size_t)i * AP + o] = mean;
              }
            }
          }
          const float ss = group16_sum(z2);
          if (st && on && cj == 0) a.log_prob[i] = -0.5f * ss - sum_ls - 0.5f * (float)A * kLog2PiF;
        }
      } else {
        const float b3c = group16_sum(b3v[0]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = row0 + 4 * rq + r;
          const float v = group16_sum(cj == 0 ? out[0][r] : 0.f) + b3c;
          if (st && i < a.mb && cj == 0) a.value[i] = v;
        }
      }
      return;
    }
    if (net == 0) {
      const float sum_ls = group16_sum(sum_ls_l);
      float dmq[OT][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 16 * rt + 4 * rq + r, i = row0 + rr;
        const bool on = i < a.mb;
        float z[OT], zz = 0.f;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          z[ot] = 0.f;
          if (on && cj + 16 * ot < A) z[ot] = (pf0[rt][ot][r] - (out[ot][r] + b3v[ot])) * __expf(-ls[ot]);
          zz += z[ot] * z[ot];
        }
        const float ss = group16_sum(zz);
        float la = 0.f, dlogp = 0.f;
        if (on) {
          const float logp = -0.5f * ss - sum_ls - 0.5f * (float)A * kLog2PiF;
          const float ratio = __expf(logp - pf1[rt][r]);
          const float g = (pf2[rt][r] - adv_mean) * adv_rstd;
          const float la1 = ratio * g;
          const float la2 = fminf(fmaxf(ratio, 1.f - a.lc.clip_eps), 1.f + a.lc.clip_eps) * g;
          la = -fminf(la1, la2) * a.inv_count;
          const bool unclipped = (ratio >= 1.f - a.lc.clip_eps) && (ratio <= 1.f + a.lc.clip_eps);
          dlogp = (unclipped || la1 < la2) ? -g * ratio * a.inv_count : 0.f;
        }
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) dmq[ot][r] = (on && cj + 16 * ot < A) ? dlogp * z[ot] * __expf(-ls[ot]) : 0.f;
        if (st) {
#pragma unroll
          for (int ot = 0; ot < OT; ++ot) {
            const int o = cj + 16 * ot;
            s_do[rr * SD + o] = dmq[ot][r];
            s_red[rr * SD + o] = o < A ? dlogp * (z[ot] * z[ot] - 1.f) : 0.f;
          }
          if (cj == 0) s_l[rr] = la;
        }
      }
      if (st) {  // d mean, k-quad layout [mb/4][DP][4]: the lane's four rows of output o are one float4 (columns A .. AP-1: zeros)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const int o = cj + 16 * ot;
          if (o < AP) store_quad<BF16>(a.dout, quad_index(row0 + 16 * rt + 4 * rq, o, a.DP), dmq[ot]);
        }
      }
    } else {
      const float b3c = group16_sum(b3v[0]);  // lane cj = 0 holds the critic's single output bias
      float dvq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 16 * rt + 4 * rq + r, i = row0 + rr;
        const bool on = i < a.mb;
        const float vnew = group16_sum(cj == 0 ? out[0][r] : 0.f) + b3c;
        float lv = 0.f, dv = 0.f;
        if (on) {
          const float ov = pf0[rt][0][r], tg = pf1[rt][r];
          const float vc = ov + fminf(fmaxf(vnew - ov, -a.lc.clip_eps), a.lc.clip_eps);
          const float vl1 = (vnew - tg) * (vnew - tg), vl2 = (vc - tg) * (vc - tg);
          lv = 0.5f * fmaxf(vl1, vl2) * a.inv_count;
          const bool vin = fabsf(vnew - ov) <= a.lc.clip_eps;
          dv = (vin || vl1 > vl2) ? (vnew - tg) * a.inv_count * a.lc.vf_coef : 0.f;
        }
        dvq[r] = dv;
        if (st) {
#pragma unroll
          for (int ot = 0; ot < OT; ++ot) s_do[rr * SD + cj + 16 * ot] = (cj == 0 && ot == 0) ? dv : 0.f;
          if (cj == 0) s_l[rr] = lv;
        }
      }
      // d value in column AP of dOut (k-quad layout), columns AP+1 .. AP+3 zero
      if (st && cj < 4) {
        const float zq[4] = {cj == 0 ? dvq[0] : 0.f, cj == 0 ? dvq[1] : 0.f, cj == 0 ? dvq[2] : 0.f, cj == 0 ? dvq[3] : 0.f};
        store_quad<BF16>(a.dout, quad_index(row0 + 16 * rt + 4 * rq, AP + cj, a.DP), zq);
      }
    }
  }
  __syncthreads();
  FT(10);
  // per-workgroup partial sums: partial[blockIdx.x][4+AP]: col 0 actor loss, col 1 value loss, 4+a d log_std[a]
  {
    float* prow_out = a.partial + (size_t)blockIdx.x * (4 + AP);
    if (net == 0) {
      if (t == 0) { float s = 0.f; for (int r = 0; r < ROWS; ++r) s += s_l[r]; prow_out[0] = s; }
      if (t >= 4 && t < 4 + AP) { float s = 0.f; if (t - 4 < A) for (int r = 0; r < ROWS; ++r) s += s_red[r * SD + (t - 4)]; prow_out[t] = s; }
    } else if (t == 0) {
      float s = 0.f; for (int r = 0; r < ROWS; ++r) s += s_l[r]; prow_out[1] = s;
    }
  }
  FT(11);
  // ---- P4: dZ2 = (dOut . W3^T) * act'(h2) on the matrix cores (K = outputs padded to 16*OT) -> LDS (over the dead x tile) + global ----
  float* dzt = xt;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    f32x4 d0, d1;
    for (int r = 0; r < 4; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
    const int c0 = n0 + 2 * cj;
    if (!(a.skip & 8)) {
#pragma unroll
      for (int m = 0; m < 4 * OT; ++m) {
        const int ai = 4 * m + rq;  // output index contracted over
        const float av = s_do[(cj + 16 * rt) * SD + ai];
        mfma_f32_16x16x4(av, w3q[m][0], d0); mfma_f32_16x16x4(av, w3q[m][1], d1);
      }
    }
    float q0[4], q1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * rt + 4 * rq + r;
      const float2 hv = *reinterpret_cast<const float2*>(h2t + rr * HS + c0);
      const float z0 = tanh_act ? d0[r] * (1.f - hv.x * hv.x) : (hv.x > 0.f ? d0[r] : 0.f);
      const float z1 = tanh_act ? d1[r] * (1.f - hv.y * hv.y) : (hv.y > 0.f ? d1[r] : 0.f);
      *reinterpret_cast<float2*>(dzt + rr * HS + c0) = make_float2(z0, z1);
      const bool on = row0 + rr < a.mb;
      q0[r] = on ? z0 : 0.f; q1[r] = on ? z1 : 0.f;
    }
    if (!(a.skip & 32)) {
      size_t qi, qi2;
      pair_index<BF16>(row0 + 16 * rt + 4 * rq, c0, H, qi, qi2);
      store_quad2<BF16>(a.dz2[net], qi, qi2, q0, q1);
    }
  }
  __syncthreads();
  FT(12);
  // ---- P5: dZ1 = (dZ2 . W2^T) * act'(h1) ----
  {
    f32x4 acc0, acc1;
    f32x4 accr[2][2];
    for (int r = 0; r < 4; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; accr[0][0][r] = accr[0][1][r] = accr[1][0][r] = accr[1][1][r] = 0.f; }
    if (RT == 2) pipe5.run2(dzt, HS, H, W2back, H, n0, lane, accr);
    else if (!(a.skip & 16)) pipe5.template run<BF16>(dzt, HS, H, W2back, H, n0, lane, acc0, acc1);
    FT(13);
    const int c0 = n0 + 2 * cj;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
    float q0[4], q1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * rt + 4 * rq + r;
      const bool on = row0 + rr < a.mb;
      const float2 gq = *reinterpret_cast<const float2*>(h1t + rr * HS + c0);
      const float a0 = RT == 2 ? accr[rt][0][r] : acc0[r], a1 = RT == 2 ? accr[rt][1][r] : acc1[r];
      const float d0 = tanh_act ? a0 * (1.f - gq.x * gq.x) : (gq.x > 0.f ? a0 : 0.f);
      const float d1 = tanh_act ? a1 * (1.f - gq.y * gq.y) : (gq.y > 0.f ? a1 : 0.f);
      q0[r] = on ? d0 : 0.f; q1[r] = on ? d1 : 0.f;
    }
    {
      size_t qi, qi2;
      pair_index<BF16>(row0 + 16 * rt + 4 * rq, c0, H, qi, qi2);
      store_quad2<BF16>(a.dz1[net], qi, qi2, q0, q1);
    }
    }
  }
  FT(14);
}

#ifdef MPPO_FUSED_TIMERS
}  // namespace mppo
extern "C" int32_t mppo_debug_fused_timers(unsigned long long* out24) {
  MPPO_CHECK_HIP(hipMemcpyFromSymbol(out24, HIP_SYMBOL(mppo::g_fused_t), sizeof(unsigned long long) * (24 + 64)));
  return MPPO_OK;
}
namespace mppo {
#endif

#include "fused_bf16.h"  // bf16_rowpass_kernel: the training row pass of a bf16 network (BASELINE configs[3])

size_t fused_smem_bytes(int O, int A, int H, int RT) {
  const int KP = (O + 31) & ~31, XS = KP + 4, HS = H + 4, OT = A > 16 ? 2 : 1;
  const size_t rows = (size_t)FRT * RT, R0 = rows * (XS > HS ? XS : HS);
  // x tile (later: partial head tiles, then the dZ2 tile) | h1 | h2 | dOut, d log_std terms | per-row loss
  return sizeof(float) * (R0 + 2 * rows * HS + 2 * rows * 16 * OT + rows);
}
size_t fused_smem_bytes(int O, int A, int H) { return fused_smem_bytes(O, A, H, 1); }

// Rows per workgroup of the training row pass for this launch: 32 (two 16-row tiles sharing every weight stage) when the 16-row tiling
// would need more workgroups than the chip has CUs - a second, mostly empty round - and the 32-row one fits in one; 16 otherwise.
// Float networks on the engine's pre-gathered path, whole 32-row tiles (the quad buffers are sized in 16-row tiles), H = 256.
#ifdef MPPO_EMU
constexpr int kFusedCUs = 4;  // (the emulator's parity tests reach the 32-row form with 64-row minibatches of small networks)
constexpr bool kFusedRt2AnyH = true;
#else
constexpr int kFusedCUs = 256;  // MI355X; one 512-thread workgroup per CU at H = 256
constexpr bool kFusedRt2AnyH = false;
#endif
int fused_rows_per_workgroup(const mppo_net_t& net, int mb, bool pre) {
  const int t16 = cdiv(mb, FRT);
  if (!pre || net.bf16 || (net.H != 256 && !kFusedRt2AnyH) || (t16 & 1) || 2 * t16 <= kFusedCUs) return FRT;
  if (fused_smem_bytes(net.O, net.A, net.H, 2) > 160 * 1024) return FRT;
  return 2 * FRT;
}

bool fused_supported(const mppo_net_t& net, const mppo_batch_t& b) {
  return net_layers(net) == 2 && net.H % 32 == 0 && net.H >= 32 && net.H <= 256 && net.A <= 32 && b.obs_ld == net.OP && (net.OP % 4) == 0 &&
         (reinterpret_cast<uintptr_t>(b.obs) & 15) == 0 && fused_smem_bytes(net.O, net.A, net.H) <= 160 * 1024 &&
         (param_layout(net).c_w2 % 4) == 0 &&  // float4 rows of W2 in the backward product
         wgrad_tile_bound(net.O, net.A, net.H) <= kSqSlots;  // the weight-gradient launch that follows the row pass: one workgroup per 32 x 32 tile, at most kSqSlots
}

bool fused_rollout_supported(const mppo_net_t& net, const float* obs, int obs_ld) {
  return net_layers(net) == 2 && net.H % 32 == 0 && net.H >= 32 && net.H <= 256 && net.A <= 32 && obs_ld == net.OP && (net.OP % 4) == 0 && (reinterpret_cast<uintptr_t>(obs) & 15) == 0 &&
         fused_smem_bytes(net.O, net.A, net.H) <= 160 * 1024;
}

static int32_t fused_set_smem(size_t smem) {
#define MPPO_FUSED_ATTR(B, R, T) \
  MPPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_kernel<B, R, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem))
  MPPO_FUSED_ATTR(false, false, 1); MPPO_FUSED_ATTR(true, false, 1); MPPO_FUSED_ATTR(false, true, 1); MPPO_FUSED_ATTR(true, true, 1);
  MPPO_FUSED_ATTR(false, false, 2); MPPO_FUSED_ATTR(true, false, 2); MPPO_FUSED_ATTR(false, true, 2); MPPO_FUSED_ATTR(true, true, 2);
#undef MPPO_FUSED_ATTR
#define MPPO_FUSED_ATTR(B, T) \
  MPPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_kernel<B, false, T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem))
  MPPO_FUSED_ATTR(false, 1); MPPO_FUSED_ATTR(true, 1); MPPO_FUSED_ATTR(false, 2); MPPO_FUSED_ATTR(true, 2);
#undef MPPO_FUSED_ATTR
#define MPPO_FUSED_ATTR(B, T) \
  MPPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_kernel<B, false, T, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem))
  MPPO_FUSED_ATTR(false, 1); MPPO_FUSED_ATTR(true, 1); MPPO_FUSED_ATTR(false, 2); MPPO_FUSED_ATTR(true, 2);
#undef MPPO_FUSED_ATTR
  MPPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_kernel<false, false, 1, true, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  MPPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_kernel<false, false, 2, true, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  MPPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_kernel<true, true, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  MPPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_kernel<true, true, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  return MPPO_OK;
}

int32_t fused_forward_backward(const mppo_net_t& net, const float* params, const mppo_batch_t& batch, const int* idx, int mb, const float* adv_stat,
                               float inv_count, const mppo_loss_cfg_t& lc, const GradBufs& g, hipStream_t stream, const XPre* pre) {
  FusedArgs a{};
  a.mb = mb; a.O = net.O; a.OP = net.OP; a.A = net.A; a.AP = g.f.AP; a.DP = g.f.AP + 4; a.H = net.H; a.use_tanh = net.use_tanh;
  a.params = params; a.L = param_layout(net); a.b = batch; a.idx = idx; a.adv_stat = adv_stat; a.inv_count = inv_count; a.lc = lc;
  a.h1[0] = g.f.h1a; a.h1[1] = g.f.h1c; a.h2[0] = g.f.h2a; a.h2[1] = g.f.h2c; a.dz2[0] = g.dz2a; a.dz2[1] = g.dz2c; a.dz1[0] = g.dz1a; a.dz1[1] = g.dz1c;
  a.dout = g.dout; a.xmb = g.xmb; a.partial = g.partial;
  const bool w2t = g.w2t_valid && g.w2t;
  a.w2t[0] = g.w2t; a.w2t[1] = g.w2t ? g.w2t + pad4((size_t)net.H * net.H) : nullptr;
  a.frag[0] = g.frag; a.frag[1] = g.frag ? g.frag + g.frag_net_stride : nullptr;
#ifdef MPPO_EXPERIMENTS
  static const int skip = [] { const char* e = getenv("MPPO_FUSED_SKIP"); return e ? atoi(e) : 0; }();  // phase-budget measurements: skips GEMM phases (results are garbage)
  a.skip = skip;
#else
  a.skip = 0;
#endif
  const size_t smem = fused_smem_bytes(net.O, net.A, net.H);
  static thread_local size_t attr_for = 0;
  if (smem > 64 * 1024 && attr_for < smem) {
    MPPO_TRY(fused_set_smem(smem));
    attr_for = smem;
  }
  MPPO_REQUIRE(!pre || (w2t && idx), "fused_forward_backward: pre-gathered rows need the shadow copies and a permutation");
  if (pre) { a.xpre = pre->cur; a.xnext = pre->next; a.idx_next = pre->idx_next; }
  if (fused_rows_per_workgroup(net, mb, pre != nullptr) == 2 * FRT) {  // 32-row tiles: one round of workgroups instead of two
    const size_t smem2 = fused_smem_bytes(net.O, net.A, net.H, 2);
    if (attr_for < smem) { MPPO_TRY(fused_set_smem(smem)); attr_for = smem; }  // (sets the 32-row instantiations' limit too)
    const dim3 grid2(cdiv(mb, 2 * FRT), pre->idx_next ? 4 : 2), block2(2 * net.H);
    if (net.A > 16) hipLaunchKernelGGL((fused_mlp_kernel<false, false, 2, true, true, 2>), grid2, block2, smem2, stream, a);
    else hipLaunchKernelGGL((fused_mlp_kernel<false, false, 1, true, true, 2>), grid2, block2, smem2, stream, a);
    MPPO_CHECK_LAUNCH("fused_mlp_kernel<32 rows>");
    return MPPO_OK;
  }
  const dim3 grid(cdiv(mb, FRT), pre && pre->idx_next ? 4 : 2), block(2 * net.H);
  if (pre && g.frag && bf16_rowpass_supported(net)) {  // the engine's minibatch loop of a bf16 network: the kernel designed for it (fused_bf16.h)
    const size_t sb = bf16_rowpass_smem_bytes(net.O, net.A, net.H);
#ifdef MPPO_TRACE_BF16
    fprintf(stderr, "[trace] bf16_rowpass_kernel mb=%d\n", mb);
#endif
    const bool exact = ((net.O + 31) & ~31) == 32 * kBf16RowpassNS1 && net.H == 256;  // the registers are filled exactly: no zero stages, no selects
    if (net.A > 16) {
      if (exact) hipLaunchKernelGGL((bf16_rowpass_kernel<2, kBf16RowpassNS1, true>), grid, block, sb, stream, a);
      else hipLaunchKernelGGL((bf16_rowpass_kernel<2, kBf16RowpassNS1, false>), grid, block, sb, stream, a);
    } else {
      if (exact) hipLaunchKernelGGL((bf16_rowpass_kernel<1, kBf16RowpassNS1, true>), grid, block, sb, stream, a);
      else hipLaunchKernelGGL((bf16_rowpass_kernel<1, kBf16RowpassNS1, false>), grid, block, sb, stream, a);
    }
    MPPO_CHECK_LAUNCH("bf16_rowpass_kernel");
    return MPPO_OK;
  }
#define MPPO_FUSED_GO(B, T) do { if (pre) hipLaunchKernelGGL((fused_mlp_kernel<B, false, T, true, true>), grid, block, smem, stream, a); \
                                else if (w2t) hipLaunchKernelGGL((fused_mlp_kernel<B, false, T, true>), grid, block, smem, stream, a); \
                                else hipLaunchKernelGGL((fused_mlp_kernel<B, false, T, false>), grid, block, smem, stream, a); } while (0)
  if (net.A > 16) {
    if (net.bf16) MPPO_FUSED_GO(true, 2); else MPPO_FUSED_GO(false, 2);
  } else {
    if (net.bf16) MPPO_FUSED_GO(true, 1); else MPPO_FUSED_GO(false, 1);
  }
#undef MPPO_FUSED_GO
  MPPO_CHECK_LAUNCH("fused_mlp_kernel");
  return MPPO_OK;
}

// The first optimizer step of an update has no previous launch to gather its rows: the gather role alone (every workgroup of a PRE
// <ROLLOUT, PRE> instantiation).
int32_t fused_gather_rows(const mppo_net_t& net, const mppo_batch_t& batch, const int* idx, int mb, float* dst, hipStream_t stream) {
  MPPO_REQUIRE(idx && dst && batch.obs, "fused_gather_rows: null argument");
  FusedArgs a{};
  a.mb = mb; a.O = net.O; a.OP = net.OP; a.A = net.A; a.H = net.H; a.b = batch; a.idx_next = idx; a.xnext = dst;
  if (net.bf16) hipLaunchKernelGGL((fused_mlp_kernel<true, true, 1, true, true>), dim3(cdiv(mb, FRT), 2), dim3(256), 0, stream, a);  // (bf16 quads)
  else hipLaunchKernelGGL((fused_mlp_kernel<false, true, 1, true, true>), dim3(cdiv(mb, FRT), 2), dim3(256), 0, stream, a);
  MPPO_CHECK_LAUNCH("fused_mlp_kernel<gather>");
  return MPPO_OK;
}

// Rollout policy step on n rows in ONE launch: both hidden layers, the heads, sample + log-prob, value (train.py:157-160);
// noise == nullptr: critic only (bootstrap value, train.py:182); value == nullptr: actor only.  Replaces two layer GEMM launches + the head kernel.
int32_t fused_policy_forward(const mppo_net_t& net, const float* params, int n, const float* obs, int obs_ld, const float* noise, float* action, float* log_prob,
                             float* value, float* mean_out, int AP, hipStream_t stream, const unsigned short* frag, size_t frag_net_stride) {
  FusedArgs a{};
  a.frag[0] = frag; a.frag[1] = frag ? frag + frag_net_stride : nullptr;
  const bool use_frag = net.bf16 && frag;
  a.mb = n; a.O = net.O; a.OP = net.OP; a.A = net.A; a.AP = AP; a.DP = AP + 4; a.H = net.H; a.use_tanh = net.use_tanh;
  a.params = params; a.L = param_layout(net);
  a.b.obs = obs; a.b.obs_ld = obs_ld;
  a.noise = noise; a.action = action; a.log_prob = log_prob; a.value = value; a.mean_out = mean_out;
  a.net0 = noise ? 0 : 1;
  const size_t smem = fused_smem_bytes(net.O, net.A, net.H);
  static thread_local size_t attr_for = 0;
  if (smem > 64 * 1024 && attr_for < smem) {
    MPPO_TRY(fused_set_smem(smem));
    attr_for = smem;
  }
  const dim3 grid(cdiv(n, FRT), noise && value ? 2 : 1);  // (noise without value: the actor alone)
  const dim3 block(2 * net.H);
  if (net.A > 16) {
    if (use_frag) hipLaunchKernelGGL((fused_mlp_kernel<true, true, 2, true>), grid, block, smem, stream, a);
    else if (net.bf16) hipLaunchKernelGGL((fused_mlp_kernel<true, true, 2>), grid, block, smem, stream, a);
    else hipLaunchKernelGGL((fused_mlp_kernel<false, true, 2>), grid, block, smem, stream, a);
  } else {
    if (use_frag) hipLaunchKernelGGL((fused_mlp_kernel<true, true, 1, true>), grid, block, smem, stream, a);
    else if (net.bf16) hipLaunchKernelGGL((fused_mlp_kernel<true, true, 1>), grid, block, smem, stream, a);
    else hipLaunchKernelGGL((fused_mlp_kernel<false, true, 1>), grid, block, smem, stream, a);
  }
  MPPO_CHECK_LAUNCH("fused_mlp_kernel<rollout>");
  return MPPO_OK;
}

}  // namespace mppo
