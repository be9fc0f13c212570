// k_wgrad.hip — all weight and bias gradients of one minibatch, COMPLETE, in one launch.
//
// What `jax.value_and_grad(_loss_fn)` (reference minppo/train.py:246-247) derives for the six Dense layers:
//   dW = H_prev^T . dZ   (contraction over the minibatch rows, K = mb)      db = column sums of dZ
// for actor and critic, plus the `log_std` gradient, the four loss scalars (train.py:240-243) and the per-workgroup sums of
// squares that `clip_by_global_norm` needs (train.py:117).  Replaces two launches of round 1 (split-K GEMM into 8 slabs, then
// a slab-sum kernel: 17.2 + 5.0 us) - there are no slabs any more:
//
//   * a workgroup owns ONE 32x32 output tile over the WHOLE K range: 8 waves, wave g contracts rows [g K/8, (g+1) K/8) on
//     v_mfma_f32_32x32x2_f32 (bf16: 32x32x8) and the eight accumulators are summed through LDS in wave order (a fixed
//     order: the gradient is bit-reproducible).
//   * exactly one workgroup per CU at the headline shape (O = 225, H = 256): 240 full tiles + 16 head-layer tiles = 256.
//     One workgroup too many costs 5 us (a CU with two workgroups needs twice as long; measured 21.8 against 16.8 us), so
//     the row band of the first layer that holds a single observation row (225 = 7 x 32 + 1) is not given 32x32 tiles of its
//     own: such a THIN band (<= 4 rows) is contracted on the vector ALU by the workgroups of the problem's FIRST row band,
//     which already hold the B operand it needs in registers (the same workgroups that sum B's columns for the bias).
//   * no operand staging: the row pass (k_fused.hip) writes the activations in K-QUAD layout [K/4][cols][4], so a lane's
//     operand for four MFMAs is ONE 16-byte load, coalesced over the 32 columns of the tile (512 bytes per half wave);
//     range-checked buffer loads return 0 past the last quad and for columns outside the operand.
//   * 328 KB of operands per workgroup come from L2 (84 MB per launch against 7.5 MB of distinct bytes): the price of
//     not exchanging partial sums between workgroups, paid at L2 bandwidth instead of with a kernel boundary.
// Measured and rejected on the way: an LDS-staged 64x64 split-K kernel (16.7 us + the reduce kernel), and the
// same with the K chunks of a tile summed inside the launch by the tile's last workgroup to arrive (agent-scope partials +
// arrival counter: correct in a 600-launch bitwise stress test, 24 us).
#include <wave_ops.h>

#include <cstdlib>
#include <vector>

#include "mppo_common.h"
#include "ppo_layout.h"
#include "wgrad.h"
#include "peer.h"

namespace mppo {

constexpr int WTILE = 32;
constexpr int WWAVES = 8;
constexpr int WTHREADS = 64 * WWAVES;
constexpr int WSTAGE = 8;  // quads per stage = 32 k
constexpr float kLog2PiW = 1.8378770664093453f;
constexpr int kThinQuads = 1280;  // LDS quads for a thin band: Kq x rows (20 KB)
constexpr int kWgradCUs = 256;    // MI355X

// PEER: the gradient goes into this rank's exchange buffer (peer.h) with system-scope write-through stores and the launch ends with the
// completion signal to the peers; the clip's sums of squares are then those of the REDUCED gradient and are not computed here.
template <bool PEER>
__device__ __forceinline__ void gstore(float* p, float v) {
  if (PEER) sys_store_f32(p, v);
  else *p = v;
}

// RING: stages in flight per wave.  5 = the whole K range of a wave at mb = 1280 (40 quads): ONE memory latency, 216 registers, one
// workgroup per CU - right when the launch has no more tiles than the chip has CUs (the headline shape: exactly 256).  2: 64 registers
// of operands in flight, <= 128 registers in all, TWO workgroups per CU hide each other's latency - for launches with more tiles than
// CUs (BASELINE configs[4]: 352 tiles were two rounds of one workgroup per CU).  Same products, same order: bit-identical results.
#ifdef MPPO_EXPERIMENTS
__device__ unsigned g_wgrad_barrier[2];  // arrivals, generation (MPPO_WGRAD_BARRIER; zero-initialised with the code object)
#endif
template <bool BF16, bool PEER = false, int RING = 5>
__global__ void __launch_bounds__(WTHREADS, RING <= 2 ? 4 : 2) wgrad_kernel(WgradArgs a, PeerStep ps) {
  __shared__ float red[WWAVES][WTILE * (WTILE + 1)];
  __shared__ float cred[WWAVES][WTILE];
  __shared__ float s_red[WWAVES];
  const int t = threadIdx.x, lane = t & 63, g = wave_uniform(t >> 6), i = lane & 31, h = lane >> 5;
  const unsigned desc = a.order[blockIdx.x];
  const int pi = desc & 15, mt = (desc >> 4) & 255, nt = (desc >> 12) & 255;
  const int tile = (desc >> 20) & 1 ? 0 : 1;  // only "is this tile 0" matters below (it owns the log_std gradient and the loss scalars)
  const WgradProb p = a.p[pi];
  const int m0 = mt * WTILE, n0 = nt * WTILE;
  const int q0 = g * a.qwave;                        // first quad of this wave
  const int nst = a.qwave / WSTAGE;
  // operand views: all quads of the minibatch; a lane offset past the extent (column outside the operand) reads 0
  // bytes per k-quad: a float network's operands are float4, a bf16 network's row pass stored them rounded (k_fused.hip store_quad):
  // 8 bytes that go into the bf16 MFMA as they are
  constexpr int QB = BF16 ? 8 : 16;
  const BufView bufA = make_buf(p.A, (unsigned)a.Kq * (unsigned)p.lda * (unsigned)QB);
  const BufView bufB = make_buf(p.B, (unsigned)a.Kq * (unsigned)p.ldb * (unsigned)QB);
  constexpr int kOutOfRange = 0x40000000;
  // bf16 network (round 5): a lane takes a column PAIR of one quad - 16 contiguous bytes (the quad rows are in plain column order, 8 bytes per
  // column) - instead of one 8-byte quad: half the vector-memory instructions for the same bytes (every such instruction costs the CU's
  // address path ~16 cycles whatever it carries: 40 per wave were 2.1 us of an 8.2 us launch).  Lane (p16 = lane & 15,
  // qg = lane >> 4) holds columns 2 p16, 2 p16 + 1 of quad qg of a group of four; that is the operand layout of v_mfma_f32_16x16x16_bf16
  // (lane (i, kq) supplies k = 4 kq + c of row / column i), so the 32 x 32 tile is computed as 2 x 2 interleaved 16 x 16 sub-tiles:
  // sub-tile (ea, eb) = rows 2 m + ea, columns 2 n + eb.
  const int p16 = lane & 15, qg = lane >> 4;
  const int ca = BF16 ? m0 + 2 * p16 : m0 + i, cb = BF16 ? n0 + 2 * p16 : n0 + i;  // this lane's column (bf16: the even one of its pair)
  const int pa = p.a_split ? (ca >> 1) + (ca & 1) * (p.lda >> 1) : ca, pb = p.b_split ? (cb >> 1) + (cb & 1) * (p.ldb >> 1) : cb;  // (float networks: WgradProb::a_split)
  const int offa = ca < p.acols ? ((BF16 ? qg : h) * p.lda + pa) * QB : kOutOfRange;
  const int offb = cb < p.bcols ? ((BF16 ? qg : h) * p.ldb + pb) * QB : kOutOfRange;
  auto ldq = [&](const BufView& b, int lane_off, int uni_off) {  // one quad; bf16: raw bits in .x, .y
    if (BF16) { const float2 r = buf_load_f2(b, lane_off, uni_off); return make_float4(r.x, r.y, 0.f, 0.f); }
    return buf_load_f4(b, lane_off, uni_off);
  };
  // tile 0 also owns the log_std gradient and the loss scalars: first level of the column sums of the row pass's partials
  // [nblk][4+AP], requested NOW so that their latency hides under the K loop (at the end they were the launch's long pole)
  __shared__ float s_part[16][40];
  __shared__ float s_col[40];
  if (tile == 0 && t < 256) {
    const int W = 4 + a.AP;
    for (int c = t & 15; c < W; c += 16) {
      float s0 = 0.f;
      for (int k = t >> 4; k < a.nblk; k += 16) s0 += a.partial[(size_t)k * W + c];
      s_part[t >> 4][c] = s0;
    }
  }
  float sum_log_std = 0.f;
  if (tile == 0 && t == 0) for (int k = 0; k < a.A; ++k) sum_log_std += a.log_std[k];
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x4 acc16[2][2];  // bf16 network: [ea][eb] sub-tiles
  for (int r = 0; r < 4; ++r) acc16[0][0][r] = acc16[0][1][r] = acc16[1][0][r] = acc16[1][1][r] = 0.f;
  float colsum = 0.f, colsum1 = 0.f;  // bias gradient: sum over k of B(k, n); tiles of the first row band only (bf16: the lane's even / odd column)
  const bool do_colsum = p.off_b >= 0 && mt == 0;
  // (RING = 5: measured - a ring of 4 costs 3 us at the headline shape)
  float4 ra[RING][4], rb[RING][4];
  // a thin band's own operand (its <= 4 rows of A over all K) is tiny: staged in LDS once, read as broadcasts
  __shared__ float4 sx[kThinQuads];
  const bool do_thin = mt == 0 && p.thin_rows > 0;  // uniform
  float tacc[4] = {0.f, 0.f, 0.f, 0.f}, tacc1[4] = {0.f, 0.f, 0.f, 0.f};  // (tacc1: a bf16 network's odd column)
  constexpr int kThinPerThread = (kThinQuads + WTHREADS - 1) / WTHREADS;
  float4 sxr[kThinPerThread];
  if (do_thin) {  // requested first, stored to LDS after the prologue's loads have been requested too (one latency, not two)
#pragma unroll
    for (int j = 0; j < kThinPerThread; ++j) {
      const int e = t + WTHREADS * j, rr = e / a.Kq, qd = e - rr * a.Kq;
      sxr[j] = ldq(bufA, e < a.Kq * p.thin_rows ? (qd * p.lda + p.thin_row0 + rr) * QB : kOutOfRange, 0);  // (per-lane offsets: the scalar one must be wave-uniform)
      if (BF16) sxr[j] = bf16x4_unpack(sxr[j].x, sxr[j].y);
    }
  }
  auto load = [&](int slot, int s) {  // stage s: quads q0 + 8 s + 2 c + h, c = 0..3  (bf16: q0 + 8 s + 4 c + qg, c = 0, 1: a column pair each)
    if (BF16) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int qu = q0 + WSTAGE * s + 4 * c;  // uniform
        ra[slot][c] = buf_load_f4(bufA, offa, qu * p.lda * QB);
        rb[slot][c] = buf_load_f4(bufB, offb, qu * p.ldb * QB);
      }
      return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int qu = q0 + WSTAGE * s + 2 * c;  // uniform
      ra[slot][c] = ldq(bufA, offa, qu * p.lda * QB);
      rb[slot][c] = ldq(bufB, offb, qu * p.ldb * QB);
    }
  };
  auto mm = [&](int slot, int s) {
    if (BF16) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float4 x = ra[slot][c], yraw = rb[slot][c];  // .xy: the even column's quad, .zw: the odd column's (raw bf16 bits)
        const float4 y0 = bf16x4_unpack(yraw.x, yraw.y), y1 = bf16x4_unpack(yraw.z, yraw.w);
        colsum += (y0.x + y0.y) + (y0.z + y0.w);
        colsum1 += (y1.x + y1.y) + (y1.z + y1.w);
        if (do_thin) {
          const int qd = q0 + WSTAGE * s + 4 * c + qg;  // this lane's quad (past the last quad: y is 0, the index is clamped)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            if (rr < p.thin_rows) {
              const float4 z = sx[rr * a.Kq + (qd < a.Kq ? qd : a.Kq - 1)];
              tacc[rr] += (z.x * y0.x + z.y * y0.y) + (z.z * y0.z + z.w * y0.w);
              tacc1[rr] += (z.x * y1.x + z.y * y1.y) + (z.z * y1.z + z.w * y1.w);
            }
        }
        const bf16x4 a0 = bf16x4_from_bits(x.x, x.y), a1 = bf16x4_from_bits(x.z, x.w), b0 = bf16x4_from_bits(yraw.x, yraw.y), b1 = bf16x4_from_bits(yraw.z, yraw.w);
        mfma_bf16_16x16x16(a0, b0, acc16[0][0]); mfma_bf16_16x16x16(a0, b1, acc16[0][1]);
        mfma_bf16_16x16x16(a1, b0, acc16[1][0]); mfma_bf16_16x16x16(a1, b1, acc16[1][1]);
      }
      return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 x = ra[slot][c], yraw = rb[slot][c];
      const float4 y = BF16 ? bf16x4_unpack(yraw.x, yraw.y) : yraw;  // (values: bias sum and thin band; the bf16 MFMA takes the raw bits)
      colsum += (y.x + y.y) + (y.z + y.w);
      if (do_thin) {
        const int qd = q0 + WSTAGE * s + 2 * c + h;  // this lane's quad (past the last quad: y is 0, the index is clamped)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          if (rr < p.thin_rows) { const float4 z = sx[rr * a.Kq + (qd < a.Kq ? qd : a.Kq - 1)]; tacc[rr] += (z.x * y.x + z.y * y.y) + (z.z * y.z + z.w * y.w); }
      }
      if (BF16) {
        // v_mfma_f32_32x32x8_bf16: lane (i, h) supplies k = 4h + c: exactly its quad
        mfma_bf16_32x32x8(bf16x4_from_bits(x.x, x.y), bf16x4_from_bits(yraw.x, yraw.y), acc);
      } else {
        // v_mfma_f32_32x32x2_f32: lane (i, h) supplies one k per MFMA: the four k's of its quad, one after the other
        mfma_f32_32x32x2(x.x, y.x, acc); mfma_f32_32x32x2(x.y, y.y, acc);
        mfma_f32_32x32x2(x.z, y.z, acc); mfma_f32_32x32x2(x.w, y.w, acc);
      }
    }
  };
  // stages past the end are clamped to the last one (a redundant load, never multiplied): straight-line prologue
#pragma unroll
  for (int d = 0; d < RING; ++d) load(d, d < nst ? d : nst - 1);
  if (do_thin) {  // uniform per workgroup
#pragma unroll
    for (int j = 0; j < kThinPerThread; ++j)
      if (t + WTHREADS * j < kThinQuads) sx[t + WTHREADS * j] = sxr[j];
    __syncthreads();
  }
  for (int s = 0; s < nst; s += RING) {
#pragma unroll
    for (int u = 0; u < RING; ++u) {
      if (s + u < nst) mm(u, s + u);
      if (s + u + RING < nst) load(u, s + u + RING);
    }
  }
  __shared__ float tred[WWAVES][4][WTILE];
  if (do_thin) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      if (BF16) {  // the four 16-lane groups hold different quads of the same column pair
        tacc[rr] += __shfl_xor(tacc[rr], 16); tacc[rr] += __shfl_xor(tacc[rr], 32);
        tacc1[rr] += __shfl_xor(tacc1[rr], 16); tacc1[rr] += __shfl_xor(tacc1[rr], 32);
        if (qg == 0) { tred[g][rr][2 * p16] = tacc[rr]; tred[g][rr][2 * p16 + 1] = tacc1[rr]; }
      } else {
        tacc[rr] += __shfl_xor(tacc[rr], 32);
        if (h == 0) tred[g][rr][i] = tacc[rr];
      }
    }
  }
  // ---- sum the eight K ranges in wave order ----
  if (BF16) {
#pragma unroll
    for (int ea = 0; ea < 2; ++ea)
#pragma unroll
      for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[g][(2 * (4 * qg + r) + ea) * (WTILE + 1) + 2 * p16 + eb] = acc16[ea][eb][r];
    colsum += __shfl_xor(colsum, 16); colsum += __shfl_xor(colsum, 32);
    colsum1 += __shfl_xor(colsum1, 16); colsum1 += __shfl_xor(colsum1, 32);
    if (qg == 0) { cred[g][2 * p16] = colsum; cred[g][2 * p16 + 1] = colsum1; }
  } else {
#pragma unroll
  for (int r = 0; r < 16; ++r) red[g][((r & 3) + 8 * (r >> 2) + 4 * h) * (WTILE + 1) + i] = acc[r];
  colsum += __shfl_xor(colsum, 32);  // the two lane halves hold different quads
  if (h == 0) cred[g][i] = colsum;
  }
  __syncthreads();
  float sq = 0.f;
  if (PEER && (p.N & 3) == 0 && (p.off_w & 3) == 0) {
    // system-scope stores are one fabric write each: four columns of a row per thread and ONE 16-byte store (a dword store moves a
    // quarter of the bytes for the same cost, MI355X_MICROARCH.md "stores of each flavour"); same sums, same order per element
    if (t < WTILE * WTILE / 4) {
      const int row = t >> 3, col = (t & 7) * 4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < WWAVES; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += red[k][row * (WTILE + 1) + col + c];
      if (m0 + row < p.M && n0 + col < p.N)  // (N a multiple of four: the quad is inside or outside as a whole)
        sys_store_f4(a.grad, ((size_t)p.off_w + (size_t)(m0 + row) * p.N + n0 + col) * 4, make_float4(v[0], v[1], v[2], v[3]));
    }
  } else {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = t + WTHREADS * e, row = idx >> 5, col = idx & 31;
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < WWAVES; ++k) v += red[k][row * (WTILE + 1) + col];
      if (m0 + row < p.M && n0 + col < p.N) {
        gstore<PEER>(&a.grad[p.off_w + (size_t)(m0 + row) * p.N + n0 + col], v);
        sq += v * v;
      }
    }
  }
  if (do_thin && t < 4 * WTILE) {
    const int rr = t >> 5, col = n0 + (t & 31);
    if (rr < p.thin_rows && col < p.N) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < WWAVES; ++k) v += tred[k][rr][t & 31];
      gstore<PEER>(&a.grad[p.off_w + (size_t)(p.thin_row0 + rr) * p.N + col], v);
      sq += v * v;
    }
  }
  if (do_colsum && t < WTILE && n0 + t < p.N) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < WWAVES; ++k) v += cred[k][t];
    gstore<PEER>(&a.grad[p.off_b + n0 + t], v);
    sq += v * v;
  }
  // ---- tile 0: second level of the partial sums (s_part was written before the K loop; the barrier above orders it) ----
  if (tile == 0) {
    const int W = 4 + a.AP;
    if (t < W) {
      float s0 = 0.f;
      for (int gq = 0; gq < 16; ++gq) s0 += s_part[gq][t];
      s_col[t] = s0;
    }
    __syncthreads();
    if (t < a.A) {
      const float d = s_col[4 + t] - a.ent_coef * a.ent_weight;
      gstore<PEER>(&a.grad[a.ls_off + t], d);
      sq += d * d;
    }
    if (t == 0 && a.loss4) {
      const float ent = (0.5f * (float)a.A * (1.f + kLog2PiW) + sum_log_std) * a.ent_weight;
      a.loss4[0] = s_col[0] + a.vf_coef * s_col[1] - a.ent_coef * ent;
      a.loss4[1] = s_col[1];
      a.loss4[2] = s_col[0];
      a.loss4[3] = ent;
    }
    if (t < a.npad) for (int k = 0; k < a.pad_cnt[t]; ++k) gstore<PEER>(&a.grad[a.pad_off[t] + k], 0.f);  // alignment words of the flat layout
    if (a.sq_partial)  // slots of workgroups that do not exist
      for (int e = a.ntiles + t; e < kSqSlots; e += WTHREADS) a.sq_partial[e] = 0.f;  // (slots are indexed by workgroup id)
  }
  // ---- the tile's sum of squares (input of clip_by_global_norm, train.py:117) ----
  sq = wave_sum(sq);
  if (lane == 0) s_red[g] = sq;
  __syncthreads();
  if (t == 0 && a.sq_partial) {
    float s = 0.f;
    for (int k = 0; k < WWAVES; ++k) s += s_red[k];
    a.sq_partial[blockIdx.x] = s;
  }
  if (PEER) peer_publish_done(ps.v, ps.epoch[0] + ps.step + 1, desc >> 24, a.slice_need);  // every store of the gradient is above this line
#ifdef MPPO_EXPERIMENTS
  // measurement (MPPO_WGRAD_BARRIER=1): what a device-wide barrier costs at the END of this launch - where clip + Adam would continue if the two
  // launches were one (every workgroup of the launch is resident: one per CU).  Arrive on one counter, the last one opens the next generation.
  if (!PEER && ps.mode == -12345) {
    __syncthreads();
    if (t == 0) {
      __threadfence();
      const unsigned gen = __hip_atomic_load(&g_wgrad_barrier[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(&g_wgrad_barrier[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        __hip_atomic_store(&g_wgrad_barrier[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&g_wgrad_barrier[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(&g_wgrad_barrier[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    // (what would follow: every workgroup re-adds the partial sums of squares - one load of 512 floats - and updates its tile)
    if (t < 64 && a.sq_partial) { float s = 0.f; for (int e = t; e < kSqSlots; e += 64) s += __hip_atomic_load(&a.sq_partial[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (s == -1.f) a.grad[0] = s; }
  }
#endif
}

bool wgrad_supported(const WgradArgs& a) {
  if (a.count < 1 || a.count > kWgradMaxProb || a.ntiles < 1 || a.ntiles > kSqSlots || a.AP + 4 > 40) return false;
  for (int k = 0; k < a.count; ++k) {
    const WgradProb& p = a.p[k];
    if (p.tiles_m > 256 || p.tiles_n > 256) return false;  // (8 bits each in WgradArgs::order)
    if (p.a_split && p.thin_rows > 0) return false;          // the thin band's staging load reads A in plain column order
    if ((reinterpret_cast<uintptr_t>(p.A) & 15) || (reinterpret_cast<uintptr_t>(p.B) & 15) || p.acols < p.M || p.bcols < p.N || p.acols > p.lda || p.bcols > p.ldb)
      return false;
  }
  return true;
}

int32_t wgrad_plan(WgradArgs& a, int mb) {
  int tiles = 0;
  for (int k = 0; k < a.count; ++k) {
    WgradProb& p = a.p[k];
    p.tiles_m = cdiv(p.M, WTILE); p.tiles_n = cdiv(p.N, WTILE); p.tile0 = tiles;
    p.thin_row0 = 0; p.thin_rows = 0;
    const int rem = p.M - (p.tiles_m - 1) * WTILE;
    if (p.tiles_m > 1 && rem <= 4 && (int)(pad16((size_t)mb) / 4) * rem <= kThinQuads) {  // a last row band of at most 4 rows is THIN: contracted by the first band's workgroups
      p.tiles_m -= 1;
      p.thin_row0 = p.tiles_m * WTILE; p.thin_rows = rem;
    }
    tiles += p.tiles_m * p.tiles_n;
  }
  a.ntiles = tiles;
  if (tiles > kSqSlots) return MPPO_OK;  // wgrad_supported says no
  // Workgroup -> tile.  Workgroup ids that are equal modulo 8 run on the same XCD (round-robin dispatch) and share its L2.
  // A tile reads a 32-column band of A and one of B over all K rows; every band a XCD touches comes from memory once.
  // Big problems (>= 16 tiles) are split by row bands over consecutive XCD pairs, so that a XCD holds tm/2 bands of A and
  // all bands of B of ONE problem (12 bands = 2 MB at H = 256) instead of bands of every problem; small ones go round.
  {
    std::vector<std::vector<int>> queue(8);
    std::vector<int> bigidx(a.count, -1);
    int big = 0, small = 0;
    for (int k = 0; k < a.count; ++k)
      if (a.p[k].tiles_m * a.p[k].tiles_n >= 16) bigidx[k] = big++;
    for (int pass = 0; pass < 2; ++pass) {  // big problems first; the small ones' tiles then level the XCDs' queues
      for (int k = 0; k < a.count; ++k) {
        const WgradProb& p = a.p[k];
        const int n = p.tiles_m * p.tiles_n;
        if ((bigidx[k] >= 0) != (pass == 0)) continue;
        for (int t = 0; t < n; ++t) {
          int x;
          if (pass == 0) {
            x = (2 * bigidx[k] + ((t / p.tiles_n) * 2 >= p.tiles_m ? 1 : 0)) & 7;
          } else {
            x = 0;
            for (int y = 1; y < 8; ++y) if (queue[y].size() < queue[x].size()) x = y;
          }
          queue[x].push_back(p.tile0 + t);
        }
      }
    }
    (void)small;
    std::vector<int> assigned(tiles, -1), free_wg;
    std::vector<size_t> pos(8, 0);
    for (int w = 0; w < tiles; ++w) {
      const int x = w & 7;
      if (pos[x] < queue[x].size()) assigned[w] = queue[x][pos[x]++];
      else free_wg.push_back(w);
    }
    size_t f = 0;
    for (int x = 0; x < 8; ++x)
      while (pos[x] < queue[x].size()) assigned[free_wg[f++]] = queue[x][pos[x]++];
    for (int w = 0; w < tiles; ++w) {
      const int tile = assigned[w];
      int pi = 0;
      while (pi + 1 < a.count && tile >= a.p[pi + 1].tile0) ++pi;
      const int lt = tile - a.p[pi].tile0, mt = lt / a.p[pi].tiles_n, nt = lt - mt * a.p[pi].tiles_n;
      a.order[w] = (unsigned)pi | ((unsigned)mt << 4) | ((unsigned)nt << 12) | (tile == 0 ? 1u << 20 : 0u);
    }
  }
  a.Kq = (int)(pad16((size_t)mb) / 4);                      // the row pass writes whole 16-row tiles (zeros past mb)
  a.qwave = cdiv(cdiv(a.Kq, WWAVES), WSTAGE) * WSTAGE;      // quads per wave, whole stages
  return MPPO_OK;
}

// Which slices of the flat gradient (slice q = float4 [q S4, (q + 1) S4), reduced by rank q: peer.h) does each workgroup store into?  Mirrors the
// kernel's stores: the 32 x 32 tile, a thin band's rows and the bias sums (first row band's workgroups), tile 0's log_std gradient and
// alignment words.  Result: the slice masks in bits 24..31 of order[], the number of workgroups per slice in slice_need[].
static int32_t wgrad_peer_slices(WgradArgs& a, const PeerView& v) {
  static_assert(kPeerMaxRanks == 8, "order[] keeps eight mask bits");
  unsigned need[kPeerMaxRanks] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long S = 4L * v.S4;  // floats per slice
  for (int w = 0; w < a.ntiles; ++w) {
    const unsigned desc = a.order[w] & 0x00FFFFFFu;
    const int pi = desc & 15, mt = (desc >> 4) & 255, nt = (desc >> 12) & 255;
    const bool tile0 = (desc >> 20) & 1;
    const WgradProb& p = a.p[pi];
    const int m0 = mt * WTILE, n0 = nt * WTILE, nc = (p.N - n0 < WTILE ? p.N - n0 : WTILE);
    unsigned mask = 0;
    auto touch = [&](long first, long count) {  // floats [first, first + count) of the flat gradient
      if (count <= 0) return;
      for (long q = first / S; q <= (first + count - 1) / S; ++q) mask |= 1u << (q < v.world ? q : v.world - 1);
    };
    for (int r = m0; r < m0 + WTILE && r < p.M && (p.thin_rows == 0 || r < p.thin_row0); ++r) touch((long)p.off_w + (long)r * p.N + n0, nc);
    if (mt == 0 && p.thin_rows > 0)
      for (int rr = 0; rr < p.thin_rows; ++rr) touch((long)p.off_w + (long)(p.thin_row0 + rr) * p.N + n0, nc);
    if (p.off_b >= 0 && mt == 0) touch((long)p.off_b + n0, nc);
    if (tile0) {
      touch(a.ls_off, a.A);
      for (int k = 0; k < a.npad; ++k) touch(a.pad_off[k], a.pad_cnt[k]);
    }
    a.order[w] = desc | (mask << 24);
    for (int q = 0; q < v.world; ++q) need[q] += (mask >> q) & 1u;
  }
  for (int q = 0; q < v.world; ++q) {
    if (need[q] == 0) {  // (a slice nobody stores into - more ranks than float4 rows: its owner still waits for the flag) workgroup 0 speaks for it
      a.order[0] |= 1u << (24 + q);
      need[q] = 1;
    }
    MPPO_REQUIRE(need[q] <= 0xFFFFu, "wgrad_launch: %u workgroups store into slice %d", need[q], q);
    a.slice_need[q] = (unsigned short)need[q];
  }
  return MPPO_OK;
}

int32_t wgrad_launch(const WgradArgs& a_in, bool bf16, hipStream_t stream, const PeerStep* peer) {
  WgradArgs a = a_in;
  const bool shallow = a.ntiles > kWgradCUs;  // more tiles than CUs: two lighter workgroups per CU instead of two rounds of one
  if (peer) {
    MPPO_REQUIRE(!a.sq_partial, "wgrad_launch: with the peer exchange the sums of squares are those of the reduced gradient (sq_partial must be null)");
    MPPO_REQUIRE(wgrad_supported(a), "wgrad_launch: operands must be 16-byte aligned k-quad buffers, at most %d tiles", kSqSlots);
    MPPO_TRY(wgrad_peer_slices(a, peer->v));
    if (shallow) {
      if (bf16) hipLaunchKernelGGL((wgrad_kernel<true, true, 2>), dim3(a.ntiles), dim3(WTHREADS), 0, stream, a, *peer);
      else hipLaunchKernelGGL((wgrad_kernel<false, true, 2>), dim3(a.ntiles), dim3(WTHREADS), 0, stream, a, *peer);
    } else {
      if (bf16) hipLaunchKernelGGL((wgrad_kernel<true, true>), dim3(a.ntiles), dim3(WTHREADS), 0, stream, a, *peer);
      else hipLaunchKernelGGL((wgrad_kernel<false, true>), dim3(a.ntiles), dim3(WTHREADS), 0, stream, a, *peer);
    }
    MPPO_CHECK_LAUNCH("wgrad_kernel<peer>");
    return MPPO_OK;
  }
  PeerStep nops{};
#ifdef MPPO_EXPERIMENTS
  static const int bar_on = [] { const char* e = getenv("MPPO_WGRAD_BARRIER"); return e && e[0] == '1' ? 1 : 0; }();
  if (bar_on) nops.mode = -12345;  // (the switch rides in a field the launch without a peer exchange does not read)
  static const int dbg = [] { const char* e = getenv("MPPO_WGRAD_DBG"); return e ? atoi(e) : 0; }();
#else
  constexpr int dbg = 0;
#endif
  a.dbg = dbg;
  MPPO_REQUIRE(wgrad_supported(a), "wgrad_launch: operands must be 16-byte aligned k-quad buffers, at most %d tiles", kSqSlots);
  if (shallow) {
    if (bf16) hipLaunchKernelGGL((wgrad_kernel<true, false, 2>), dim3(a.ntiles), dim3(WTHREADS), 0, stream, a, nops);
    else hipLaunchKernelGGL((wgrad_kernel<false, false, 2>), dim3(a.ntiles), dim3(WTHREADS), 0, stream, a, nops);
  } else if (bf16) hipLaunchKernelGGL((wgrad_kernel<true, false>), dim3(a.ntiles), dim3(WTHREADS), 0, stream, a, nops);
  else hipLaunchKernelGGL((wgrad_kernel<false, false>), dim3(a.ntiles), dim3(WTHREADS), 0, stream, a, nops);
  if (dbg & 4) {  // timing experiment: the same launch again, operands now warm in the L2s
    if (bf16) hipLaunchKernelGGL((wgrad_kernel<true, false>), dim3(a.ntiles), dim3(WTHREADS), 0, stream, a, nops);
    else hipLaunchKernelGGL((wgrad_kernel<false, false>), dim3(a.ntiles), dim3(WTHREADS), 0, stream, a, nops);
  }
  MPPO_CHECK_LAUNCH("wgrad_kernel");
  return MPPO_OK;
}

}  // namespace mppo
