// fused_bf16.h — the training row pass of a bf16 network (BASELINE configs[3]), designed for bf16 instead of being the float kernel
// with cheaper MFMAs.  Included by k_fused.hip (same translation unit: FusedArgs, the k-quad store helpers, the gather role and the
// phase timers are shared); launched by fused_forward_backward for <bf16 network, W2^T / fragment shadow copies current, pre-gathered rows>.
//
// Same decomposition as fused_mlp_kernel - one workgroup = 16 minibatch rows of ONE network, H / 32 waves, wave w owns hidden columns
// [32 w, 32 w + 32) as two interleaved 16-column tiles - and the same arithmetic, rounding point for rounding point (oracle:
// oracle/ppo_oracle.py loss_and_grad(bf16=True): bf16 operands in x.W1, h1.W2 and dZ2.W2^T, exact float output layer and dOut.W3^T,
// activation derivatives from the float activations).  What differs is where the time went (profiles/r05_a_fused_phases_bf16_before.txt:
// the float kernel's structure spends 4.2 of its 11.1 us in-kernel streaming three 128 KB weight slabs through a three-stage ring, phase
// after phase, with < 1 us of matrix work; DESIGN.md 3.1b has the measurements this file is built on):
//   * A wave's slab of a WHOLE layer is 16 KB of bf16 fragments = 64 VGPRs (k_ppo.hip keeps the fragment-order copies current).  W1 is
//     requested at kernel entry, W2 at the start of P0 / layer 1, W2^T at the start of layer 2 / the heads - each in front of the
//     phase's own work, a phase or two ahead of its use.  No ring, no clamped re-loads, no per-stage waits: a GEMM phase is 16 LDS reads
//     (all of them first) and 16 v_mfma_f32_16x16x32_bf16 per wave on four accumulation chains.
//   * The 16-row activation tiles that feed the matrix cores (x, h1, dZ2) live in LDS ALREADY ROUNDED to bf16 (what the float kernel's
//     pack at every read produced): 8-byte ds_read_b64 per four k instead of 16-byte reads + four conversions, row stride 4 (mod 64)
//     dwords = conflict-free.  h2 stays float in LDS (A operand of the exact head product); the activations a lane needs again for its
//     derivative (its own 4 rows x 2 columns of h1, h2) stay in its registers.
//   * A wave's 32 rows of W3 arrive as two coalesced loads and are re-arranged through a wave-local LDS scratch (every vector-memory
//     instruction costs the CU's address path ~16 cycles whatever it carries: sixteen 4-byte gathers per lane cost as much as 16 KB of
//     fragments); the wave index lives in a scalar register (else: a waterfall loop around every fragment load).
//   * The loss block runs on wave 0 only (the emulator: every wave), its 16-row partial sums are 16-lane DPP sums instead of one
//     thread's serial loop behind a barrier, and the x tile is brought in by the first half of the waves.
#pragma once

// two floats -> the bits of two bf16 (round to nearest even) in one dword
__device__ __forceinline__ float bf16x2_bits(float a, float b) { return bf16x4_bits(pack_bf16x4(a, b, 0.f, 0.f)).x; }

// A wave's fragments of one whole [K][H] weight: NS stages of 32 k, two 16-byte registers per stage (ppo_layout.h frag_index: register
// tau of a stage holds the lane's eight k's of column tile tau - k = 16 g + 4 (lane >> 4) + c, g = 0, 1, c = 0 .. 3 - in that order).
// Stages past the matrix (K < 32 NS) lie outside the buffer view and read as zeros: branch-free, no memory traffic.
// One v_mfma_f32_16x16x32_bf16 per tile and stage: the instruction's k slot 8 (lane >> 4) + (4 g + c) carries k = 16 g + 4 (lane >> 4) + c
// of the stage for A and B alike (a sum over k does not care which slot a k sits in), so the A operand is the lane's two 8-byte LDS reads
// (g = 0, 1) side by side and the B operand the fragment register as it was loaded.
template <int NS>
struct FragSlab {
  float4 q[NS][2];
  BufView wb;
  int lane_off, stage_stride, wave_off;
  __device__ __forceinline__ void open(const unsigned short* frag, int K, int H, int wave, int lane) {
    wb = make_buf(reinterpret_cast<const float*>(frag), (unsigned)K * (unsigned)H * 2u);
    lane_off = lane * (2 * kFragLaneElems); stage_stride = (H >> 5) * 2048; wave_off = wave * 2048;  // bytes: a (stage, wave slab) block is 2 KB
  }
  __device__ __forceinline__ void load_one(int S, int tau) { q[S][tau] = buf_load_f4(wb, lane_off + tau * 2 * kFragTileElems, S * stage_stride + wave_off); }
  __device__ __forceinline__ void load_stage(int S) { load_one(S, 0); load_one(S, 1); }
  // The lane's A operands of ALL stages of a tile (bf16 in LDS; `arow` = the lane's row at its k offset 4 (lane >> 4)), requested together:
  // read stage by stage in front of its two MFMAs, every stage paid a full LDS round trip (1.2 k cycles for a product whose 16 MFMAs
  // take 256).  EXACT: the tile has exactly NS stages; otherwise stages >= nst are replaced by zeros (a select, not a branch: what lies
  // behind the tile's K columns may be anything, and the fragments there are zeros).
  struct ATile { float4 a[NS]; };
  template <bool EXACT>
  __device__ __forceinline__ ATile read_a(const unsigned short* arow, int nst) const {
    ATile r;
#pragma unroll
    for (int S = 0; S < NS; ++S) {
      float2 a0 = *reinterpret_cast<const float2*>(arow + 32 * S), a1 = *reinterpret_cast<const float2*>(arow + 32 * S + 16);
      if (!EXACT && S >= nst) { a0 = make_float2(0.f, 0.f); a1 = a0; }
      r.a[S] = make_float4(a0.x, a0.y, a1.x, a1.y);
    }
    return r;
  }
  // stage S: even and odd stages accumulate into chains of their own (acc[S & 1][tile]): four independent MFMA chains per wave
  __device__ __forceinline__ void mfma_stage(int S, const ATile& A, f32x4 (&acc)[2][2]) const {
    mfma_bf16_16x16x32(A.a[S], q[S][0], acc[S & 1][0]);
    mfma_bf16_16x16x32(A.a[S], q[S][1], acc[S & 1][1]);
  }
};

// NS1: 32-k stages of the first layer's registers (KP <= 32 NS1); the hidden layers have H / 32 <= 8
// EXACT: the geometry fills the registers exactly (KP = 32 NS1, H = 256): no zero stages, no selects - the headline shape; the other
// instantiation runs every supported geometry (observations up to 32 NS1 wide, H = 32 .. 256)
template <int OT, int NS1, bool EXACT>
__global__ void __launch_bounds__(512, 2) bf16_rowpass_kernel(FusedArgs a) {
  constexpr int SD = 16 * OT, NSH = 8;
  constexpr bool ROLLOUT = false;  // (the phase-timer macros of k_fused.hip test it)
  (void)ROLLOUT;
  FT(0);
  FTW(56);
  if (blockIdx.y >= 2) {  // grid rows 2, 3: gather the NEXT step's rows (uniform per workgroup; see fused_mlp_kernel)
    gather_rows_tile<true>(a, (int)blockIdx.x, (int)blockIdx.y - 2);
    return;
  }
  MPPO_DYN_SMEM(smem_raw);
  const int H = a.H, O = a.O, OP = a.OP, A = a.A, AP = a.AP;
  const int net = blockIdx.y;  // 0 actor, 1 critic
  const int KP = (O + 31) & ~31;
  const int XSB = KP + 8, HSB = H + 8, HS = H + 4;  // bf16 row strides (elements): 4 (mod 64) dwords; float row stride of h2
  const int t = threadIdx.x, nthr = blockDim.x, lane = t & 63, nw = nthr >> 6;
  const int wave = wave_uniform(t >> 6);  // (in a scalar register: a buffer load whose scalar offset the compiler cannot prove uniform is wrapped in a waterfall loop)
  unsigned short* xb = reinterpret_cast<unsigned short*>(smem_raw);  // [16][XSB] bf16   x tile
  unsigned short* h1b = xb + FRT * XSB;                             // [16][HSB] bf16   h1
  unsigned short* dzb = h1b + FRT * HSB;                            // [16][HSB] bf16   dZ2
  float* h2t = reinterpret_cast<float*>(dzb + FRT * HSB);           // [16][HS]  float  h2 (A operand of the exact head product)
  float* s_hp = h2t + FRT * HS;                                     // [nw][OT][64][4]  partial head tiles
  float* s_do = s_hp + nw * OT * 256;                               // [16][SD]  d mean | d value
  float* s_red = s_do + FRT * SD;                                   // [16][SD]  d log_std terms
  float* s_l = s_red + FRT * SD;                                    // [16]      per-row loss term
  const int row0 = blockIdx.x * FRT;
  const bool tanh_act = net == 0 && a.use_tanh;
  const float* B1 = a.params + (net ? a.L.c_b1 : a.L.a_b1);
  const float* B2 = a.params + (net ? a.L.c_b2 : a.L.a_b2);
  const float* W3 = a.params + (net ? a.L.c_w3 : a.L.a_w3);
  const float* B3 = a.params + (net ? a.L.c_b3 : a.L.a_b3);
  const int nout = net ? 1 : A;
  const int n0 = 32 * wave, cj = lane & 15, rq = lane >> 4;
  const int nst1 = KP >> 5, nsth = H >> 5;

  // ---- requests.  The 384 KB of weight fragments a workgroup needs are what bounds this kernel (a CU takes in ~100 GB/s from its XCD's
  // L2: 3.8 us), so the stream must run from the first microsecond to the last GEMM WITHOUT the waves standing in the issue queue while
  // they could compute: a vector-memory instruction issues only when the CU's address path has room, and with eight waves requesting
  // 1 KB per instruction that is one instruction per ~130 cycles and wave.  W1 (needed first) is requested here, all of it; the 16
  // loads of W2 are requested at the start of P0 (4) and P1 (12), those of W2^T at the start of P2 (12) and P3 (4) - each a phase or two
  // ahead of its use, in front of the phase's own work, which then runs while they are served.  (Measured and dropped, DESIGN.md 9: the same
  // loads one at a time between blocks of work; the two waves of a SIMD requesting at opposite ends of a phase - 9.5 us either way.)  
  // Every load of the common path is unconditional on a clamped address (a conditional load is a basic block of its own, and the s_waitcnt
  // insertion then assumes the worse of two histories at the join: the wait for the x tile waited for W1 as well); the two wave-uniform
  // exceptions - the x tile's loads (first half of the waves) and the loss block's inputs (wave 0) - sit in front of / behind W1's loads.
  const int nxq = 4 * OP;
  const float* xtile = a.xpre + (size_t)(row0 >> 2) * OP * 2;  // this step's rows: a contiguous block of 8-byte bf16 quads (store_quad<true>)
  // brought in by the FIRST HALF of the waves, four quads per lane: the address path serves the older wave of every SIMD first, so waves
  // nw/2 .. get through their requests ~2 k cycles after waves 0 .. nw/2 - 1 (stamps: profiles/r05_k_fused_phases_bf16.txt) - by the time
  // they would write their share of the tile to LDS the first half has written all of it, and the barrier behind P0 waits for nobody
  const int nwx = (nw + 1) >> 1, nthx = 64 * nwx;
  const bool xw = wave < nwx;
  float2 xr0 = make_float2(0.f, 0.f), xr1 = xr0, xr2 = xr0, xr3 = xr0;
  if (xw) {
    auto xl = [&](int e) { return *reinterpret_cast<const float2*>(xtile + 2 * (e < nxq ? e : nxq - 1)); };
    xr0 = xl(t); xr1 = xl(t + nthx); xr2 = xl(t + 2 * nthx); xr3 = xl(t + 3 * nthx);
  }
  const unsigned short* fr = a.frag[net];
  FragSlab<NS1> w1;
  w1.open(fr, KP, H, wave, lane);
#pragma unroll
  for (int S = 0; S < NS1; ++S) w1.load_stage(S);
  const float2 bz1 = *reinterpret_cast<const float2*>(B1 + n0 + 2 * cj), bz2 = *reinterpret_cast<const float2*>(B2 + n0 + 2 * cj);
  MPPO_SCHED_FENCE();
  FT(1);
  // what the heads and the loss need, phases later.  Every vector-memory instruction of a wave costs the CU's address path ~16 cycles
  // whatever it carries (64 lanes, 4 per clock), and the 8 x 48 weight loads already keep it busy for 2.6 us: sixteen 4-byte gathers per
  // lane for the output-layer weights cost as much as 16 KB of fragments.  So this wave's 32 rows of W3 (one contiguous block of
  // 32 x nout floats) come in as NW3 coalesced 16-byte loads and are re-arranged through LDS (P1); the rows' loss scalars are whole
  // float4s of the pre-gathered quads.  RAW values on clamped addresses; every use is guarded by the row / column conditions (a select
  // HERE would be the value's first use, and the wait goes where the first use is).
  constexpr int NW3 = OT == 1 ? 2 : 4;  // 8 nout float4s per wave: nout <= 16 / 32
  const int nq4 = 8 * nout;
  const float* W3w = W3 + (size_t)n0 * nout;
  auto w3_load = [&](int it) { return *reinterpret_cast<const float4*>(W3w + 4 * (lane + 64 * it < nq4 ? lane + 64 * it : nq4 - 1)); };
  const float4 w3r0 = w3_load(0), w3r1 = w3_load(1), w3r2 = NW3 > 2 ? w3_load(2) : w3r0, w3r3 = NW3 > 2 ? w3_load(3) : w3r0;  // (named: an array went to scratch)
  const float adv_mean = a.adv_stat[0], adv_rstd = a.adv_stat[1];
  float ls[OT], b3v[OT], w3p[OT][8], w3q[4 * OT][2], pf0[OT][4], pf1[4], pf2[4];
#ifdef MPPO_EMU
  constexpr bool kLossAllWaves = true;  // (the emulator's cross-lane shims are workgroup-wide barriers: every wave runs the loss block)
#else
  constexpr bool kLossAllWaves = false;  // on the GPU only wave 0 runs the loss block (see there) - and only it requests the block's inputs
#endif
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) {
    ls[ot] = b3v[ot] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) pf0[ot][r] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) pf1[r] = pf2[r] = 0.f;
  if (kLossAllWaves || wave == 0) {
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) {
    const int o = cj + 16 * ot;
    ls[ot] = a.params[a.L.log_std + (o < A ? o : A - 1)];  // meaningful for the actor's outputs o < A only
    b3v[ot] = B3[o < nout ? o : nout - 1];
  }
  {  // actor: action[o], old log_prob, advantage | critic: old value, target, -
    const int SC = A + 4;
    const float* sq = a.xpre + xquad_obs_floats(OP, a.mb) + ((size_t)((row0 >> 2) + rq) * SC) * 4;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      const int o = cj + 16 * ot;
      const float4 q = *reinterpret_cast<const float4*>(sq + 4 * (net ? A + 2 : (o < A ? o : A - 1)));
      pf0[ot][0] = q.x; pf0[ot][1] = q.y; pf0[ot][2] = q.z; pf0[ot][3] = q.w;
    }
    const float4 q1 = *reinterpret_cast<const float4*>(sq + 4 * (net ? A + 3 : A)), q2 = *reinterpret_cast<const float4*>(sq + 4 * (A + 1));
    pf1[0] = q1.x; pf1[1] = q1.y; pf1[2] = q1.z; pf1[3] = q1.w;
    pf2[0] = q2.x; pf2[1] = q2.y; pf2[2] = q2.z; pf2[3] = q2.w;
  }
  }
  MPPO_SCHED_FENCE();
  FT(2);
  FTW(64);
  FragSlab<NSH> w2, w5;
  w2.open(fr + (size_t)KP * H, H, H, wave, lane);
  w5.open(fr + (size_t)KP * H + (size_t)H * H, H, H, wave, lane);
#define WL(slab, S, tau) do { MPPO_SCHED_FENCE(); slab.load_one(S, tau); MPPO_SCHED_FENCE(); } while (0)
  // ---- P0: x tile -> LDS, row-major bf16.  A k-quad element is one column of four consecutive rows: four 2-byte words a row apart ----
  {
    auto put = [&](int e, const float2& q) {
      const int qd = (int)(e >= OP) + (int)(e >= 2 * OP) + (int)(e >= 3 * OP), c = e - qd * OP;
      const unsigned lo = __float_as_uint(q.x), hi = __float_as_uint(q.y);
      unsigned short* d = xb + (4 * qd) * XSB + c;
      d[0] = (unsigned short)(lo & 0xFFFFu); d[XSB] = (unsigned short)(lo >> 16); d[2 * XSB] = (unsigned short)(hi & 0xFFFFu); d[3 * XSB] = (unsigned short)(hi >> 16);
    };
    WL(w2, 0, 0); WL(w2, 0, 1); WL(w2, 1, 0); WL(w2, 1, 1);
    if (xw) {
      if (t < nxq) put(t, xr0);
      if (t + nthx < nxq) put(t + nthx, xr1);
      if (t + 2 * nthx < nxq) put(t + 2 * nthx, xr2);
      if (t + 3 * nthx < nxq) put(t + 3 * nthx, xr3);
      for (int e = t + 4 * nthx; e < nxq; e += nthx) put(e, *reinterpret_cast<const float2*>(xtile + 2 * e));  // (small workgroups only)
    }
    for (int r = t >> 5; r < FRT; r += nthr >> 5)  // K padding of the first layer: fewer than 32 columns per row, one lane each
      if (OP + (t & 31) < KP) xb[r * XSB + OP + (t & 31)] = 0;
  }
  FTW(72);
  __syncthreads();
  FT(3);
  FTW(80);
  const int c0 = n0 + 2 * cj;  // the lane's two adjacent columns, rows 4 rq .. 4 rq + 3, in every epilogue below
  float h1r[2][4], h2r[2][4];  // the lane's own float activations (for the derivatives in P4 / P5)
  // ---- P1: h1 = act(x . W1 + b1); W2 requested between the stages ----
  {
    f32x4 acc[2][2], acc0, acc1;
    for (int r = 0; r < 4; ++r) { acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f; }
#ifdef MPPO_FUSED_TIMERS
    if (blockIdx.x == 40 && blockIdx.y == 0 && lane == 0) g_fused_t[24 + wave] = __builtin_amdgcn_s_memtime();
#endif
    {
#pragma unroll
      for (int S = 2; S < NSH; ++S) { WL(w2, S, 0); WL(w2, S, 1); }
      const auto At = w1.template read_a<EXACT>(xb + cj * XSB + 4 * rq, nst1);
      MPPO_SCHED_FENCE();
#pragma unroll
      for (int S = 0; S < NS1; ++S) w1.mfma_stage(S, At, acc);
      MPPO_SCHED_FENCE();
    }
    for (int r = 0; r < 4; ++r) { acc0[r] = acc[0][0][r] + acc[1][0][r]; acc1[r] = acc[0][1][r] + acc[1][1][r]; }
    MPPO_SCHED_FENCE();
    {
      // this wave's 32 rows of W3 -> its own LDS scratch -> the two register arrangements the matrix cores want: B operands of the head
      // product (k = 32 wave + 16 g + 4 rq + c, output o) and of dZ2 = dOut . W3^T (hidden columns c0, c0 + 1, output ai = 4 m + rq).
      // Wave-local: no barrier, the scratch (over this wave's share of the h2 / head-partial tiles) is dead again before anybody writes there.
      float* wsc = h2t + wave * ((FRT * HS + nw * OT * 256) / nw);
      if (lane < nq4) *reinterpret_cast<float4*>(wsc + 4 * lane) = w3r0;
      if (lane + 64 < nq4) *reinterpret_cast<float4*>(wsc + 4 * (lane + 64)) = w3r1;
      if (NW3 > 2 && lane + 128 < nq4) *reinterpret_cast<float4*>(wsc + 4 * (lane + 128)) = w3r2;
      if (NW3 > 2 && lane + 192 < nq4) *reinterpret_cast<float4*>(wsc + 4 * (lane + 192)) = w3r3;
      MPPO_WAVE_SYNC();
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int gc = 0; gc < 8; ++gc) {
          const int o = cj + 16 * ot;
          w3p[ot][gc] = wsc[(16 * (gc >> 2) + 4 * rq + (gc & 3)) * nout + (o < nout ? o : nout - 1)];  // (columns o >= nout of the product are never read)
        }
#pragma unroll
      for (int m = 0; m < 4 * OT; ++m) {
        const int ai = 4 * m + rq, ac = ai < nout ? ai : nout - 1;
        w3q[m][0] = wsc[(2 * cj) * nout + ac];  // (dOut is zero in the columns ai >= nout it multiplies)
        w3q[m][1] = wsc[(2 * cj + 1) * nout + ac];
      }
    }
    MPPO_SCHED_FENCE();
#ifdef MPPO_FUSED_TIMERS
    if (blockIdx.x == 40 && blockIdx.y == 0 && lane == 0) g_fused_t[24 + 8 + wave] = __builtin_amdgcn_s_memtime();
#endif
    FT(4);
    float q0[4], q1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 4 * rq + r;
      float v0 = acc0[r] + bz1.x, v1 = acc1[r] + bz1.y;
      if (tanh_act) { v0 = fused_tanh(v0); v1 = fused_tanh(v1); } else { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      h1r[0][r] = v0; h1r[1][r] = v1;
      *reinterpret_cast<float*>(h1b + rr * HSB + c0) = bf16x2_bits(v0, v1);
      const bool on = row0 + rr < a.mb;
      q0[r] = on ? v0 : 0.f; q1[r] = on ? v1 : 0.f;
    }
    size_t qi, qi2;
    pair_index<true>(row0 + 4 * rq, c0, H, qi, qi2);
    store_quad2<true>(a.h1[net], qi, qi2, q0, q1);
  }
  __syncthreads();
  FT(5);
  // ---- P2: h2 = act(h1 . W2 + b2); the first half of W2^T requested between the stages ----
  {
    f32x4 acc[2][2], acc0, acc1;
    for (int r = 0; r < 4; ++r) { acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f; }
#ifdef MPPO_FUSED_TIMERS
    if (blockIdx.x == 40 && blockIdx.y == 0 && lane == 0) g_fused_t[24 + 16 + wave] = __builtin_amdgcn_s_memtime();
#endif
    {
#pragma unroll
      for (int S = 0; S < 6; ++S) { WL(w5, S, 0); WL(w5, S, 1); }
      const auto At = w2.template read_a<EXACT>(h1b + cj * HSB + 4 * rq, nsth);
      MPPO_SCHED_FENCE();
#pragma unroll
      for (int S = 0; S < NSH; ++S) w2.mfma_stage(S, At, acc);
      MPPO_SCHED_FENCE();
    }
    for (int r = 0; r < 4; ++r) { acc0[r] = acc[0][0][r] + acc[1][0][r]; acc1[r] = acc[0][1][r] + acc[1][1][r]; }
#ifdef MPPO_FUSED_TIMERS
    if (blockIdx.x == 40 && blockIdx.y == 0 && lane == 0) g_fused_t[24 + 16 + 8 + wave] = __builtin_amdgcn_s_memtime();
#endif
    FT(6);
    float q0[4], q1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 4 * rq + r;
      float v0 = acc0[r] + bz2.x, v1 = acc1[r] + bz2.y;
      if (tanh_act) { v0 = fused_tanh(v0); v1 = fused_tanh(v1); } else { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      h2r[0][r] = v0; h2r[1][r] = v1;
      *reinterpret_cast<float2*>(h2t + rr * HS + c0) = make_float2(v0, v1);
      const bool on = row0 + rr < a.mb;
      q0[r] = on ? v0 : 0.f; q1[r] = on ? v1 : 0.f;
    }
    size_t qi, qi2;
    pair_index<true>(row0 + 4 * rq, c0, H, qi, qi2);
    store_quad2<true>(a.h2[net], qi, qi2, q0, q1);
  }
  __syncthreads();
  FT(7);
  FT(8);
  // ---- P3: output layer, exact float: OT 16x16 tiles (rows x outputs), K = H split over the waves, partial tiles summed through LDS ----
  {
    WL(w5, 6, 0); WL(w5, 6, 1); WL(w5, 7, 0); WL(w5, 7, 1);
    f32x4 hp[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
      for (int r = 0; r < 4; ++r) hp[ot][r] = 0.f;
    const float* arow = h2t + cj * HS + 4 * rq + 32 * wave;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float4 av = *reinterpret_cast<const float4*>(arow + 16 * g);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        mfma_f32_16x16x4(av.x, w3p[ot][4 * g + 0], hp[ot]); mfma_f32_16x16x4(av.y, w3p[ot][4 * g + 1], hp[ot]);
        mfma_f32_16x16x4(av.z, w3p[ot][4 * g + 2], hp[ot]); mfma_f32_16x16x4(av.w, w3p[ot][4 * g + 3], hp[ot]);
      }
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
      *reinterpret_cast<float4*>(s_hp + ((wave * OT + ot) * 64 + lane) * 4) = make_float4(hp[ot][0], hp[ot][1], hp[ot][2], hp[ot][3]);
  }
  __syncthreads();
  FT(9);
  // ---- loss terms and d(loss)/d(outputs): the float kernel's arithmetic, statement for statement (k_fused.hip).  One wave's lanes cover
  // all 16 rows x 16 OT outputs, and only wave 0's results are stored: on the GPU only wave 0 runs this block - its SIMD's second wave
  // (wave 4) waits at the barrier instead of taking every other issue slot for a copy of the same arithmetic.  (The emulator's cross-lane
  // shims are workgroup-wide barriers: there every wave takes part, as in fused_mlp_kernel.)
  if (kLossAllWaves || wave == 0) {
    float out[OT][4];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      out[ot][0] = out[ot][1] = out[ot][2] = out[ot][3] = 0.f;
      float4 hq[8];  // all partial tiles requested together (a loop over the run-time wave count waited for every one of them in turn)
#pragma unroll
      for (int w = 0; w < 8; ++w) hq[w] = *reinterpret_cast<const float4*>(s_hp + (((EXACT || w < nw ? w : 0) * OT + ot) * 64 + lane) * 4);
#pragma unroll
      for (int w = 0; w < 8; ++w)
        if (EXACT || w < nw) { out[ot][0] += hq[w].x; out[ot][1] += hq[w].y; out[ot][2] += hq[w].z; out[ot][3] += hq[w].w; }
    }
    const bool st = wave == 0;
    float sum_ls_l = 0.f;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) sum_ls_l += cj + 16 * ot < A ? ls[ot] : 0.f;
    if (net == 0) {
      const float sum_ls = group16_sum(sum_ls_l);
      float dmq[OT][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 4 * rq + r, i = row0 + rr;
        const bool on = i < a.mb;
        float z[OT], zz = 0.f;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          z[ot] = 0.f;
          if (on && cj + 16 * ot < A) z[ot] = (pf0[ot][r] - (out[ot][r] + b3v[ot])) * __expf(-ls[ot]);
          zz += z[ot] * z[ot];
        }
        const float ss = group16_sum(zz);
        float la = 0.f, dlogp = 0.f;
        if (on) {
          const float logp = -0.5f * ss - sum_ls - 0.5f * (float)A * kLog2PiF;
          const float ratio = __expf(logp - pf1[r]);
          const float g = (pf2[r] - adv_mean) * adv_rstd;
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
      if (st) {
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const int o = cj + 16 * ot;
          if (o < AP) store_quad<true>(a.dout, quad_index(row0 + 4 * rq, o, a.DP), dmq[ot]);
        }
      }
    } else {
      const float b3c = group16_sum(cj == 0 ? b3v[0] : 0.f);  // lane cj = 0 holds the critic's single output bias
      float dvq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 4 * rq + r, i = row0 + rr;
        const bool on = i < a.mb;
        const float vnew = group16_sum(cj == 0 ? out[0][r] : 0.f) + b3c;
        float lv = 0.f, dv = 0.f;
        if (on) {
          const float ov = pf0[0][r], tg = pf1[r];
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
      if (st && cj < 4) {
        const float zq[4] = {cj == 0 ? dvq[0] : 0.f, cj == 0 ? dvq[1] : 0.f, cj == 0 ? dvq[2] : 0.f, cj == 0 ? dvq[3] : 0.f};
        store_quad<true>(a.dout, quad_index(row0 + 4 * rq, AP + cj, a.DP), zq);
      }
    }
  }
  __syncthreads();
  FT(10);
  // per-workgroup partial sums, partial[blockIdx.x][4 + AP]: col 0 actor loss, col 1 value loss, 4 + o: d log_std[o].  Sixteen rows = one
  // 16-lane DPP sum (every wave takes part - the cross-lane operations are workgroup-uniform - one lane stores)
  {
    float* prow_out = a.partial + (size_t)blockIdx.x * (4 + AP);
    const float sl = group16_sum(s_l[t & 15]);
    if (t == 0) prow_out[net] = sl;
    if (net == 0) {
      const int cpr = nthr >> 4;  // columns per round
      for (int o0 = 0; o0 < AP; o0 += cpr) {
        const int o = o0 + (t >> 4);
        const float s = group16_sum(o < A ? s_red[(t & 15) * SD + o] : 0.f);
        if ((t & 15) == 0 && o < AP) prow_out[4 + o] = s;
      }
    }
  }
  FT(11);
  // ---- P4: dZ2 = (dOut . W3^T) * act'(h2), exact float product -> LDS (rounded: the backward product's A operand) + global ----
  {
    f32x4 d0, d1;
    for (int r = 0; r < 4; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
    float av[4 * OT];
#pragma unroll
    for (int m = 0; m < 4 * OT; ++m) av[m] = s_do[cj * SD + 4 * m + rq];
    MPPO_SCHED_FENCE();
#pragma unroll
    for (int m = 0; m < 4 * OT; ++m) { mfma_f32_16x16x4(av[m], w3q[m][0], d0); mfma_f32_16x16x4(av[m], w3q[m][1], d1); }
    float q0[4], q1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 4 * rq + r;
      const float hx = h2r[0][r], hy = h2r[1][r];
      const float z0 = tanh_act ? d0[r] * (1.f - hx * hx) : (hx > 0.f ? d0[r] : 0.f);
      const float z1 = tanh_act ? d1[r] * (1.f - hy * hy) : (hy > 0.f ? d1[r] : 0.f);
      *reinterpret_cast<float*>(dzb + rr * HSB + c0) = bf16x2_bits(z0, z1);
      const bool on = row0 + rr < a.mb;
      q0[r] = on ? z0 : 0.f; q1[r] = on ? z1 : 0.f;
    }
    size_t qi, qi2;
    pair_index<true>(row0 + 4 * rq, c0, H, qi, qi2);
    store_quad2<true>(a.dz2[net], qi, qi2, q0, q1);
  }
  __syncthreads();
  FT(12);
  // ---- P5: dZ1 = (dZ2 . W2^T) * act'(h1) ----
  {
    f32x4 acc[2][2], acc0, acc1;
    for (int r = 0; r < 4; ++r) { acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f; }
    {
      const auto At = w5.template read_a<EXACT>(dzb + cj * HSB + 4 * rq, nsth);
      MPPO_SCHED_FENCE();
#pragma unroll
      for (int S = 0; S < NSH; ++S) w5.mfma_stage(S, At, acc);
    }
    for (int r = 0; r < 4; ++r) { acc0[r] = acc[0][0][r] + acc[1][0][r]; acc1[r] = acc[0][1][r] + acc[1][1][r]; }
    FT(13);
    float q0[4], q1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool on = row0 + 4 * rq + r < a.mb;
      const float gx = h1r[0][r], gy = h1r[1][r];
      const float d0 = tanh_act ? acc0[r] * (1.f - gx * gx) : (gx > 0.f ? acc0[r] : 0.f);
      const float d1 = tanh_act ? acc1[r] * (1.f - gy * gy) : (gy > 0.f ? acc1[r] : 0.f);
      q0[r] = on ? d0 : 0.f; q1[r] = on ? d1 : 0.f;
    }
    size_t qi, qi2;
    pair_index<true>(row0 + 4 * rq, c0, H, qi, qi2);
    store_quad2<true>(a.dz1[net], qi, qi2, q0, q1);
  }
  FT(14);
#undef WL
}

inline size_t bf16_rowpass_smem_bytes(int O, int A, int H) {
  const int KP = (O + 31) & ~31, OT = A > 16 ? 2 : 1, nw = H / 32;
  return (size_t)FRT * (KP + 8) * 2 + 2 * (size_t)FRT * (H + 8) * 2 + sizeof(float) * ((size_t)FRT * (H + 4) + (size_t)nw * OT * 256 + 2 * FRT * 16 * OT + FRT);
}
constexpr int kBf16RowpassNS1 = 8;  // first-layer stages held in registers: observations up to 256 wide; wider ones take fused_mlp_kernel<true, ...>
inline bool bf16_rowpass_supported(const mppo_net_t& net) {
  return net.bf16 && ((net.O + 31) & ~31) <= 32 * kBf16RowpassNS1 && net.H % 32 == 0 && net.H >= 32 && net.H <= 256 && net.A <= 32 &&
         bf16_rowpass_smem_bytes(net.O, net.A, net.H) <= 64 * 1024;
}
