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
// Outputs to HBM/L2: h1, h2, dZ2, dZ1 (operands of the weight-gradient GEMM), dOut, the gathered rows (xmb), loss partials.
#include <wave_ops.h>

#include "mppo_common.h"
#include "ppo_layout.h"

namespace mppo {

constexpr float kLog2PiF = 1.8378770664093453f;
constexpr int FRT = 16;  // rows per workgroup

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
};

// one 16-row x 32-col slab of  act(A_tile . W + bias)  (A_tile in LDS, row stride AS; W [K,H] in global memory)
//   NT = false: W stored [K][H] (forward);  NT = true: W stored [H][K], i.e. B(k,n) = W[n*K + k] (backward dZ1)
template <bool NT>
__device__ __forceinline__ void tile_gemm(const float* At, int AS, int K, int Kvalid, const float* W, int H, int n0, int lane, f32x4& acc0, f32x4& acc1) {
  const int i = lane & 15, kq = lane >> 4;
  const float* arow = At + i * AS + 4 * kq;
  const int ngroups = (K + 15) / 16;
  float bw0[4], bw1[4], nb0[4], nb1[4];
  auto loadB = [&](int G, float (&x0)[4], float (&x1)[4]) {
    if (NT) {
      const float4 q0 = *reinterpret_cast<const float4*>(W + (size_t)(n0 + i) * K + 16 * G + 4 * kq);
      const float4 q1 = *reinterpret_cast<const float4*>(W + (size_t)(n0 + 16 + i) * K + 16 * G + 4 * kq);
      x0[0] = q0.x; x0[1] = q0.y; x0[2] = q0.z; x0[3] = q0.w; x1[0] = q1.x; x1[1] = q1.y; x1[2] = q1.z; x1[3] = q1.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int k = 16 * G + 4 * kq + c;
        k = k < Kvalid ? k : Kvalid - 1;  // rows past K meet zero activations; clamp keeps the read inside W
        x0[c] = W[(size_t)k * H + n0 + i];
        x1[c] = W[(size_t)k * H + n0 + 16 + i];
      }
    }
  };
  loadB(0, bw0, bw1);
  for (int G = 0; G < ngroups; ++G) {
    const float4 av = *reinterpret_cast<const float4*>(arow + 16 * G);
    const bool more = G + 1 < ngroups;
    if (more) loadB(G + 1, nb0, nb1);
    mfma_f32_16x16x4(av.x, bw0[0], acc0); mfma_f32_16x16x4(av.x, bw1[0], acc1);
    mfma_f32_16x16x4(av.y, bw0[1], acc0); mfma_f32_16x16x4(av.y, bw1[1], acc1);
    mfma_f32_16x16x4(av.z, bw0[2], acc0); mfma_f32_16x16x4(av.z, bw1[2], acc1);
    mfma_f32_16x16x4(av.w, bw0[3], acc0); mfma_f32_16x16x4(av.w, bw1[3], acc1);
    if (more) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { bw0[c] = nb0[c]; bw1[c] = nb1[c]; }
    }
  }
}

template <int LRW>
__device__ __forceinline__ float row32_sum(float x) {
  x = group16_sum(x);
  x += __shfl_xor(x, 16);
  return x;
}

__global__ void __launch_bounds__(512) fused_mlp_kernel(FusedArgs a) {
  MPPO_DYN_SMEM(smem_raw);
  float* sm = reinterpret_cast<float*>(smem_raw);
  const int H = a.H, O = a.O, OP = a.OP, A = a.A, AP = a.AP;
  const int net = blockIdx.y;  // 0 actor, 1 critic
  const int KP = (O + 15) & ~15;
  const int XS = KP + 4, HS = H + 4;
  const int R0 = FRT * (XS > HS ? XS : HS);
  float* xt = sm;            // [16][XS]  then dZ2 tile [16][HS]
  float* h1t = sm + R0;      // [16][HS]
  float* h2t = h1t + FRT * HS;
  float* w3s = h2t + FRT * HS;          // actor: [H][A], critic: [H]
  float* s_do = w3s + H * (A > 1 ? A : 1);  // [16][32]  d mean (cols < A) | critic: col 0 = d value
  float* s_red = s_do + FRT * 32;       // [16][32]
  float* s_l = s_red + FRT * 32;        // [16]
  const int t = threadIdx.x, nthr = blockDim.x, lane = t & 63, wave = t >> 6;
  const int row0 = blockIdx.x * FRT;
  const bool tanh_act = net == 0 && a.use_tanh;
  const float* W1 = a.params + (net ? a.L.c_w1 : a.L.a_w1);
  const float* B1 = a.params + (net ? a.L.c_b1 : a.L.a_b1);
  const float* W2 = a.params + (net ? a.L.c_w2 : a.L.a_w2);
  const float* B2 = a.params + (net ? a.L.c_b2 : a.L.a_b2);
  const float* W3 = a.params + (net ? a.L.c_w3 : a.L.a_w3);
  const float* B3 = a.params + (net ? a.L.c_b3 : a.L.a_b3);
  const int nout = net ? 1 : A;

  // ---- P0: gathered observation rows -> LDS (zero-padded to KP columns); the actor workgroup also writes xmb ----
  for (int e = t; e < FRT * (KP / 4); e += nthr) {
    const int r = e / (KP / 4), c4 = (e % (KP / 4)) * 4;
    const int gi = row0 + r < a.mb ? row0 + r : a.mb - 1;
    const long row = a.idx ? a.idx[gi] : gi;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < OP) q = *reinterpret_cast<const float4*>(a.b.obs + row * a.b.obs_ld + c4);
    *reinterpret_cast<float4*>(xt + r * XS + c4) = q;
    if (net == 0 && c4 < OP && row0 + r < a.mb) *reinterpret_cast<float4*>(a.xmb + (size_t)(row0 + r) * OP + c4) = q;
  }
  for (int e = t; e < H * nout; e += nthr) w3s[e] = W3[e];
  __syncthreads();

  const int n0 = 32 * wave;
  const int cj = lane & 15, rq = lane >> 4;
  // ---- P1 / P2: hidden layers ----
  for (int layer = 0; layer < 2; ++layer) {
    f32x4 acc0, acc1;
    for (int r = 0; r < 4; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    if (layer == 0) tile_gemm<false>(xt, XS, KP, O, W1, H, n0, lane, acc0, acc1);
    else tile_gemm<false>(h1t, HS, H, H, W2, H, n0, lane, acc0, acc1);
    const float* bias = layer == 0 ? B1 : B2;
    float* ht = layer == 0 ? h1t : h2t;
    float* hg = layer == 0 ? a.h1[net] : a.h2[net];
    const float bz0 = bias[n0 + cj], bz1 = bias[n0 + 16 + cj];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 4 * rq + r;
      float v0 = acc0[r] + bz0, v1 = acc1[r] + bz1;
      if (tanh_act) { v0 = fused_tanh(v0); v1 = fused_tanh(v1); } else { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      ht[rr * HS + n0 + cj] = v0;
      ht[rr * HS + n0 + 16 + cj] = v1;
      if (row0 + rr < a.mb) {
        hg[(size_t)(row0 + rr) * H + n0 + cj] = v0;
        hg[(size_t)(row0 + rr) * H + n0 + 16 + cj] = v1;
      }
    }
    __syncthreads();
  }

  // ---- P3: output layer + loss terms of this network; 32 lanes per row ----
  const int rows_per_pass = nthr / 32;
  for (int rbase = 0; rbase < FRT; rbase += rows_per_pass) {
    const int r = rbase + t / 32, o = t & 31;
    const int i = row0 + r;
    const bool on = i < a.mb;
    long row = 0;
    if (on) row = a.idx ? a.idx[i] : i;
    const float* hrow = h2t + r * HS;
    if (net == 0) {
      float out = 0.f;
      if (o < A) {
        float s0 = 0.f, s1 = 0.f;
        for (int k = 0; k < H; k += 2) { s0 += hrow[k] * w3s[k * A + o]; s1 += hrow[k + 1] * w3s[(k + 1) * A + o]; }
        out = s0 + s1 + B3[o];
      }
      const float ls = o < A ? a.params[a.L.log_std + o] : 0.f;
      const float inv_std = __expf(-ls);
      const float sum_ls = row32_sum<32>(ls);
      float z = 0.f;
      if (on && o < A) z = (a.b.action[row * a.b.act_ld + o] - out) * inv_std;
      const float ss = row32_sum<32>(z * z);
      float la = 0.f, dlogp = 0.f;
      if (on) {
        const float logp = -0.5f * ss - sum_ls - 0.5f * (float)A * kLog2PiF;
        const float ratio = __expf(logp - a.b.log_prob[row]);
        const float g = (a.b.adv[row] - a.adv_stat[0]) * a.adv_stat[1];
        const float la1 = ratio * g;
        const float la2 = fminf(fmaxf(ratio, 1.f - a.lc.clip_eps), 1.f + a.lc.clip_eps) * g;
        la = -fminf(la1, la2) * a.inv_count;
        const bool unclipped = (ratio >= 1.f - a.lc.clip_eps) && (ratio <= 1.f + a.lc.clip_eps);
        dlogp = (unclipped || la1 < la2) ? -g * ratio * a.inv_count : 0.f;
      }
      const float dm = o < A ? dlogp * z * inv_std : 0.f;
      s_do[r * 32 + o] = dm;
      s_red[r * 32 + o] = o < A ? dlogp * (z * z - 1.f) : 0.f;
      if (o == 0) s_l[r] = la;
      if (on && o < AP) a.dout[(size_t)i * a.DP + o] = dm;
    } else {
      float s0 = 0.f;
      for (int k = o; k < H; k += 32) s0 += hrow[k] * w3s[k];
      const float vnew = row32_sum<32>(s0) + B3[0];
      float lv = 0.f, dv = 0.f;
      if (on) {
        const float ov = a.b.value[row], tg = a.b.target[row];
        const float vc = ov + fminf(fmaxf(vnew - ov, -a.lc.clip_eps), a.lc.clip_eps);
        const float vl1 = (vnew - tg) * (vnew - tg), vl2 = (vc - tg) * (vc - tg);
        lv = 0.5f * fmaxf(vl1, vl2) * a.inv_count;
        const bool vin = fabsf(vnew - ov) <= a.lc.clip_eps;
        dv = (vin || vl1 > vl2) ? (vnew - tg) * a.inv_count * a.lc.vf_coef : 0.f;
      }
      if (o == 0) { s_do[r * 32] = dv; s_l[r] = lv; }
      if (on && o < 4) a.dout[(size_t)i * a.DP + AP + o] = o == 0 ? dv : 0.f;
    }
  }
  __syncthreads();
  // per-workgroup partial sums: partial[blockIdx.x][4+AP]: col 0 actor loss, col 1 value loss, 4+a d log_std[a]
  {
    float* prow = a.partial + (size_t)blockIdx.x * (4 + AP);
    if (net == 0) {
      if (t == 0) { float s = 0.f; for (int r = 0; r < FRT; ++r) s += s_l[r]; prow[0] = s; }
      if (t >= 4 && t < 4 + AP) { float s = 0.f; if (t - 4 < A) for (int r = 0; r < FRT; ++r) s += s_red[r * 32 + (t - 4)]; prow[t] = s; }
    } else if (t == 0) {
      float s = 0.f; for (int r = 0; r < FRT; ++r) s += s_l[r]; prow[1] = s;
    }
  }
  // ---- P4: dZ2 = (dOut . W3^T) * act'(h2) -> LDS (over the dead x tile) + global ----
  float* dzt = xt;
  for (int e = t; e < FRT * H; e += nthr) {
    const int r = e / H, n = e - r * H;
    float s = 0.f;
    if (net == 0) { for (int k = 0; k < A; ++k) s += s_do[r * 32 + k] * w3s[n * A + k]; }
    else s = s_do[r * 32] * w3s[n];
    const float hv = h2t[r * HS + n];
    s = tanh_act ? s * (1.f - hv * hv) : (hv > 0.f ? s : 0.f);
    dzt[r * HS + n] = s;
    if (row0 + r < a.mb) a.dz2[net][(size_t)(row0 + r) * H + n] = s;
  }
  __syncthreads();
  // ---- P5: dZ1 = (dZ2 . W2^T) * act'(h1) ----
  {
    f32x4 acc0, acc1;
    for (int r = 0; r < 4; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    tile_gemm<true>(dzt, HS, H, H, W2, H, n0, lane, acc0, acc1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 4 * rq + r;
      if (row0 + rr < a.mb) {
        const float g0 = h1t[rr * HS + n0 + cj], g1 = h1t[rr * HS + n0 + 16 + cj];
        a.dz1[net][(size_t)(row0 + rr) * H + n0 + cj] = tanh_act ? acc0[r] * (1.f - g0 * g0) : (g0 > 0.f ? acc0[r] : 0.f);
        a.dz1[net][(size_t)(row0 + rr) * H + n0 + 16 + cj] = tanh_act ? acc1[r] * (1.f - g1 * g1) : (g1 > 0.f ? acc1[r] : 0.f);
      }
    }
  }
}

size_t fused_smem_bytes(int O, int A, int H) {
  const int KP = (O + 15) & ~15, XS = KP + 4, HS = H + 4;
  const size_t R0 = (size_t)FRT * (XS > HS ? XS : HS);
  return sizeof(float) * (R0 + 2 * (size_t)FRT * HS + (size_t)H * (A > 1 ? A : 1) + 2 * FRT * 32 + FRT);
}

bool fused_supported(const mppo_net_t& net, const mppo_batch_t& b) {
  return net.H % 32 == 0 && net.H >= 32 && net.H <= 512 && net.A <= 31 && net.bf16 == 0 && b.obs_ld == net.OP && (net.OP % 4) == 0 &&
         (reinterpret_cast<uintptr_t>(b.obs) & 15) == 0 && fused_smem_bytes(net.O, net.A, net.H) <= 160 * 1024 &&
         (param_layout(net.O, net.A, net.H).c_w2 % 4) == 0;  // float4 rows of W2 in the backward product
}

int32_t fused_forward_backward(const mppo_net_t& net, const float* params, const mppo_batch_t& batch, const int* idx, int mb, const float* adv_stat,
                               float inv_count, const mppo_loss_cfg_t& lc, const GradBufs& g, hipStream_t stream) {
  FusedArgs a{};
  a.mb = mb; a.O = net.O; a.OP = net.OP; a.A = net.A; a.AP = g.f.AP; a.DP = g.f.AP + 4; a.H = net.H; a.use_tanh = net.use_tanh;
  a.params = params; a.L = param_layout(net.O, net.A, net.H); a.b = batch; a.idx = idx; a.adv_stat = adv_stat; a.inv_count = inv_count; a.lc = lc;
  a.h1[0] = g.f.h1a; a.h1[1] = g.f.h1c; a.h2[0] = g.f.h2a; a.h2[1] = g.f.h2c; a.dz2[0] = g.dz2a; a.dz2[1] = g.dz2c; a.dz1[0] = g.dz1a; a.dz1[1] = g.dz1c;
  a.dout = g.dout; a.xmb = g.xmb; a.partial = g.partial;
  const size_t smem = fused_smem_bytes(net.O, net.A, net.H);
  static thread_local size_t attr_for = 0;
  if (smem > 64 * 1024 && attr_for < smem) {
    MPPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_mlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_for = smem;
  }
  hipLaunchKernelGGL(fused_mlp_kernel, dim3(cdiv(mb, FRT), 2), dim3(2 * net.H), smem, stream, a);
  MPPO_CHECK_LAUNCH("fused_mlp_kernel");
  return MPPO_OK;
}

}  // namespace mppo
