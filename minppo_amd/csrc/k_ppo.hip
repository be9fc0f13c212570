// k_ppo.hip — PPO stages around the MLP GEMMs: policy sample / log-prob, GAE reverse scan,
// clipped-surrogate loss and its output gradients, per-minibatch advantage statistics,
// split-K slab reduction, global-norm clip + Adam.  Reference lines are cited per kernel.
#include <wave_ops.h>

#include <cmath>
#include <cstdlib>

#include "gemm.h"
#include "philox.h"
#include "mppo_common.h"
#include "ppo_layout.h"
#include "wgrad.h"
#include "peer.h"

namespace mppo {

constexpr float kLog2Pi = 1.8378770664093453f;

// ------------------------------------------------------------------------------------------------
// _calculate_gae (train.py:185-205): reverse scan over t, one thread per environment.
// HBM-streaming: 9 bytes read + 8 written per sample, coalesced over n.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gae_kernel(int T, int N, float gamma, float lam, const float* __restrict__ reward,
                                                  const float* __restrict__ value, const unsigned char* __restrict__ done,
                                                  const float* __restrict__ last_val, float* __restrict__ adv, float* __restrict__ target) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float gae = 0.f, next_value = last_val[n];
  const float gl = gamma * lam;
  for (int t = T - 1; t >= 0; --t) {
    const size_t o = (size_t)t * N + n;
    const float nd = done[o] ? 0.f : 1.f;  // (1 - done), bool -> int -> float
    const float v = value[o];
    const float delta = reward[o] + gamma * next_value * nd - v;
    gae = delta + gl * nd * gae;
    adv[o] = gae;
    target[o] = gae + v;
    next_value = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Output layers fused with what consumes them.  One workgroup = RT = 256/LR rows; LR lanes (one or two DPP rows) per
// row, lane o < A owns action dimension o, lane o == A owns the critic value.  W3 of both networks and the two h2 tiles
// are staged in LDS; per-row reductions over the action dimensions are DPP row sums.
//   head_sample_kernel : mean = h2a.W3a + b3a, value = h2c.W3c + b3c, action = mean + exp(log_std)*noise, log_prob
//                        (train.py:79-83,157-160)
//   head_loss_kernel   : the same heads, then `_loss_fn` on them (train.py:223-243), d(loss)/d(mean, value, log_std) and
//                        the first backward product dZ2 = (dOut . W3^T) * act'(h2) for both networks.
//                        dOut [mb, DP]: columns [0,A) = d mean, column AP = d value (DP = AP + 4), the layout the
//                        weight-gradient GEMM reads.  Per-workgroup partial sums -> partial[blk][4 + AP].
// ------------------------------------------------------------------------------------------------
template <int LR>
__device__ __forceinline__ float row_sum(float x) {
  x = group16_sum(x);
  if (LR >= 32) x += __shfl_xor(x, 16);
  if (LR == 64) x += __shfl_xor(x, 32);
  return x;
}

struct HeadArgs {
  int n, A, AP, DP, H, use_tanh;
  const float *h2a, *h2c, *w3a, *b3a, *w3c, *b3c, *log_std;
  // sample
  const float* noise; float* action; float* log_prob; float* value; float* mean_out;
  // loss
  const int* idx; mppo_batch_t b; const float* adv_stat; float inv_count; mppo_loss_cfg_t lc;
  float *dout, *dz2a, *dz2c, *partial;
};

template <int LR, bool LOSS>
__global__ void __launch_bounds__(256) head_kernel(HeadArgs a) {
  constexpr int RT = 256 / LR;
  MPPO_DYN_SMEM(smem_raw);
  float* sm = reinterpret_cast<float*>(smem_raw);
  const int H = a.H, A = a.A, AP = a.AP;
  const int HS = H + 4;                // padded row stride: rows r, r+1 start 4 banks apart (unpadded: 16-way conflict)
  float* s_h2a = sm;                  // [RT][HS]
  float* s_h2c = s_h2a + RT * HS;     // [RT][HS]
  float* s_w3a = s_h2c + RT * HS;     // [H][A]
  float* s_w3c = s_w3a + H * A;       // [H]
  float* s_do = s_w3c + H;            // [RT][LR]   d mean (cols < A), d value (col A)
  float* s_red = s_do + RT * LR;      // [RT][LR]   per-row partials for the block reduction
  const int t = threadIdx.x, r = t / LR, o = t % LR;
  const int row0 = blockIdx.x * RT;
  const int i = row0 + r;
  const bool on = i < a.n;
  // stage h2 tiles (rows past n are clamped, never stored) and the head weights
  for (int e = t; e < RT * H / 4; e += 256) {
    const int rr = (e * 4) / H, cc = (e * 4) % H;
    const int gi = row0 + rr < a.n ? row0 + rr : a.n - 1;
    *reinterpret_cast<float4*>(s_h2a + rr * HS + cc) = *reinterpret_cast<const float4*>(a.h2a + (size_t)gi * H + cc);
    *reinterpret_cast<float4*>(s_h2c + rr * HS + cc) = *reinterpret_cast<const float4*>(a.h2c + (size_t)gi * H + cc);
  }
  for (int e = t; e < H * A; e += 256) s_w3a[e] = a.w3a[e];
  for (int e = t; e < H; e += 256) s_w3c[e] = a.w3c[e];
  __syncthreads();
  // head outputs: lane o < A -> mean[o], lane o == A -> value
  float out = 0.f;
  if (o < A) {
    float s0 = 0.f, s1 = 0.f;
    const float* hrow = s_h2a + r * HS;
    for (int k = 0; k < H; k += 2) { s0 += hrow[k] * s_w3a[k * A + o]; s1 += hrow[k + 1] * s_w3a[(k + 1) * A + o]; }
    out = s0 + s1 + a.b3a[o];
  } else if (o == A) {
    float s0 = 0.f, s1 = 0.f;
    const float* hrow = s_h2c + r * HS;
    for (int k = 0; k < H; k += 2) { s0 += hrow[k] * s_w3c[k]; s1 += hrow[k + 1] * s_w3c[k + 1]; }
    out = s0 + s1 + a.b3c[0];
  }
  const float ls = o < A ? a.log_std[o] : 0.f;
  const float inv_std = __expf(-ls);
  const float sum_ls = row_sum<LR>(ls);
  if (!LOSS) {
    // ---- pi.sample + pi.log_prob (train.py:158-160) ----
    float z2 = 0.f;
    if (a.noise && o < A && on) {
      const float act = out + __expf(ls) * a.noise[(size_t)i * A + o];
      a.action[(size_t)i * A + o] = act;
      const float z = (act - out) * inv_std;
      z2 = z * z;
      if (a.mean_out) a.mean_out[(size_t)i * AP + o] = out;
    }
    const float ss = row_sum<LR>(z2);
    if (on && o == 0 && a.noise) a.log_prob[i] = -0.5f * ss - sum_ls - 0.5f * (float)A * kLog2Pi;
    if (on && o == A) a.value[i] = out;
    return;
  }
  // ---- _loss_fn (train.py:223-243) ----
  long row = 0;
  float z = 0.f;
  if (on) row = a.idx ? a.idx[i] : i;
  if (on && o < A) z = (a.b.action[row * a.b.act_ld + o] - out) * inv_std;
  const float ss = row_sum<LR>(z * z);
  const float vnew = row_sum<LR>(o == A ? out : 0.f);  // broadcast the value to the row's lanes
  float la = 0.f, lv = 0.f, dlogp = 0.f, dv = 0.f;
  if (on) {
    const float logp = -0.5f * ss - sum_ls - 0.5f * (float)A * kLog2Pi;
    const float ratio = __expf(logp - a.b.log_prob[row]);
    const float g = (a.b.adv[row] - a.adv_stat[0]) * a.adv_stat[1];
    const float la1 = ratio * g;
    const float la2 = fminf(fmaxf(ratio, 1.f - a.lc.clip_eps), 1.f + a.lc.clip_eps) * g;
    la = -fminf(la1, la2) * a.inv_count;
    const bool unclipped = (ratio >= 1.f - a.lc.clip_eps) && (ratio <= 1.f + a.lc.clip_eps);
    dlogp = (unclipped || la1 < la2) ? -g * ratio * a.inv_count : 0.f;
    const float ov = a.b.value[row], tg = a.b.target[row];
    const float vc = ov + fminf(fmaxf(vnew - ov, -a.lc.clip_eps), a.lc.clip_eps);
    const float vl1 = (vnew - tg) * (vnew - tg), vl2 = (vc - tg) * (vc - tg);
    lv = 0.5f * fmaxf(vl1, vl2) * a.inv_count;
    const bool vin = fabsf(vnew - ov) <= a.lc.clip_eps;
    dv = (vin || vl1 > vl2) ? (vnew - tg) * a.inv_count * a.lc.vf_coef : 0.f;
  }
  const float dm = (o < A) ? dlogp * z * inv_std : (o == A ? dv : 0.f);
  const float dls = (o < A) ? dlogp * (z * z - 1.f) : 0.f;
  s_do[r * LR + o] = dm;
  // per-row partials for the block sums: lane 0 carries la, lane 1 carries lv, lanes 2.. are free; dls by lane o
  s_red[r * LR + o] = dls;
  if (on) {
    if (o < A) a.dout[(size_t)i * a.DP + o] = dm;
    else if (o < AP) a.dout[(size_t)i * a.DP + o] = 0.f;
    if (o == A) { a.dout[(size_t)i * a.DP + AP] = dv; a.dout[(size_t)i * a.DP + AP + 1] = 0.f; a.dout[(size_t)i * a.DP + AP + 2] = 0.f; a.dout[(size_t)i * a.DP + AP + 3] = 0.f; }
  }
  __shared__ float s_l[2][256 / 16];
  if (o == 0) { s_l[0][r] = la; s_l[1][r] = lv; }
  __syncthreads();
  if (t < 4 + AP) {
    float s = 0.f;
    if (t == 0) for (int rr = 0; rr < RT; ++rr) s += s_l[0][rr];
    else if (t == 1) for (int rr = 0; rr < RT; ++rr) s += s_l[1][rr];
    else if (t >= 4 && t - 4 < A) for (int rr = 0; rr < RT; ++rr) s += s_red[rr * LR + (t - 4)];
    a.partial[(size_t)blockIdx.x * (4 + AP) + t] = s;
  }
  // ---- dZ2 = (dOut . W3^T) * act'(h2): thread = hidden column n, loop over the RT rows ----
  for (int n = t; n < H; n += 256) {
    float acc[RT];
#pragma unroll
    for (int rr = 0; rr < RT; ++rr) acc[rr] = 0.f;
    for (int k = 0; k < A; ++k) {
      const float wk = s_w3a[n * A + k];
#pragma unroll
      for (int rr = 0; rr < RT; ++rr) acc[rr] += s_do[rr * LR + k] * wk;
    }
    const float wc = s_w3c[n];
#pragma unroll
    for (int rr = 0; rr < RT; ++rr) {
      if (row0 + rr < a.n) {
        const float ha = s_h2a[rr * HS + n], hc = s_h2c[rr * HS + n];
        a.dz2a[(size_t)(row0 + rr) * H + n] = a.use_tanh ? acc[rr] * (1.f - ha * ha) : (ha > 0.f ? acc[rr] : 0.f);
        a.dz2c[(size_t)(row0 + rr) * H + n] = hc > 0.f ? s_do[rr * LR + A] * wc : 0.f;
      }
    }
  }
}

static size_t head_smem_bytes(int LR, int H, int A) { const int RT = 256 / LR; return sizeof(float) * ((size_t)2 * RT * (H + 4) + (size_t)H * A + H + 2 * RT * LR); }

int32_t head_launch(const HeadArgs& a, bool loss, hipStream_t stream) {
  // lanes per row: one per action dimension + one for the value; a whole wave per row covers up to 63 action dimensions
  MPPO_REQUIRE(a.A + 1 <= 64 && (a.H % 4) == 0, "head kernel: A = %d must be <= 63 and H %% 4 == 0", a.A);
  const int LR = a.A + 1 <= 32 ? 32 : 64;  // 8 (4) rows per workgroup: 160 (320) workgroups at mb = 1280
  const int RT = 256 / LR;
  const size_t smem = head_smem_bytes(LR, a.H, a.A);
  MPPO_REQUIRE(smem <= 160 * 1024, "head kernel: %zu bytes of LDS needed (H = %d, A = %d too large)", smem, a.H, a.A);
  static thread_local size_t attr_for = 0;
  if (smem > 64 * 1024 && attr_for < smem) {
    MPPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(head_kernel<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    MPPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(head_kernel<32, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    MPPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(head_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    MPPO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(head_kernel<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_for = smem;
  }
  dim3 grid(cdiv(a.n, RT));
  if (LR == 64) {
    if (loss) hipLaunchKernelGGL((head_kernel<64, true>), grid, dim3(256), smem, stream, a);
    else hipLaunchKernelGGL((head_kernel<64, false>), grid, dim3(256), smem, stream, a);
  } else {
    if (loss) hipLaunchKernelGGL((head_kernel<32, true>), grid, dim3(256), smem, stream, a);
    else hipLaunchKernelGGL((head_kernel<32, false>), grid, dim3(256), smem, stream, a);
  }
  MPPO_CHECK_LAUNCH("head_kernel");
  return MPPO_OK;
}

// ------------------------------------------------------------------------------------------------
// grad = sum of split-K slabs (+ log_std gradient from the loss partials); loss4 from the partials.
// ------------------------------------------------------------------------------------------------
constexpr int kNormBlocks = 256;
static_assert(kSqSlots == 2 * kNormBlocks, "adam_kernel adds kSqSlots partials; the reduce / sumsq kernels fill the first half");

// grad = sum of the split-K slabs (one float4 per slab per thread, every load in flight at once), the log_std gradient
// and loss4 from the head kernel's per-workgroup partials, and the per-workgroup sums of squares of the result in
// `sq_partial[kNormBlocks]` (consumed by the clip when no all-reduce sits between this kernel and Adam).
// Launched with exactly kNormBlocks workgroups of 256 threads; requires P <= kNormBlocks * 256 * 4 * kReduceIter.
constexpr int kReduceIter = 4;
struct PadList { int n, off[kMaxPads], cnt[kMaxPads]; };  // alignment words of the flat layout: no GEMM writes them into the slabs
__device__ __forceinline__ bool is_pad(const PadList& pl, size_t e) {
  bool r = false;
  for (int k = 0; k < pl.n; ++k) r = r || (e >= (size_t)pl.off[k] && e < (size_t)(pl.off[k] + pl.cnt[k]));
  return r;
}
__global__ void __launch_bounds__(256) grad_reduce_kernel(size_t P, int ksplit, size_t slab_stride, const float* __restrict__ slabs, int ls_off, int A,
                                                          int AP, int nblk, const float* __restrict__ partial, const float* __restrict__ log_std,
                                                          float ent_coef, float vf_coef, float ent_weight, float* __restrict__ grad,
                                                          float* __restrict__ loss4, float* __restrict__ sq_partial, PadList pl) {
  __shared__ float red[4];
  __shared__ float s_part[16][72];  // (4 + AP columns: up to 63 action dimensions on the layer-wise path)
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < kReduceIter; ++it) {
    const size_t i = ((size_t)(it * kNormBlocks + blockIdx.x) * 256 + threadIdx.x) * 4;
    if (i >= P) continue;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (i + 3 < P) {
      for (int k = 0; k < ksplit; ++k) {
        const float4 q = *reinterpret_cast<const float4*>(slabs + (size_t)k * slab_stride + i);
        v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
      }
    } else {
      for (int c = 0; c < 4; ++c) if (i + c < P) for (int k = 0; k < ksplit; ++k) v[c] += slabs[(size_t)k * slab_stride + i + c];
    }
    for (int c = 0; c < 4; ++c) {
      const size_t e = i + c;
      if (e < P && !(e >= (size_t)ls_off && e < (size_t)ls_off + A)) {  // log_std: below
        const float g = is_pad(pl, e) ? 0.f : v[c];
        grad[e] = g; sq += g * g;
      }
    }
  }
  if (blockIdx.x == kNormBlocks - 1) {
    // column sums of partial[nblk][4+AP] in two levels: thread (g = t/16 .. , c = t%16 ..) adds rows g, g+16, ... of column c
    const int W = 4 + AP;
    for (int c = threadIdx.x & 15; c < W; c += 16) {
      float s0 = 0.f;
      for (int k = threadIdx.x >> 4; k < nblk; k += 16) s0 += partial[(size_t)k * W + c];
      s_part[threadIdx.x >> 4][c] = s0;
    }
    __syncthreads();
    __shared__ float s_col[72];
    if (threadIdx.x < W) {
      float s0 = 0.f;
      for (int gq = 0; gq < 16; ++gq) s0 += s_part[gq][threadIdx.x];
      s_col[threadIdx.x] = s0;
    }
    __syncthreads();
    if (threadIdx.x < A) {
      const float d = s_col[4 + threadIdx.x] - ent_coef * ent_weight;
      grad[ls_off + threadIdx.x] = d;
      sq += d * d;
    }
    if (threadIdx.x == 0 && loss4) {
      float sl = 0.f;
      for (int a = 0; a < A; ++a) sl += log_std[a];
      const float ent = (0.5f * (float)A * (1.f + kLog2Pi) + sl) * ent_weight;
      loss4[0] = s_col[0] + vf_coef * s_col[1] - ent_coef * ent;
      loss4[1] = s_col[1];
      loss4[2] = s_col[0];
      loss4[3] = ent;
    }
  }
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0 && sq_partial) sq_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  if (blockIdx.x == 0 && sq_partial) sq_partial[kNormBlocks + threadIdx.x] = 0.f;  // the clip adds kSqSlots = 2 x kNormBlocks partials
}

// ------------------------------------------------------------------------------------------------
// per-minibatch advantage statistics (train.py:235), float64 sums so that ranks can be added exactly
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) adv_sums_kernel(const float* __restrict__ adv, const int* __restrict__ idx, int mb, double* __restrict__ sums) {
  __shared__ double red[4][2];
  const int k = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < mb; i += blockDim.x) {
    const double x = (double)adv[idx[(size_t)k * mb + i]];
    s1 += x;
    s2 += x * x;
  }
  s1 = wave_sum_f64(s1);
  s2 = wave_sum_f64(s2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = s1; red[wave][1] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    sums[2 * k] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    sums[2 * k + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
  }
}

__global__ void __launch_bounds__(256) adv_finalize_kernel(const double* __restrict__ sums, int n, double count, float* __restrict__ stats) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double mean = sums[2 * k] / count;
  double var = sums[2 * k + 1] / count - mean * mean;
  var = var > 0.0 ? var : 0.0;
  stats[2 * k] = (float)mean;
  stats[2 * k + 1] = (float)(1.0 / (sqrt(var) + 1e-8));
}

// ------------------------------------------------------------------------------------------------
// clip_by_global_norm + adam + apply (train.py:115-124,248); lr schedule train.py:98-101
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sumsq_kernel(size_t P, const float* __restrict__ g, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (size_t)gridDim.x * blockDim.x) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  if (blockIdx.x == 0) partial[kNormBlocks + threadIdx.x] = 0.f;  // the clip adds kSqSlots = 2 x kNormBlocks partials
}

// One finished 32 x 32 tile (rows k0.., columns n0.. of W1 or W2 of network `netc`, new parameter values in LDS) into the shadow
// copies; called by all 256 threads of the workgroup after a barrier.
__device__ __forceinline__ void shadow_write_tile(const ShadowRef& sh, const float (&tile)[32][33], bool is_w1, int netc, int k0, int n0) {
  const int t = threadIdx.x;
  if (!is_w1) {
    // w2t[n0 + nr][k0 + 4 kq ..]: four consecutive k of one column, eight threads per 128-byte row segment
    float* T = sh.w2t + (netc ? (((size_t)sh.H * sh.H + 3) & ~(size_t)3) : 0);
    const int nr = t >> 3, kq = (t & 7) * 4;
    wt_store(T, (size_t)(n0 + nr) * sh.H + k0 + kq, make_float4(tile[kq][nr], tile[kq + 1][nr], tile[kq + 2][nr], tile[kq + 3][nr]));
  }
  if (sh.frag) {
    // fragment block (S = k0 / 32, w = n0 / 32): thread = (lane = t / 4, part = t % 4 = 2 tau + g), four values c = 0..3
    const int lane = t >> 2, part = t & 3, tau = part >> 1, gg = part & 1, kq = lane >> 4, j = lane & 15;
    const int KP = (sh.O + 31) & ~31, tn = sh.H / 32;
    unsigned short* F = sh.frag + (size_t)netc * sh.frag_net_stride + (is_w1 ? 0 : (size_t)KP * sh.H);
    {
      const bf16x4 v = pack_bf16x4(tile[16 * gg + 4 * kq][2 * j + tau], tile[16 * gg + 4 * kq + 1][2 * j + tau], tile[16 * gg + 4 * kq + 2][2 * j + tau],
                                   tile[16 * gg + 4 * kq + 3][2 * j + tau]);
      *reinterpret_cast<bf16x4*>(F + ((size_t)(k0 >> 5) * tn + (n0 >> 5)) * 1024 + (size_t)lane * kFragLaneElems + (size_t)tau * kFragTileElems + 4 * gg) = v;
    }
    if (!is_w1) {  // the transposed matrix B'(k', n') = W2[n'][k']: block (S = n0 / 32, w = k0 / 32), element from tile[n' - k0][k' - n0]
      unsigned short* FT_ = F + (size_t)sh.H * sh.H;
      const bf16x4 v = pack_bf16x4(tile[2 * j + tau][16 * gg + 4 * kq], tile[2 * j + tau][16 * gg + 4 * kq + 1], tile[2 * j + tau][16 * gg + 4 * kq + 2],
                                   tile[2 * j + tau][16 * gg + 4 * kq + 3]);
      *reinterpret_cast<bf16x4*>(FT_ + ((size_t)(n0 >> 5) * tn + (k0 >> 5)) * 1024 + (size_t)lane * kFragLaneElems + (size_t)tau * kFragTileElems + 4 * gg) = v;
    }
  }
}

// PEER (several ranks, peer.h): `g` / `partial` are the reduced gradient and its sums of squares in this rank's exchange buffer.  The
// launch first reduces and broadcasts this rank's slice of the gradient (phase bit 1: workgroups < nA), then waits until every slice
// has arrived (phase bit 2) and applies the update.  One launch does both on the GPU (no workgroup waits for a workgroup of its own
// launch that has not had its turn: phase A comes first in every workgroup); the CPU emulator, which runs workgroups one after
// another, launches the two phases separately.
// The PEER instantiation is held to 64 registers: its workgroups WAIT (phase B) while other kernels must still find room on the same
// CUs when several ranks share one GPU - a weight-gradient workgroup needs 432 of a SIMD's 512 registers per lane (csrc/peer.h).
template <bool PEER>
__global__ void __launch_bounds__(256, PEER ? 8 : 1) adam_kernel(size_t P, float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ g, const float* __restrict__ partial, const int* __restrict__ count_base,
                                                   int step_offset, mppo_adam_cfg_t c, ShadowRef sh, PeerStep ps, int phase) {
  if (PEER) {
    if ((phase & 1) && (int)blockIdx.x < ps.v.nA) peer_reduce_piece(ps.v, ps.epoch[0] + ps.step + 1, (int)blockIdx.x, !(phase & 4), (phase & 4) != 0);
    if (!(phase & 2)) return;
  }
  // ---- addresses depend on the kernel arguments only: the four arrays are REQUESTED FIRST, and the clip scale, learning rate and
  // bias corrections (a wave reduction, a square root, two powf: ~0.7 us of scalar-ish work on a lone wave) are computed while they
  // fly.  (In program order the other way round, the loads left only after all of that: two memory latencies and the powf in series.)
  const bool tile_wg = sh.w2t && blockIdx.x >= sh.flat_blocks;
  // tile workgroup: W2 (and, for a bf16 network, W1) of the actor / critic, one 32 x 32 tile (k x n) per workgroup; thread =
  // (row k = t / 8, four columns n = 4 (t % 8) ..): one float4 per array, exactly like a flat thread
  int tb = 0, tn = 1, w2_tiles = 1, netc = 0, k0 = 0, n0 = 0;
  bool is_w1 = false, row_on = true;
  const int kr = threadIdx.x >> 3, nq = (threadIdx.x & 7) * 4;
  size_t i = 0;       // first of this thread's four parameters
  bool have4 = false;  // a whole float4 inside [0, P)
  if (tile_wg) {
    tb = blockIdx.x - sh.flat_blocks; tn = sh.H / 32; w2_tiles = tn * tn;
    is_w1 = tb >= 2 * w2_tiles;
    const int kt1 = (sh.O + 31) / 32, per_net = is_w1 ? kt1 * tn : w2_tiles, tb2 = is_w1 ? tb - 2 * w2_tiles : tb;
    netc = tb2 / per_net;
    const int tt = tb2 - netc * per_net;
    k0 = 32 * (tt / tn); n0 = 32 * (tt % tn);
    const int krows = is_w1 ? sh.O : sh.H;  // rows of the matrix (W1's last tile is ragged)
    const size_t base = (size_t)(is_w1 ? (netc ? sh.c_w1 : sh.a_w1) : (netc ? sh.c_w2 : sh.a_w2));
    row_on = k0 + kr < krows;
    i = base + (size_t)(row_on ? k0 + kr : 0) * sh.H + n0 + nq;
    have4 = true;
  } else {
    // four parameters per thread (the flat layout is a whole number of float4: ppo_layout.h): a quarter of the workgroups to dispatch.
    // The flat workgroups enumerate everything EXCEPT the ranges the tile workgroups own, so the grid is no larger than without
    // the shadow copies; all boundaries are multiples of four floats (16-byte aligned tensors)
    size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int r = 0; r < 4; ++r) i4 += ((int)(r < sh.nskip) & (int)(i4 >= sh.skip_start4[r])) ? sh.skip_len4[r] : 0u;  // (static indices, no branches: one batch of scalar loads)
    i = 4 * i4;
    have4 = i + 3 < P;
  }
  float4 gq = make_float4(0.f, 0.f, 0.f, 0.f), mq = gq, vq = gq, pq = gq;
  if (have4) {
    if (!PEER) gq = *reinterpret_cast<const float4*>(g + i);
    mq = *reinterpret_cast<const float4*>(m + i); vq = *reinterpret_cast<const float4*>(v + i);
    pq = *reinterpret_cast<const float4*>(p + i);
  }
  // every wave adds the same kSqSlots (= 512) partials in the same order (eight per lane, fixed reduction tree):
  // bitwise-identical clip scale everywhere without a second pass
  const int ln = threadIdx.x & 63;
  float pr[8];
  if (PEER) {
    // the reduced gradient and its sums of squares arrive tagged with the step's epoch (peer.h): every thread takes its own four floats
    // when their tags say so - no flag, no barrier (the moments and the parameters are in flight meanwhile).  Ranks sharing a GPU have
    // waited for the red_done flags in a kernel of their own: the tags are there already.
    const int epoch = ps.epoch[0] + ps.step + 1;
    if (have4) gq = peer_load_reduced4(ps.v, g, i, epoch);
    const int nslots = ps.v.world * ps.v.nA;  // (the other slots are never written: they count as 0)
#pragma unroll
    for (int k = 0; k < 8; ++k) pr[k] = ln + 64 * k < nslots ? peer_load_reduced1(ps.v, g, P + (size_t)(ln + 64 * k), epoch) : 0.f;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) pr[k] = partial[ln + 64 * k];
  }
  const float ss = wave_sum(((pr[0] + pr[1]) + (pr[2] + pr[3])) + ((pr[4] + pr[5]) + (pr[6] + pr[7])));
  const float norm = sqrtf(ss);
  const float scale = norm < c.max_grad_norm ? 1.f : c.max_grad_norm / norm;
  const int count = count_base[0] + step_offset;
  float lr = c.lr;
  if (c.anneal) lr = c.lr * (1.f - (float)(count / c.sched_div) / (float)c.num_updates);
  const float t = (float)(count + 1);
  const float bc1 = 1.f - powf(c.b1, t), bc2 = 1.f - powf(c.b2, t);
  const float gs[4] = {gq.x * scale, gq.y * scale, gq.z * scale, gq.w * scale};
  const float mo[4] = {mq.x, mq.y, mq.z, mq.w}, vo[4] = {vq.x, vq.y, vq.z, vq.w}, po[4] = {pq.x, pq.y, pq.z, pq.w};
  float mn[4], vn[4], pn[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mn[k] = c.b1 * mo[k] + (1.f - c.b1) * gs[k];
    vn[k] = c.b2 * vo[k] + (1.f - c.b2) * gs[k] * gs[k];
    pn[k] = po[k] - lr * (mn[k] / bc1) / (sqrtf(vn[k] / bc2) + c.eps);
  }
  if (tile_wg) {
    // the tile goes into the shadow copies through LDS - W2^T as floats (128-byte row segments both ways; a flat thread would
    // scatter four 4-byte stores), bf16 fragments as one contiguous 2 KB block per copy
    __shared__ float tile[32][33];
#pragma unroll
    for (int q = 0; q < 4; ++q) tile[kr][nq + q] = row_on ? pn[q] : 0.f;  // rows past the matrix are the zero padding of the fragments
    if (row_on) {
      wt_store(m, i, make_float4(mn[0], mn[1], mn[2], mn[3]));  // like the flat path: next read by the next step's Adam
      wt_store(v, i, make_float4(vn[0], vn[1], vn[2], vn[3]));
      wt_store(p, i, make_float4(pn[0], pn[1], pn[2], pn[3]));
    }
    __syncthreads();
    shadow_write_tile(sh, tile, is_w1, netc, k0, n0);
    return;
  }
  if (i >= P) return;
  if (have4) {
    wt_store(m, i, make_float4(mn[0], mn[1], mn[2], mn[3]));  // the moments are next read by the next step's Adam, two kernels and ~40 MB of traffic later
    wt_store(v, i, make_float4(vn[0], vn[1], vn[2], vn[3]));
    wt_store(p, i, make_float4(pn[0], pn[1], pn[2], pn[3]));
  } else {
    for (size_t e = i; e < P; ++e) {  // (a parameter count that is not a multiple of four: the stand-alone entry point only)
      const float gi = (PEER ? sys_load_f32(g + 2 * e) : g[e]) * scale;  // (with the exchange P is a multiple of four: never reached)
      const float mi = c.b1 * m[e] + (1.f - c.b1) * gi;
      const float vi = c.b2 * v[e] + (1.f - c.b2) * gi * gi;
      m[e] = mi; v[e] = vi;
      p[e] = p[e] - lr * (mi / bc1) / (sqrtf(vi / bc2) + c.eps);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (philox.h): action noise, permutation keys
// ------------------------------------------------------------------------------------------------
// `ctr` (optional, device int): the engine's update index; it takes the place of the high counter word so that a
// captured launch draws a fresh sub-stream every replay.
__global__ void __launch_bounds__(256) normal_fill_kernel(unsigned long long seed, unsigned long long stream_id, const int* __restrict__ ctr, size_t n,
                                                          float* __restrict__ out) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (4 * q >= n) return;
  const U4 r = philox4x32((unsigned)q, ctr ? (unsigned)ctr[0] : (unsigned)(q >> 32), (unsigned)stream_id, (unsigned)(stream_id >> 32), (unsigned)seed, (unsigned)(seed >> 32));
  const float u0 = ((float)r.x + 0.5f) * 2.3283064365386963e-10f, u1 = ((float)r.y + 0.5f) * 2.3283064365386963e-10f;
  const float u2 = ((float)r.z + 0.5f) * 2.3283064365386963e-10f, u3 = ((float)r.w + 0.5f) * 2.3283064365386963e-10f;
  const float ra = sqrtf(-2.f * logf(fminf(fmaxf(u0, 1e-10f), 1.f))), rb = sqrtf(-2.f * logf(fminf(fmaxf(u2, 1e-10f), 1.f)));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u1, &s0, &c0);
  sincosf(6.283185307179586f * u3, &s1, &c1);
  const float z[4] = {ra * c0, ra * s0, rb * c1, rb * s1};
  for (int k = 0; k < 4; ++k) if (4 * q + k < n) out[4 * q + k] = z[k];
}

__global__ void __launch_bounds__(256) perm_keys_kernel(unsigned long long seed, unsigned long long stream_id, const int* __restrict__ ctr, int B,
                                                        unsigned* __restrict__ keys, int* __restrict__ vals) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (4 * q >= B) return;
  const U4 r = philox4x32((unsigned)q, ctr ? (unsigned)ctr[0] : 0u, (unsigned)stream_id, (unsigned)(stream_id >> 32) ^ 0x5045524Du, (unsigned)seed, (unsigned)(seed >> 32));
  const unsigned z[4] = {r.x, r.y, r.z, r.w};
  for (int k = 0; k < 4; ++k) if (4 * q + k < B) { keys[4 * q + k] = z[k]; vals[4 * q + k] = 4 * q + k; }
}

// E permutations' worth of keys in one launch: epoch e draws the stream `stream_id0 + e` exactly as perm_keys_kernel would, and its
// keys carry e above bit 32, so that ONE stable sort of all E B pairs leaves every epoch's block sorted by its own keys
__global__ void __launch_bounds__(256) perm_keys_batch_kernel(unsigned long long seed, unsigned long long stream_id0, const int* __restrict__ ctr, int B,
                                                              unsigned long long* __restrict__ keys, int* __restrict__ vals) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x, e = blockIdx.y;
  if (4 * q >= B) return;
  const unsigned long long stream_id = stream_id0 + (unsigned long long)e;
  const U4 r = philox4x32((unsigned)q, ctr ? (unsigned)ctr[0] : 0u, (unsigned)stream_id, (unsigned)(stream_id >> 32) ^ 0x5045524Du, (unsigned)seed, (unsigned)(seed >> 32));
  const unsigned z[4] = {r.x, r.y, r.z, r.w};
  for (int k = 0; k < 4; ++k)
    if (4 * q + k < B) { keys[(size_t)e * B + 4 * q + k] = ((unsigned long long)e << 32) | z[k]; vals[(size_t)e * B + 4 * q + k] = 4 * q + k; }
}

int32_t perm_fill_keys_batch(unsigned long long seed, unsigned long long stream_id0, const int* ctr, int B, int E, unsigned long long* keys, int* vals,
                             hipStream_t stream) {
  hipLaunchKernelGGL(perm_keys_batch_kernel, dim3(cdiv(cdiv(B, 4), 256), E), dim3(256), 0, stream, seed, stream_id0, ctr, B, keys, vals);
  MPPO_CHECK_LAUNCH("perm_keys_batch_kernel");
  return MPPO_OK;
}

int32_t perm_fill_keys(unsigned long long seed, unsigned long long stream_id, const int* ctr, int B, unsigned* keys, int* vals, hipStream_t stream) {
  hipLaunchKernelGGL(perm_keys_kernel, dim3(cdiv(cdiv(B, 4), 256)), dim3(256), 0, stream, seed, stream_id, ctr, B, keys, vals);
  MPPO_CHECK_LAUNCH("perm_keys_kernel");
  return MPPO_OK;
}

int32_t normal_fill_ctr(unsigned long long seed, unsigned long long stream_id, const int* ctr, size_t n, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(normal_fill_kernel, dim3(cdiv((long)((n + 3) / 4), 256)), dim3(256), 0, stream, seed, stream_id, ctr, n, out);
  MPPO_CHECK_LAUNCH("normal_fill_kernel");
  return MPPO_OK;
}

// ------------------------------------------------------------------------------------------------
// stage launchers shared with the engine
// ------------------------------------------------------------------------------------------------
static GemmProb fwd_prob(const float* A, int lda, const int* gather, int M, int K, const float* W, int N, const float* bias, int act, float* C, int ldc) {
  GemmProb p{};
  p.A = A; p.lda = lda; p.gather = gather; p.M = M; p.K = K; p.B = W; p.ldb = N; p.N = N; p.bias = bias; p.act = act; p.C = C; p.ldc = ldc;
  return p;
}

// hidden layers of actor + critic on n rows of `obs` (optionally gathered): the activations of every layer of both networks into the FwdBufs
int32_t mlp_hidden_forward(const mppo_net_t& net, const float* params, int n, const float* obs, int obs_ld, const int* gather, const FwdBufs& fb,
                           float* xcopy, hipStream_t stream) {
  const ParamLayout L = param_layout(net);
  const int H = net.H, act_a = net.use_tanh ? ACT_TANH : ACT_RELU;
  GemmBatch gb{};
  gb.count = 2; gb.ksplit = 1;
  gb.p[0] = fwd_prob(obs, obs_ld, gather, n, net.O, params + L.a_w[0], H, params + L.a_b[0], act_a, fb.ha[0], H);
  gb.p[1] = fwd_prob(obs, obs_ld, gather, n, net.O, params + L.c_w[0], H, params + L.c_b[0], ACT_RELU, fb.hc[0], H);
  gb.p[0].a_copy = xcopy;  // (row stride obs_ld; only the actor problem writes it)
  MPPO_TRY(gemm_launch(gb, 0, 0, EPI_BIAS_ACT, net.bf16, stream));
  for (int l = 1; l < L.nl; ++l) {  // `for feat in self.features[:-1]` (train.py:62-67)
    gb.p[0] = fwd_prob(fb.ha[l - 1], H, nullptr, n, H, params + L.a_w[l], H, params + L.a_b[l], act_a, fb.ha[l], H);
    gb.p[1] = fwd_prob(fb.hc[l - 1], H, nullptr, n, H, params + L.c_w[l], H, params + L.c_b[l], ACT_RELU, fb.hc[l], H);
    MPPO_TRY(gemm_launch(gb, 0, 0, EPI_BIAS_ACT, net.bf16, stream));
  }
  return MPPO_OK;
}

static HeadArgs head_args(const mppo_net_t& net, const float* params, int n, const FwdBufs& fb) {
  const ParamLayout L = param_layout(net);  // (h2 / w3 / b3: the last hidden layer and the output layer, whatever the depth)
  HeadArgs a{};
  a.n = n; a.A = net.A; a.AP = fb.AP; a.DP = fb.AP + 4; a.H = net.H; a.use_tanh = net.use_tanh;
  a.h2a = fb.h2a; a.h2c = fb.h2c; a.w3a = params + L.a_w3; a.b3a = params + L.a_b3; a.w3c = params + L.c_w3; a.b3c = params + L.c_b3;
  a.log_std = params + L.log_std;
  return a;
}

// full policy step on n rows: hidden layers, heads, sample + log-prob (noise may be null: value only)
int32_t policy_forward(const mppo_net_t& net, const float* params, int n, const float* obs, int obs_ld, const FwdBufs& fb, const float* noise, float* action,
                       float* log_prob, float* value, float* mean_out, hipStream_t stream) {
  static const char* nofuse = MPPO_EXPERIMENT_ENV("MPPO_NO_FUSED");  // A/B switch for measurements
  if (fused_rollout_supported(net, obs, obs_ld) && !(nofuse && nofuse[0] == '1'))
    return fused_policy_forward(net, params, n, obs, obs_ld, noise, action, log_prob, value, noise ? mean_out : nullptr, fb.AP, stream, fb.frag, fb.frag_net_stride);  // one launch
  MPPO_TRY(mlp_hidden_forward(net, params, n, obs, obs_ld, nullptr, fb, nullptr, stream));
  HeadArgs a = head_args(net, params, n, fb);
  a.noise = noise; a.action = action; a.log_prob = log_prob; a.value = value; a.mean_out = mean_out;
  if (!noise) {  // value-only call (bootstrap, train.py:182): sample outputs go to the scratch `mean` buffer and are ignored
    a.noise = nullptr; a.action = nullptr;
  }
  return head_launch(a, false, stream);
}

// Everything of a minibatch step that is local to a row: forward of both networks, loss terms, d(loss)/d(outputs), dZ2, dZ1
// (one fused launch where supported, k_fused.hip; otherwise layer-wise GEMMs + the head kernel).  Leaves h1, h2, dZ2, dZ1,
// dOut, xmb and the loss partials in the GradBufs; *nblk_out = number of partial rows written.
int32_t minibatch_rowpass(const mppo_net_t& net, const float* params, const mppo_batch_t& batch, const int* idx, int mb, const float* adv_stat,
                          float inv_count, const mppo_loss_cfg_t& lc, const GradBufs& gbuf, int* nblk_out, bool* fused_out, hipStream_t stream, const XPre* pre = nullptr) {
  const ParamLayout L = param_layout(net);
  const int H = net.H;
  const int act_a = net.use_tanh ? ACT_TANH : ACT_RELU;
  MPPO_REQUIRE(batch.obs_ld == net.OP, "minibatch_grad: obs_ld (%d) must equal the padded observation width OP (%d)", batch.obs_ld, net.OP);
  static const char* nofuse = MPPO_EXPERIMENT_ENV("MPPO_NO_FUSED");  // A/B switch for measurements
  const bool fused = fused_supported(net, batch) && !(nofuse && nofuse[0] == '1');
  if (fused_out) *fused_out = fused;
  if (fused) {
    MPPO_TRY(fused_forward_backward(net, params, batch, idx, mb, adv_stat, inv_count, lc, gbuf, stream, pre));
    *nblk_out = cdiv(mb, fused_rows_per_workgroup(net, mb, pre != nullptr));  // one row of loss partials per row-pass workgroup
    return MPPO_OK;
  }
  MPPO_REQUIRE(!net.bf16, "minibatch_grad: bf16 products need the fused kernels and this geometry (O=%d A=%d H=%d: e.g. an observation too wide for the weight-gradient "
               "tile table) runs the layer-wise path, which is float only", net.O, net.A, net.H);
  MPPO_TRY(mlp_hidden_forward(net, params, mb, batch.obs, batch.obs_ld, idx, gbuf.f, gbuf.xmb, stream));
  HeadArgs ha = head_args(net, params, mb, gbuf.f);
  ha.idx = idx; ha.b = batch; ha.adv_stat = adv_stat; ha.inv_count = inv_count; ha.lc = lc;
  ha.dout = gbuf.dout; ha.dz2a = gbuf.dz2a; ha.dz2c = gbuf.dz2c; ha.partial = gbuf.partial;
  MPPO_TRY(head_launch(ha, true, stream));  // heads + loss + dZ2
  *nblk_out = cdiv(mb, net.A + 1 <= 32 ? 8 : 4);  // (rows per workgroup of the head kernel)
  GemmBatch gb{};
  gb.count = 2; gb.ksplit = 1;
  for (int l = L.nl - 2; l >= 0; --l) {  // dZ_l = (dZ_{l+1} . W_{l+1}^T) * act'(h_l)
    GemmProb& a = gb.p[0]; a = GemmProb{};
    a.A = gbuf.dza[l + 1]; a.lda = H; a.M = mb; a.K = H; a.B = params + L.a_w[l + 1]; a.ldb = H; a.N = H; a.aux = gbuf.f.ha[l]; a.ldaux = H; a.act = act_a; a.C = gbuf.dza[l]; a.ldc = H;
    GemmProb& c = gb.p[1]; c = GemmProb{};
    c.A = gbuf.dzc[l + 1]; c.lda = H; c.M = mb; c.K = H; c.B = params + L.c_w[l + 1]; c.ldb = H; c.N = H; c.aux = gbuf.f.hc[l]; c.ldaux = H; c.act = ACT_RELU; c.C = gbuf.dzc[l]; c.ldc = H;
    MPPO_TRY(gemm_launch(gb, 0, 1, EPI_DACT, net.bf16, stream));
  }
  return MPPO_OK;
}

int32_t minibatch_grad(const mppo_net_t& net, const float* params, const mppo_batch_t& batch, const int* idx, int mb, const float* adv_stat,
                       float inv_count, const mppo_loss_cfg_t& lc, float* grad, float* loss4, float* sq_partial, const GradBufs& gbuf, hipStream_t stream, const XPre* pre,
                       const PeerStep* peer) {
  const ParamLayout L = param_layout(net);
  const int H = net.H, A = net.A, AP = gbuf.f.AP, DP = AP + 4, O = net.O;
  int nblk = 0;
  bool fused = false;
  MPPO_TRY(minibatch_rowpass(net, params, batch, idx, mb, adv_stat, inv_count, lc, gbuf, &nblk, &fused, stream, pre));
  MPPO_REQUIRE(fused || !peer, "minibatch_grad: the layer-wise path writes a plain gradient (the engine publishes it with peer_publish)");
  const float* xq = (pre && fused) ? pre->cur : gbuf.xmb;  // the step's observation rows, k-quad layout (the layer-wise path gathers for itself)
  const float ent_weight = (float)mb * inv_count;
  if (fused) {
    // the row pass left its outputs in k-quad layout: one launch produces the complete gradient (k_wgrad.hip)
    WgradArgs w{};
    auto wp = [&](const float* Aq, int lda, int Min, const float* Bq, int ldb, int bcols, int N, int off_w, int off_b) {
      WgradProb p{};
      p.A = Aq; p.lda = lda; p.acols = lda; p.M = Min; p.B = Bq; p.ldb = ldb; p.bcols = bcols; p.N = N; p.off_w = off_w; p.off_b = off_b;
      return p;
    };
    w.count = 6; w.grad = grad; w.sq_partial = sq_partial;
    const int sp = net.bf16 ? 0 : 1;  // float networks: hidden activations and dZ in split-pair column order (wgrad.h); x and dOut plain
    w.p[0] = wp(gbuf.f.h1a, H, H, gbuf.dz2a, H, H, H, L.a_w2, L.a_b2);  w.p[0].a_split = sp; w.p[0].b_split = sp;
    w.p[1] = wp(xq, net.OP, O, gbuf.dz1a, H, H, H, L.a_w1, L.a_b1);          w.p[1].b_split = sp;
    w.p[2] = wp(gbuf.f.h1c, H, H, gbuf.dz2c, H, H, H, L.c_w2, L.c_b2);  w.p[2].a_split = sp; w.p[2].b_split = sp;
    w.p[3] = wp(xq, net.OP, O, gbuf.dz1c, H, H, H, L.c_w1, L.c_b1);          w.p[3].b_split = sp;
    w.p[4] = wp(gbuf.f.h2a, H, H, gbuf.dout, DP, AP, A, L.a_w3, L.a_b3);    w.p[4].a_split = sp;
    w.p[5] = wp(gbuf.f.h2c, H, H, gbuf.dout + (net.bf16 ? 2 : 4) * AP, DP, 4, 1, L.c_w3, L.c_b3);  w.p[5].a_split = sp;  // column AP of the quad rows (a bf16 network's quads are 8 bytes)
    w.ls_off = L.log_std; w.A = A; w.AP = AP; w.nblk = nblk; w.partial = gbuf.partial; w.log_std = params + L.log_std;
    w.ent_coef = lc.ent_coef; w.vf_coef = lc.vf_coef; w.ent_weight = ent_weight; w.loss4 = loss4;
    w.npad = L.npad;
    for (int k = 0; k < L.npad; ++k) { w.pad_off[k] = L.pad_off[k]; w.pad_cnt[k] = L.pad_cnt[k]; }
#ifdef MPPO_EXPERIMENTS
    { static const char* e8 = getenv("MPPO_WGRAD_DBG"); if (e8 && (atoi(e8) & 8)) w.count = 4; }  // timing experiment: big problems only
#endif
    MPPO_TRY(wgrad_plan(w, mb));
    MPPO_REQUIRE(wgrad_supported(w), "minibatch_grad: weight-gradient launch not applicable (%d tiles)", w.ntiles);
    return wgrad_launch(w, net.bf16 != 0, stream, peer);  // (peer: `grad` is the rank's exchange buffer, the launch signals the peers)
  }
  {
  GemmBatch gb{};
  // weight gradients dW = H_prev^T . dZ into split-K slabs; the bias gradients (column sums of dZ) ride along.
  // dOut is [mb, DP] with d mean in columns [0,A) and d value in column AP: every operand is 16-byte aligned -> fast path.
  gb.count = 6; gb.ksplit = gbuf.ksplit; gb.slab_stride = gbuf.slab_stride;
  auto wprob = [&](const float* Aprev, int lda, const int* gather, int Min, const float* dZ, int ldz, int N, int off_w, int off_b) {
    GemmProb p{};
    p.A = Aprev; p.lda = lda; p.gather = gather; p.M = Min; p.K = mb; p.B = dZ; p.ldb = ldz; p.N = N; p.C = gbuf.slabs + off_w; p.ldc = N;
    p.bias_out = gbuf.slabs + off_b;
    return p;
  };
  // two problems per layer and network (output layers first), at most kGemmMaxProb per launch
  GemmProb all[2 * (kMaxHidden + 1)];
  int np = 0;
  all[np++] = wprob(gbuf.f.h2a, H, nullptr, H, gbuf.dout, DP, A, L.a_w3, L.a_b3);
  all[np++] = wprob(gbuf.f.h2c, H, nullptr, H, gbuf.dout + AP, DP, 1, L.c_w3, L.c_b3);
  for (int l = L.nl - 1; l >= 1; --l) {
    all[np++] = wprob(gbuf.f.ha[l - 1], H, nullptr, H, gbuf.dza[l], H, H, L.a_w[l], L.a_b[l]);
    all[np++] = wprob(gbuf.f.hc[l - 1], H, nullptr, H, gbuf.dzc[l], H, H, L.c_w[l], L.c_b[l]);
  }
  all[np++] = wprob(gbuf.xmb, net.OP, nullptr, O, gbuf.dza[0], H, H, L.a_w[0], L.a_b[0]);
  all[np++] = wprob(gbuf.xmb, net.OP, nullptr, O, gbuf.dzc[0], H, H, L.c_w[0], L.c_b[0]);
  for (int at = 0; at < np; at += kGemmMaxProb) {
    gb.count = np - at < kGemmMaxProb ? np - at : kGemmMaxProb;
    for (int k = 0; k < gb.count; ++k) gb.p[k] = all[at + k];
    MPPO_TRY(gemm_launch(gb, 1, 0, EPI_STORE, net.bf16, stream));
  }
  }
  MPPO_REQUIRE((size_t)L.total <= (size_t)kNormBlocks * 256 * 4 * kReduceIter, "minibatch_grad: %d parameters exceed the reduce kernel's range", L.total);
  PadList pl{};
  pl.n = L.npad;
  for (int k = 0; k < L.npad; ++k) { pl.off[k] = L.pad_off[k]; pl.cnt[k] = L.pad_cnt[k]; }
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(kNormBlocks), dim3(256), 0, stream, (size_t)L.total, gbuf.ksplit, gbuf.slab_stride, gbuf.slabs, L.log_std, A, AP, nblk,
                     gbuf.partial, params + L.log_std, lc.ent_coef, lc.vf_coef, ent_weight, grad, loss4, sq_partial, pl);
  MPPO_CHECK_LAUNCH("grad_reduce_kernel");
  return MPPO_OK;
}

// Builds every shadow copy from the parameters: one 32 x 32 tile per workgroup, same tile enumeration and same writer as Adam's
__global__ void __launch_bounds__(256) shadow_refresh_kernel(const float* __restrict__ params, ShadowRef sh) {
  __shared__ float tile[32][33];
  const int tb = blockIdx.x, tn = sh.H / 32, w2_tiles = tn * tn;
  const bool is_w1 = tb >= 2 * w2_tiles;
  const int kt1 = (sh.O + 31) / 32, per_net = is_w1 ? kt1 * tn : w2_tiles, tb2 = is_w1 ? tb - 2 * w2_tiles : tb;
  const int netc = tb2 / per_net, tt = tb2 - netc * per_net, k0 = 32 * (tt / tn), n0 = 32 * (tt % tn);
  const int krows = is_w1 ? sh.O : sh.H;
  const size_t base = (size_t)(is_w1 ? (netc ? sh.c_w1 : sh.a_w1) : (netc ? sh.c_w2 : sh.a_w2));
  const int kr = threadIdx.x >> 3, nq = (threadIdx.x & 7) * 4;
  const bool row_on = k0 + kr < krows;
  const float4 pq = *reinterpret_cast<const float4*>(params + base + (size_t)(row_on ? k0 + kr : 0) * sh.H + n0 + nq);
  tile[kr][nq] = row_on ? pq.x : 0.f; tile[kr][nq + 1] = row_on ? pq.y : 0.f; tile[kr][nq + 2] = row_on ? pq.z : 0.f; tile[kr][nq + 3] = row_on ? pq.w : 0.f;
  __syncthreads();
  shadow_write_tile(sh, tile, is_w1, netc, k0, n0);
}

static int shadow_tile_blocks(const ShadowRef& sh) {
  const int tn = sh.H / 32;
  return 2 * tn * tn + (sh.frag ? 2 * ((sh.O + 31) / 32) * tn : 0);
}

int32_t shadow_refresh(const mppo_net_t& net, const float* params, const GradBufs& gbuf, hipStream_t stream) {
  MPPO_REQUIRE(net.H % 32 == 0, "shadow_refresh: the shadow copies need H %% 32 == 0 (H = %d)", net.H);
  const ShadowRef sh = make_shadow_ref(net, gbuf);
  hipLaunchKernelGGL(shadow_refresh_kernel, dim3(shadow_tile_blocks(sh)), dim3(256), 0, stream, params, sh);
  MPPO_CHECK_LAUNCH("shadow_refresh_kernel");
  return MPPO_OK;
}

int32_t clip_adam(size_t P, float* params, float* m, float* v, const float* grad, const int* count_base, int step_offset, const mppo_adam_cfg_t& cfg,
                  float* ws, bool have_sumsq, hipStream_t stream, const ShadowRef* shadow, const PeerStep* peer) {
  MPPO_REQUIRE(!peer || have_sumsq, "clip_adam: with the peer exchange the sums of squares come with the reduced gradient");
  if (!have_sumsq) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(kNormBlocks), dim3(256), 0, stream, P, grad, ws);
    MPPO_CHECK_LAUNCH("sumsq_kernel");
  }
  MPPO_REQUIRE((reinterpret_cast<uintptr_t>(params) & 15) == 0 && (reinterpret_cast<uintptr_t>(m) & 15) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(grad) & 15) == 0, "clip_adam: params / m / v / grad must be 16-byte aligned");
  ShadowRef sh = shadow ? *shadow : ShadowRef{};
  MPPO_REQUIRE(!sh.w2t || sh.H % 32 == 0, "clip_adam: the W2^T shadow copies need H %% 32 == 0 (H = %d)", sh.H);
  const int tile_blocks = sh.w2t ? shadow_tile_blocks(sh) : 0;
  long owned4 = 0;  // float4 slots that tile workgroups update
  sh.nskip = 0;
  if (sh.w2t) {
    struct R { int start, len; } rs[4] = {{sh.a_w2, sh.H * sh.H}, {sh.c_w2, sh.H * sh.H}, {sh.a_w1, sh.frag ? sh.O * sh.H : 0}, {sh.c_w1, sh.frag ? sh.O * sh.H : 0}};
    for (int a = 0; a < 4; ++a) for (int b = a + 1; b < 4; ++b) if (rs[b].start < rs[a].start) { const R t_ = rs[a]; rs[a] = rs[b]; rs[b] = t_; }
    for (int a = 0; a < 4; ++a) {
      if (rs[a].len == 0) continue;
      MPPO_REQUIRE(rs[a].start % 4 == 0 && rs[a].len % 4 == 0, "clip_adam: a shadowed tensor is not float4-aligned");
      sh.skip_start4[sh.nskip] = (unsigned)(rs[a].start / 4); sh.skip_len4[sh.nskip] = (unsigned)(rs[a].len / 4); ++sh.nskip;
      owned4 += rs[a].len / 4;
    }
  }
  const int flat_blocks = cdiv((long)((P + 3) / 4) - owned4, 256);
  sh.flat_blocks = flat_blocks;
  if (peer) {
    MPPO_REQUIRE(flat_blocks + tile_blocks >= peer->v.nA, "clip_adam: %d workgroups cannot reduce %d pieces of the gradient", flat_blocks + tile_blocks, peer->v.nA);
    const dim3 gridA(peer->v.nA), gridB(flat_blocks + tile_blocks);
#ifdef MPPO_EMU  // workgroups run one after another: a workgroup of phase B would wait for a piece whose workgroup comes after it
    const int mode = 1;
#else
    const int mode = peer->mode;
#endif
    if (mode == 2) {
      // several ranks on ONE GPU: nothing that occupies more than one wave may wait, or the peer's kernels find no room beside the
      // waiting workgroups.  A one-wave kernel does each of the two waits; the reduction and the update run without waiting.
      MPPO_TRY(peer_wait_launch(*peer, 1, stream));
      hipLaunchKernelGGL(adam_kernel<true>, gridA, dim3(256), 0, stream, P, params, m, v, grad, ws, count_base, step_offset, cfg, sh, *peer, 1 | 4);
      MPPO_TRY(peer_wait_launch(*peer, 2, stream));
      hipLaunchKernelGGL(adam_kernel<true>, gridB, dim3(256), 0, stream, P, params, m, v, grad, ws, count_base, step_offset, cfg, sh, *peer, 2 | 4);
    } else if (mode == 1) {  // two launches, each waiting for itself (the emulator's form; MPPO_PEER_MODE=split on the GPU: measurements)
      hipLaunchKernelGGL(adam_kernel<true>, gridA, dim3(256), 0, stream, P, params, m, v, grad, ws, count_base, step_offset, cfg, sh, *peer, 1);
      hipLaunchKernelGGL(adam_kernel<true>, gridB, dim3(256), 0, stream, P, params, m, v, grad, ws, count_base, step_offset, cfg, sh, *peer, 2);
    } else {
      hipLaunchKernelGGL(adam_kernel<true>, gridB, dim3(256), 0, stream, P, params, m, v, grad, ws, count_base, step_offset, cfg, sh, *peer, 3);
    }
    MPPO_CHECK_LAUNCH("adam_kernel<peer>");
    return MPPO_OK;
  }
  hipLaunchKernelGGL(adam_kernel<false>, dim3(flat_blocks + tile_blocks), dim3(256), 0, stream, P, params, m, v, grad, ws, count_base, step_offset, cfg,
                     sh, PeerStep{}, 0);
  MPPO_CHECK_LAUNCH("adam_kernel");
  return MPPO_OK;
}

int32_t gae_launch(int T, int N, float gamma, float lam, const float* reward, const float* value, const unsigned char* done, const float* last_val,
                   float* adv, float* target, hipStream_t stream) {
  hipLaunchKernelGGL(gae_kernel, dim3(cdiv(N, 256)), dim3(256), 0, stream, T, N, gamma, lam, reward, value, done, last_val, adv, target);
  MPPO_CHECK_LAUNCH("gae_kernel");
  return MPPO_OK;
}

}  // namespace mppo

// =================================================================================================
// C ABI
// =================================================================================================
using namespace mppo;

static int32_t check_net(const mppo_net_t* net) {
  MPPO_REQUIRE(net, "null net");
  MPPO_REQUIRE(net->O >= 1 && net->A >= 1 && net->A <= 63 && net->H >= 4 && (net->H % 4) == 0, "unsupported network geometry O=%d A=%d H=%d (need A<=63, H%%4==0)",
               net->O, net->A, net->H);
  MPPO_REQUIRE(net->OP >= net->O && (net->OP % 4) == 0, "OP=%d must be a multiple of 4 and >= O=%d", net->OP, net->O);
  MPPO_REQUIRE(net->num_layers >= 0 && net->num_layers <= kMaxHidden, "num_layers=%d: 1 .. %d hidden layers (0 = 2)", net->num_layers, kMaxHidden);
  // (the engine says the same when it is created; a stand-alone call used to get as far as "gemm_launch: no bf16 variant ..." - tests/test_ppo_fuzz.py)
  MPPO_REQUIRE(!net->bf16 || ((net->num_layers == 0 || net->num_layers == 2) && net->H % 32 == 0 && net->H <= 256 && net->A <= 32),
               "bf16 products need the fused kernels (two hidden layers, hidden size a multiple of 32 up to 256, at most 32 actuators); O=%d A=%d H=%d layers=%d runs the "
               "layer-wise path, which is float only", net->O, net->A, net->H, net->num_layers);
  return MPPO_OK;
}

extern "C" size_t mppo_param_count(const mppo_net_t* net) { return net ? (size_t)param_layout(*net).total : 0; }

extern "C" size_t mppo_policy_ws_bytes(const mppo_net_t* net, int32_t n) { return net ? fwd_bufs_floats(*net, n) * sizeof(float) : 0; }

extern "C" int32_t mppo_policy_forward(const mppo_net_t* net, const float* params, int32_t n, const float* obs, int32_t obs_ld, const float* noise,
                                       float* action, float* log_prob, float* value, float* mean_out, void* ws, size_t ws_bytes, void* stream) {
  MPPO_TRY(check_net(net));
  MPPO_REQUIRE(params && obs && value && ws && n >= 1, "mppo_policy_forward: null argument or n < 1");
  MPPO_REQUIRE(obs_ld >= net->O, "mppo_policy_forward: obs_ld %d < O %d", obs_ld, net->O);
  MPPO_REQUIRE(!noise || (action && log_prob), "mppo_policy_forward: noise given without action/log_prob outputs");
  MPPO_REQUIRE(!mean_out || noise, "mppo_policy_forward: mean_out is only produced together with a sample (noise != NULL)");
  if (ws_bytes < mppo_policy_ws_bytes(net, n)) return fail(MPPO_ENOMEM, "mppo_policy_forward: workspace %zu < %zu bytes", ws_bytes, mppo_policy_ws_bytes(net, n));
  const FwdBufs fb = carve_fwd(*net, n, static_cast<float*>(ws));
  return policy_forward(*net, params, n, obs, obs_ld, fb, noise, action, log_prob, value, mean_out, static_cast<hipStream_t>(stream));
}

extern "C" int32_t mppo_gae(int32_t T, int32_t N, float gamma, float lam, const float* reward, const float* value, const uint8_t* done,
                            const float* last_val, float* adv, float* target, void* stream) {
  MPPO_REQUIRE(T >= 1 && N >= 1 && reward && value && done && last_val && adv && target, "mppo_gae: bad argument");
  return gae_launch(T, N, gamma, lam, reward, value, done, last_val, adv, target, static_cast<hipStream_t>(stream));
}

extern "C" int32_t mppo_minibatch_rows_per_workgroup(const mppo_net_t* net, int32_t mb, int32_t pre_gathered, int32_t* rows) {
  MPPO_REQUIRE(net && rows && mb >= 1, "mppo_minibatch_rows_per_workgroup: null argument or mb < 1");
  *rows = mppo::fused_rows_per_workgroup(*net, mb, pre_gathered != 0);
  return MPPO_OK;
}

extern "C" int32_t mppo_minibatch_path(const mppo_net_t* net, const mppo_batch_t* batch, int32_t* fused) {
  MPPO_TRY(check_net(net));
  MPPO_REQUIRE(batch && fused, "mppo_minibatch_path: null argument");
  static const char* nofuse = MPPO_EXPERIMENT_ENV("MPPO_NO_FUSED");
  *fused = (fused_supported(*net, *batch) && !(nofuse && nofuse[0] == '1')) ? 1 : 0;
  return MPPO_OK;
}

extern "C" size_t mppo_grad_ws_bytes(const mppo_net_t* net, int32_t mb) { return net ? grad_bufs_floats(*net, mb) * sizeof(float) : 0; }

extern "C" int32_t mppo_minibatch_grad(const mppo_net_t* net, const float* params, const mppo_batch_t* batch, const int32_t* idx, int32_t mb,
                                       const float* adv_stat, float inv_count, const mppo_loss_cfg_t* lc, float* grad, float* loss4, void* ws,
                                       size_t ws_bytes, void* stream) {
  MPPO_TRY(check_net(net));
  MPPO_REQUIRE(params && batch && adv_stat && lc && grad && ws && mb >= 1, "mppo_minibatch_grad: null argument or mb < 1");
  MPPO_REQUIRE(batch->obs && batch->action && batch->value && batch->log_prob && batch->adv && batch->target, "mppo_minibatch_grad: null batch field");
  MPPO_REQUIRE(batch->obs_ld >= net->O && batch->act_ld >= net->A, "mppo_minibatch_grad: leading dimensions too small");
  if (ws_bytes < mppo_grad_ws_bytes(net, mb)) return fail(MPPO_ENOMEM, "mppo_minibatch_grad: workspace %zu < %zu bytes", ws_bytes, mppo_grad_ws_bytes(net, mb));
  const GradBufs gb = carve_grad(*net, mb, static_cast<float*>(ws));
  return minibatch_grad(*net, params, *batch, idx, mb, adv_stat, inv_count, *lc, grad, loss4, nullptr, gb, static_cast<hipStream_t>(stream));
}

extern "C" int32_t mppo_minibatch_rowpass(const mppo_net_t* net, const float* params, const mppo_batch_t* batch, const int32_t* idx, int32_t mb,
                                          const float* adv_stat, float inv_count, const mppo_loss_cfg_t* lc, void* ws, size_t ws_bytes, void* stream) {
  MPPO_TRY(check_net(net));
  MPPO_REQUIRE(params && batch && adv_stat && lc && ws && mb >= 1, "mppo_minibatch_rowpass: null argument or mb < 1");
  if (ws_bytes < mppo_grad_ws_bytes(net, mb)) return fail(MPPO_ENOMEM, "mppo_minibatch_rowpass: workspace %zu < %zu bytes", ws_bytes, mppo_grad_ws_bytes(net, mb));
  const GradBufs gb = carve_grad(*net, mb, static_cast<float*>(ws));
  int nblk = 0;
  return minibatch_rowpass(*net, params, *batch, idx, mb, adv_stat, inv_count, *lc, gb, &nblk, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int32_t mppo_adv_sums(const float* adv, const int32_t* idx, int32_t nmb, int32_t mb, double* sums, void* stream) {
  MPPO_REQUIRE(adv && idx && sums && nmb >= 1 && mb >= 1, "mppo_adv_sums: bad argument");
  hipLaunchKernelGGL(adv_sums_kernel, dim3(nmb), dim3(256), 0, static_cast<hipStream_t>(stream), adv, idx, mb, sums);
  MPPO_CHECK_LAUNCH("adv_sums_kernel");
  return MPPO_OK;
}

extern "C" int32_t mppo_adv_stats_finalize(const double* sums, int32_t nmb, double count, float* stats, void* stream) {
  MPPO_REQUIRE(sums && stats && nmb >= 1 && count >= 1.0, "mppo_adv_stats_finalize: bad argument");
  hipLaunchKernelGGL(adv_finalize_kernel, dim3(cdiv(nmb, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), sums, nmb, count, stats);
  MPPO_CHECK_LAUNCH("adv_finalize_kernel");
  return MPPO_OK;
}

extern "C" size_t mppo_adam_ws_bytes(size_t) { return kSqSlots * sizeof(float); }

extern "C" int32_t mppo_clip_adam(size_t P, float* params, float* m, float* v, const float* grad, const int32_t* count_base, int32_t step_offset,
                                  const mppo_adam_cfg_t* cfg, void* ws, size_t ws_bytes, void* stream) {
  MPPO_REQUIRE(P >= 1 && params && m && v && grad && count_base && cfg && ws, "mppo_clip_adam: null argument");
  MPPO_REQUIRE(!cfg->anneal || (cfg->sched_div >= 1 && cfg->num_updates >= 1), "mppo_clip_adam: anneal needs sched_div, num_updates >= 1");
  if (ws_bytes < mppo_adam_ws_bytes(P)) return fail(MPPO_ENOMEM, "mppo_clip_adam: workspace too small");
  return clip_adam(P, params, m, v, grad, count_base, step_offset, *cfg, static_cast<float*>(ws), false, static_cast<hipStream_t>(stream));
}

// ---- the same three entry points with the W2^T shadow copies (GradBufs::w2t inside `grad_ws`) in play ----------------------------
extern "C" int32_t mppo_shadow_refresh(const mppo_net_t* net, const float* params, int32_t mb, void* grad_ws, size_t grad_ws_bytes, void* stream) {
  MPPO_TRY(check_net(net));
  MPPO_REQUIRE(params && grad_ws && mb >= 1, "mppo_shadow_refresh: null argument or mb < 1");
  if (grad_ws_bytes < mppo_grad_ws_bytes(net, mb)) return fail(MPPO_ENOMEM, "mppo_shadow_refresh: workspace %zu < %zu bytes", grad_ws_bytes, mppo_grad_ws_bytes(net, mb));
  return shadow_refresh(*net, params, carve_grad(*net, mb, static_cast<float*>(grad_ws)), static_cast<hipStream_t>(stream));
}

extern "C" int32_t mppo_minibatch_rowpass_shadow(const mppo_net_t* net, const float* params, const mppo_batch_t* batch, const int32_t* idx, int32_t mb,
                                                 const float* adv_stat, float inv_count, const mppo_loss_cfg_t* lc, void* ws, size_t ws_bytes, void* stream) {
  MPPO_TRY(check_net(net));
  MPPO_REQUIRE(params && batch && adv_stat && lc && ws && mb >= 1, "mppo_minibatch_rowpass_shadow: null argument or mb < 1");
  if (ws_bytes < mppo_grad_ws_bytes(net, mb)) return fail(MPPO_ENOMEM, "mppo_minibatch_rowpass_shadow: workspace %zu < %zu bytes", ws_bytes, mppo_grad_ws_bytes(net, mb));
  GradBufs gb = carve_grad(*net, mb, static_cast<float*>(ws));
  gb.w2t_valid = true;
  int nblk = 0;
  return minibatch_rowpass(*net, params, *batch, idx, mb, adv_stat, inv_count, *lc, gb, &nblk, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int32_t mppo_minibatch_grad_shadow(const mppo_net_t* net, const float* params, const mppo_batch_t* batch, const int32_t* idx, int32_t mb,
                                              const float* adv_stat, float inv_count, const mppo_loss_cfg_t* lc, float* grad, float* loss4, void* ws,
                                              size_t ws_bytes, void* stream) {
  MPPO_TRY(check_net(net));
  MPPO_REQUIRE(params && batch && adv_stat && lc && grad && ws && mb >= 1, "mppo_minibatch_grad_shadow: null argument or mb < 1");
  MPPO_REQUIRE(batch->obs && batch->action && batch->value && batch->log_prob && batch->adv && batch->target, "mppo_minibatch_grad_shadow: null batch field");
  MPPO_REQUIRE(batch->obs_ld >= net->O && batch->act_ld >= net->A, "mppo_minibatch_grad_shadow: leading dimensions too small");
  if (ws_bytes < mppo_grad_ws_bytes(net, mb)) return fail(MPPO_ENOMEM, "mppo_minibatch_grad_shadow: workspace %zu < %zu bytes", ws_bytes, mppo_grad_ws_bytes(net, mb));
  GradBufs gb = carve_grad(*net, mb, static_cast<float*>(ws));
  gb.w2t_valid = true;
  return minibatch_grad(*net, params, *batch, idx, mb, adv_stat, inv_count, *lc, grad, loss4, nullptr, gb, static_cast<hipStream_t>(stream));
}

static int32_t pre_args(const char* who, const mppo_net_t* net, const mppo_batch_t* batch, const int32_t* idx, int32_t mb, void* ws, size_t ws_bytes, int32_t parity,
                        GradBufs* gb) {
  MPPO_TRY(check_net(net));
  MPPO_REQUIRE(batch && batch->obs && idx && ws && mb >= 1 && (parity == 0 || parity == 1), "%s: null argument, mb < 1 or parity not 0 / 1", who);
  if (ws_bytes < mppo_grad_ws_bytes(net, mb)) return fail(MPPO_ENOMEM, "%s: workspace %zu < %zu bytes", who, ws_bytes, mppo_grad_ws_bytes(net, mb));
  MPPO_REQUIRE(fused_supported(*net, *batch), "%s: this geometry takes the layer-wise path, which gathers for itself (mppo_minibatch_path)", who);
  *gb = carve_grad(*net, mb, static_cast<float*>(ws));
  gb->w2t_valid = true;
  return MPPO_OK;
}

extern "C" int32_t mppo_gather_rows(const mppo_net_t* net, const mppo_batch_t* batch, const int32_t* idx, int32_t mb, void* grad_ws, size_t grad_ws_bytes,
                                    int32_t parity, void* stream) {
  GradBufs gb;
  MPPO_TRY(pre_args("mppo_gather_rows", net, batch, idx, mb, grad_ws, grad_ws_bytes, parity, &gb));
  return fused_gather_rows(*net, *batch, idx, mb, parity ? gb.xmb2 : gb.xmb, static_cast<hipStream_t>(stream));
}

extern "C" int32_t mppo_minibatch_rowpass_pre(const mppo_net_t* net, const float* params, const mppo_batch_t* batch, const int32_t* idx, const int32_t* idx_next,
                                              int32_t mb, const float* adv_stat, float inv_count, const mppo_loss_cfg_t* lc, void* ws, size_t ws_bytes,
                                              int32_t parity, void* stream) {
  GradBufs gb;
  MPPO_TRY(pre_args("mppo_minibatch_rowpass_pre", net, batch, idx, mb, ws, ws_bytes, parity, &gb));
  MPPO_REQUIRE(params && adv_stat && lc, "mppo_minibatch_rowpass_pre: null argument");
  const XPre pre{parity ? gb.xmb2 : gb.xmb, parity ? gb.xmb : gb.xmb2, idx_next};
  int nblk = 0;
  return minibatch_rowpass(*net, params, *batch, idx, mb, adv_stat, inv_count, *lc, gb, &nblk, nullptr, static_cast<hipStream_t>(stream), &pre);
}

extern "C" int32_t mppo_minibatch_grad_pre(const mppo_net_t* net, const float* params, const mppo_batch_t* batch, const int32_t* idx, const int32_t* idx_next,
                                           int32_t mb, const float* adv_stat, float inv_count, const mppo_loss_cfg_t* lc, float* grad, float* loss4, void* ws,
                                           size_t ws_bytes, int32_t parity, void* stream) {
  GradBufs gb;
  MPPO_TRY(pre_args("mppo_minibatch_grad_pre", net, batch, idx, mb, ws, ws_bytes, parity, &gb));
  MPPO_REQUIRE(params && adv_stat && lc && grad, "mppo_minibatch_grad_pre: null argument");
  MPPO_REQUIRE(batch->action && batch->value && batch->log_prob && batch->adv && batch->target, "mppo_minibatch_grad_pre: null batch field");
  const XPre pre{parity ? gb.xmb2 : gb.xmb, parity ? gb.xmb : gb.xmb2, idx_next};
  return minibatch_grad(*net, params, *batch, idx, mb, adv_stat, inv_count, *lc, grad, loss4, nullptr, gb, static_cast<hipStream_t>(stream), &pre);
}

extern "C" int32_t mppo_clip_adam_shadow(const mppo_net_t* net, int32_t mb, void* grad_ws, size_t grad_ws_bytes, size_t P, float* params, float* m, float* v,
                                         const float* grad, const int32_t* count_base, int32_t step_offset, const mppo_adam_cfg_t* cfg, void* ws, size_t ws_bytes,
                                         void* stream) {
  MPPO_TRY(check_net(net));
  MPPO_REQUIRE(P >= 1 && params && m && v && grad && count_base && cfg && ws && grad_ws && mb >= 1, "mppo_clip_adam_shadow: null argument");
  MPPO_REQUIRE(P == mppo_param_count(net), "mppo_clip_adam_shadow: P = %zu is not this network's parameter count (%zu)", P, mppo_param_count(net));
  MPPO_REQUIRE(!cfg->anneal || (cfg->sched_div >= 1 && cfg->num_updates >= 1), "mppo_clip_adam_shadow: anneal needs sched_div, num_updates >= 1");
  if (ws_bytes < mppo_adam_ws_bytes(P)) return fail(MPPO_ENOMEM, "mppo_clip_adam_shadow: workspace too small");
  if (grad_ws_bytes < mppo_grad_ws_bytes(net, mb)) return fail(MPPO_ENOMEM, "mppo_clip_adam_shadow: gradient workspace too small");
  const GradBufs gb = carve_grad(*net, mb, static_cast<float*>(grad_ws));
  const ShadowRef sh = make_shadow_ref(*net, gb);
  return clip_adam(P, params, m, v, grad, count_base, step_offset, *cfg, static_cast<float*>(ws), false, static_cast<hipStream_t>(stream), &sh);
}

extern "C" int32_t mppo_normal_fill(uint64_t seed, uint64_t stream_id, size_t n, float* out, void* stream) {
  MPPO_REQUIRE(out && n >= 1, "mppo_normal_fill: bad argument");
  return normal_fill_ctr(seed, stream_id, nullptr, n, out, static_cast<hipStream_t>(stream));
}
