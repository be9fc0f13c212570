// k_ppo.hip — PPO stages around the MLP GEMMs: policy sample / log-prob, GAE reverse scan,
// clipped-surrogate loss and its output gradients, per-minibatch advantage statistics,
// split-K slab reduction, global-norm clip + Adam.  Reference lines are cited per kernel.
#include <wave_ops.h>

#include <cmath>

#include "gemm.h"
#include "mppo_common.h"
#include "ppo_layout.h"

namespace mppo {

constexpr float kLog2Pi = 1.8378770664093453f;

// ------------------------------------------------------------------------------------------------
// pi.sample + pi.log_prob (train.py:158-160; distrax MultivariateNormalDiag)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sample_kernel(int n, int A, int AP, const float* __restrict__ mean, const float* __restrict__ log_std,
                                                     const float* __restrict__ noise, float* __restrict__ action, float* __restrict__ log_prob) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float ss = 0.f, sl = 0.f;
  for (int a = 0; a < A; ++a) {
    const float ls = log_std[a];
    const float mu = mean[(size_t)i * AP + a];
    const float act = mu + expf(ls) * noise[(size_t)i * A + a];
    action[(size_t)i * A + a] = act;
    const float z = (act - mu) * expf(-ls);
    ss += z * z;
    sl += ls;
  }
  log_prob[i] = -0.5f * ss - sl - 0.5f * (float)A * kLog2Pi;
}

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
// _loss_fn on the network outputs (train.py:223-243) and d(loss)/d(mean, value, log_std).
// One thread per minibatch row; per-workgroup partial sums go to `partial[blk][4 + AP]`:
//   [0] sum -min(ratio*g, clip(ratio)*g) * w   [1] sum 0.5*max((v-t)^2,(vc-t)^2) * w   [4+a] d log_std[a]
// with w = inv_count.  Tie / clip-boundary gradient conventions as in oracle/ppo_oracle.py.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) loss_kernel(int mb, int A, int AP, const int* __restrict__ idx, const float* __restrict__ mean,
                                                   const float* __restrict__ vnew, const float* __restrict__ log_std, mppo_batch_t b,
                                                   const float* __restrict__ adv_stat, float inv_count, mppo_loss_cfg_t lc,
                                                   float* __restrict__ dmean, float* __restrict__ dv, float* __restrict__ partial) {
  __shared__ float red[4][40];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = i < mb;
  float la = 0.f, lv = 0.f, dlogp = 0.f;
  long row = 0;
  if (on) {
    row = idx ? idx[i] : i;
    float ss = 0.f, sl = 0.f;
    for (int a = 0; a < A; ++a) {
      const float ls = log_std[a];
      const float z = (b.action[row * b.act_ld + a] - mean[(size_t)i * AP + a]) * expf(-ls);
      ss += z * z;
      sl += ls;
    }
    const float logp = -0.5f * ss - sl - 0.5f * (float)A * kLog2Pi;
    const float ratio = expf(logp - b.log_prob[row]);
    const float g = (b.adv[row] - adv_stat[0]) * adv_stat[1];
    const float la1 = ratio * g;
    const float la2 = fminf(fmaxf(ratio, 1.f - lc.clip_eps), 1.f + lc.clip_eps) * g;
    la = -fminf(la1, la2) * inv_count;
    const bool unclipped = (ratio >= 1.f - lc.clip_eps) && (ratio <= 1.f + lc.clip_eps);
    dlogp = (unclipped || la1 < la2) ? -g * ratio * inv_count : 0.f;
    const float v = vnew[i], ov = b.value[row], tg = b.target[row];
    const float vc = ov + fminf(fmaxf(v - ov, -lc.clip_eps), lc.clip_eps);
    const float vl1 = (v - tg) * (v - tg), vl2 = (vc - tg) * (vc - tg);
    lv = 0.5f * fmaxf(vl1, vl2) * inv_count;
    const bool vin = fabsf(v - ov) <= lc.clip_eps;
    dv[i] = (vin || vl1 > vl2) ? (v - tg) * inv_count * lc.vf_coef : 0.f;
  }
  // block reduction of la, lv and the A log_std gradients
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float r_la = wave_sum(la), r_lv = wave_sum(lv);
  if (lane == 0) { red[wave][0] = r_la; red[wave][1] = r_lv; }
  for (int a = 0; a < AP; ++a) {
    float dm = 0.f, dls = 0.f;
    if (on && a < A) {
      const float inv_std = expf(-log_std[a]);
      const float z = (b.action[row * b.act_ld + a] - mean[(size_t)i * AP + a]) * inv_std;
      dm = dlogp * z * inv_std;
      dls = dlogp * (z * z - 1.f);
    }
    if (on) dmean[(size_t)i * AP + a] = dm;
    const float s = wave_sum(dls);
    if (lane == 0) red[wave][4 + a] = s;
  }
  __syncthreads();
  if (threadIdx.x < 4 + AP) {
    const int k = threadIdx.x;
    float s = 0.f;
    if (k < 2 || k >= 4) s = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    partial[(size_t)blockIdx.x * (4 + AP) + k] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// grad = sum of split-K slabs (+ log_std gradient from the loss partials); loss4 from the partials.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) grad_reduce_kernel(size_t P, int ksplit, size_t slab_stride, const float* __restrict__ slabs, int ls_off, int A,
                                                          int AP, int nblk, const float* __restrict__ partial, const float* __restrict__ log_std,
                                                          float ent_coef, float vf_coef, float ent_weight, float* __restrict__ grad,
                                                          float* __restrict__ loss4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P) {
    float s = 0.f;
    if (i >= (size_t)ls_off && i < (size_t)ls_off + A) {
      const int a = (int)(i - ls_off);
      for (int k = 0; k < nblk; ++k) s += partial[(size_t)k * (4 + AP) + 4 + a];
      s -= ent_coef * ent_weight;
    } else {
      for (int k = 0; k < ksplit; ++k) s += slabs[(size_t)k * slab_stride + i];
    }
    grad[i] = s;
  }
  if (i == 0 && loss4) {
    float la = 0.f, lv = 0.f, sl = 0.f;
    for (int k = 0; k < nblk; ++k) { la += partial[(size_t)k * (4 + AP)]; lv += partial[(size_t)k * (4 + AP) + 1]; }
    for (int a = 0; a < A; ++a) sl += log_std[a];
    const float ent = (0.5f * (float)A * (1.f + kLog2Pi) + sl) * ent_weight;
    loss4[0] = la + vf_coef * lv - ent_coef * ent;
    loss4[1] = lv;
    loss4[2] = la;
    loss4[3] = ent;
  }
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
constexpr int kNormBlocks = 128;

__global__ void __launch_bounds__(256) sumsq_kernel(size_t P, const float* __restrict__ g, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (size_t)gridDim.x * blockDim.x) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) adam_kernel(size_t P, float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ g, const float* __restrict__ partial, const int* __restrict__ count_base,
                                                   int step_offset, mppo_adam_cfg_t c) {
  float ss = 0.f;
  for (int k = 0; k < kNormBlocks; ++k) ss += partial[k];  // same order in every workgroup: bitwise-identical scale
  const float norm = sqrtf(ss);
  const float scale = norm < c.max_grad_norm ? 1.f : c.max_grad_norm / norm;
  const int count = count_base[0] + step_offset;
  float lr = c.lr;
  if (c.anneal) lr = c.lr * (1.f - (float)(count / c.sched_div) / (float)c.num_updates);
  const float t = (float)(count + 1);
  const float bc1 = 1.f - powf(c.b1, t), bc2 = 1.f - powf(c.b2, t);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float gi = g[i] * scale;
  const float mi = c.b1 * m[i] + (1.f - c.b1) * gi;
  const float vi = c.b2 * v[i] + (1.f - c.b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] = p[i] - lr * (mi / bc1) / (sqrtf(vi / bc2) + c.eps);
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator
// ------------------------------------------------------------------------------------------------
struct U4 { unsigned x, y, z, w; };
__host__ __device__ inline U4 philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}

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

// actor + critic forward on n rows of `obs` (optionally gathered); outputs into the FwdBufs
int32_t mlp_forward(const mppo_net_t& net, const float* params, int n, const float* obs, int obs_ld, const int* gather, const FwdBufs& fb,
                    hipStream_t stream) {
  const ParamLayout L = param_layout(net.O, net.A, net.H);
  const int H = net.H, act_a = net.use_tanh ? ACT_TANH : ACT_RELU;
  GemmBatch gb{};
  gb.count = 2; gb.ksplit = 1;
  gb.p[0] = fwd_prob(obs, obs_ld, gather, n, net.O, params + L.a_w1, H, params + L.a_b1, act_a, fb.h1a, H);
  gb.p[1] = fwd_prob(obs, obs_ld, gather, n, net.O, params + L.c_w1, H, params + L.c_b1, ACT_RELU, fb.h1c, H);
  MPPO_TRY(gemm_launch(gb, 0, 0, EPI_BIAS_ACT, net.bf16, stream));
  gb.p[0] = fwd_prob(fb.h1a, H, nullptr, n, H, params + L.a_w2, H, params + L.a_b2, act_a, fb.h2a, H);
  gb.p[1] = fwd_prob(fb.h1c, H, nullptr, n, H, params + L.c_w2, H, params + L.c_b2, ACT_RELU, fb.h2c, H);
  MPPO_TRY(gemm_launch(gb, 0, 0, EPI_BIAS_ACT, net.bf16, stream));
  gb.p[0] = fwd_prob(fb.h2a, H, nullptr, n, H, params + L.a_w3, net.A, params + L.a_b3, ACT_NONE, fb.mean, fb.AP);
  gb.p[1] = fwd_prob(fb.h2c, H, nullptr, n, H, params + L.c_w3, 1, params + L.c_b3, ACT_NONE, fb.value, 1);
  MPPO_TRY(gemm_launch(gb, 0, 0, EPI_BIAS_ACT, 0, stream));  // heads stay f32
  return MPPO_OK;
}

int32_t policy_sample(const mppo_net_t& net, const float* params, int n, const FwdBufs& fb, const float* noise, float* action, float* log_prob,
                      hipStream_t stream) {
  const ParamLayout L = param_layout(net.O, net.A, net.H);
  hipLaunchKernelGGL(sample_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, n, net.A, fb.AP, fb.mean, params + L.log_std, noise, action, log_prob);
  MPPO_CHECK_LAUNCH("sample_kernel");
  return MPPO_OK;
}

int32_t minibatch_grad(const mppo_net_t& net, const float* params, const mppo_batch_t& batch, const int* idx, int mb, const float* adv_stat,
                       float inv_count, const mppo_loss_cfg_t& lc, float* grad, float* loss4, const GradBufs& gbuf, hipStream_t stream) {
  const ParamLayout L = param_layout(net.O, net.A, net.H);
  const int H = net.H, A = net.A, AP = gbuf.f.AP, O = net.O;
  const int act_a = net.use_tanh ? ACT_TANH : ACT_RELU;
  MPPO_TRY(mlp_forward(net, params, mb, batch.obs, batch.obs_ld, idx, gbuf.f, stream));
  const int nblk = cdiv(mb, 256);
  hipLaunchKernelGGL(loss_kernel, dim3(nblk), dim3(256), 0, stream, mb, A, AP, idx, gbuf.f.mean, gbuf.f.value, params + L.log_std, batch, adv_stat,
                     inv_count, lc, gbuf.dmean, gbuf.dv, gbuf.partial);
  MPPO_CHECK_LAUNCH("loss_kernel");
  GemmBatch gb{};
  gb.count = 2; gb.ksplit = 1;
  // dZ2 = (dOut . W3^T) * act'(h2)
  {
    GemmProb& a = gb.p[0]; a = GemmProb{};
    a.A = gbuf.dmean; a.lda = AP; a.M = mb; a.K = A; a.B = params + L.a_w3; a.ldb = A; a.N = H; a.aux = gbuf.f.h2a; a.ldaux = H; a.act = act_a; a.C = gbuf.dz2a; a.ldc = H;
    GemmProb& c = gb.p[1]; c = GemmProb{};
    c.A = gbuf.dv; c.lda = 1; c.M = mb; c.K = 1; c.B = params + L.c_w3; c.ldb = 1; c.N = H; c.aux = gbuf.f.h2c; c.ldaux = H; c.act = ACT_RELU; c.C = gbuf.dz2c; c.ldc = H;
  }
  MPPO_TRY(gemm_launch(gb, 0, 1, EPI_DACT, 0, stream));
  // dZ1 = (dZ2 . W2^T) * act'(h1)
  {
    GemmProb& a = gb.p[0]; a = GemmProb{};
    a.A = gbuf.dz2a; a.lda = H; a.M = mb; a.K = H; a.B = params + L.a_w2; a.ldb = H; a.N = H; a.aux = gbuf.f.h1a; a.ldaux = H; a.act = act_a; a.C = gbuf.dz1a; a.ldc = H;
    GemmProb& c = gb.p[1]; c = GemmProb{};
    c.A = gbuf.dz2c; c.lda = H; c.M = mb; c.K = H; c.B = params + L.c_w2; c.ldb = H; c.N = H; c.aux = gbuf.f.h1c; c.ldaux = H; c.act = ACT_RELU; c.C = gbuf.dz1c; c.ldc = H;
  }
  MPPO_TRY(gemm_launch(gb, 0, 1, EPI_DACT, net.bf16, stream));
  // weight gradients dW = H_prev^T . dZ into split-K slabs; the bias gradients (column sums of dZ) ride along
  gb.count = 6; gb.ksplit = gbuf.ksplit; gb.slab_stride = gbuf.slab_stride;
  auto wprob = [&](const float* Aprev, int lda, const int* gather, int Min, const float* dZ, int ldz, int N, int off_w, int off_b) {
    GemmProb p{};
    p.A = Aprev; p.lda = lda; p.gather = gather; p.M = Min; p.K = mb; p.B = dZ; p.ldb = ldz; p.N = N; p.C = gbuf.slabs + off_w; p.ldc = N;
    p.bias_out = gbuf.slabs + off_b;
    return p;
  };
  gb.p[0] = wprob(gbuf.f.h2a, H, nullptr, H, gbuf.dmean, AP, A, L.a_w3, L.a_b3);
  gb.p[1] = wprob(gbuf.f.h1a, H, nullptr, H, gbuf.dz2a, H, H, L.a_w2, L.a_b2);
  gb.p[2] = wprob(batch.obs, batch.obs_ld, idx, O, gbuf.dz1a, H, H, L.a_w1, L.a_b1);
  gb.p[3] = wprob(gbuf.f.h2c, H, nullptr, H, gbuf.dv, 1, 1, L.c_w3, L.c_b3);
  gb.p[4] = wprob(gbuf.f.h1c, H, nullptr, H, gbuf.dz2c, H, H, L.c_w2, L.c_b2);
  gb.p[5] = wprob(batch.obs, batch.obs_ld, idx, O, gbuf.dz1c, H, H, L.c_w1, L.c_b1);
  MPPO_TRY(gemm_launch(gb, 1, 0, EPI_STORE, net.bf16, stream));
  const float ent_weight = (float)mb * inv_count;
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(cdiv((long)L.total, 256)), dim3(256), 0, stream, (size_t)L.total, gbuf.ksplit, gbuf.slab_stride, gbuf.slabs,
                     L.log_std, A, AP, nblk, gbuf.partial, params + L.log_std, lc.ent_coef, lc.vf_coef, ent_weight, grad, loss4);
  MPPO_CHECK_LAUNCH("grad_reduce_kernel");
  return MPPO_OK;
}

int32_t clip_adam(size_t P, float* params, float* m, float* v, const float* grad, const int* count_base, int step_offset, const mppo_adam_cfg_t& cfg,
                  float* ws, hipStream_t stream) {
  hipLaunchKernelGGL(sumsq_kernel, dim3(kNormBlocks), dim3(256), 0, stream, P, grad, ws);
  MPPO_CHECK_LAUNCH("sumsq_kernel");
  hipLaunchKernelGGL(adam_kernel, dim3(cdiv((long)P, 256)), dim3(256), 0, stream, P, params, m, v, grad, ws, count_base, step_offset, cfg);
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
  MPPO_REQUIRE(net->O >= 1 && net->A >= 1 && net->A <= 32 && net->H >= 4 && (net->H % 4) == 0, "unsupported network geometry O=%d A=%d H=%d (need A<=32, H%%4==0)",
               net->O, net->A, net->H);
  MPPO_REQUIRE(net->OP >= net->O && (net->OP % 4) == 0, "OP=%d must be a multiple of 4 and >= O=%d", net->OP, net->O);
  return MPPO_OK;
}

extern "C" size_t mppo_param_count(const mppo_net_t* net) { return net ? (size_t)param_layout(net->O, net->A, net->H).total : 0; }

extern "C" size_t mppo_policy_ws_bytes(const mppo_net_t* net, int32_t n) { return net ? fwd_bufs_floats(*net, n) * sizeof(float) : 0; }

extern "C" int32_t mppo_policy_forward(const mppo_net_t* net, const float* params, int32_t n, const float* obs, int32_t obs_ld, const float* noise,
                                       float* action, float* log_prob, float* value, float* mean_out, void* ws, size_t ws_bytes, void* stream) {
  MPPO_TRY(check_net(net));
  MPPO_REQUIRE(params && obs && value && ws && n >= 1, "mppo_policy_forward: null argument or n < 1");
  MPPO_REQUIRE(obs_ld >= net->O, "mppo_policy_forward: obs_ld %d < O %d", obs_ld, net->O);
  MPPO_REQUIRE(!noise || (action && log_prob), "mppo_policy_forward: noise given without action/log_prob outputs");
  if (ws_bytes < mppo_policy_ws_bytes(net, n)) return fail(MPPO_ENOMEM, "mppo_policy_forward: workspace %zu < %zu bytes", ws_bytes, mppo_policy_ws_bytes(net, n));
  hipStream_t s = static_cast<hipStream_t>(stream);
  FwdBufs fb = carve_fwd(*net, n, static_cast<float*>(ws));
  fb.value = value;
  MPPO_TRY(mlp_forward(*net, params, n, obs, obs_ld, nullptr, fb, s));
  if (noise) MPPO_TRY(policy_sample(*net, params, n, fb, noise, action, log_prob, s));
  if (mean_out) MPPO_CHECK_HIP(hipMemcpyAsync(mean_out, fb.mean, (size_t)n * fb.AP * sizeof(float), hipMemcpyDeviceToDevice, s));
  return MPPO_OK;
}

extern "C" int32_t mppo_gae(int32_t T, int32_t N, float gamma, float lam, const float* reward, const float* value, const uint8_t* done,
                            const float* last_val, float* adv, float* target, void* stream) {
  MPPO_REQUIRE(T >= 1 && N >= 1 && reward && value && done && last_val && adv && target, "mppo_gae: bad argument");
  return gae_launch(T, N, gamma, lam, reward, value, done, last_val, adv, target, static_cast<hipStream_t>(stream));
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
  return minibatch_grad(*net, params, *batch, idx, mb, adv_stat, inv_count, *lc, grad, loss4, gb, static_cast<hipStream_t>(stream));
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

extern "C" size_t mppo_adam_ws_bytes(size_t) { return kNormBlocks * sizeof(float); }

extern "C" int32_t mppo_clip_adam(size_t P, float* params, float* m, float* v, const float* grad, const int32_t* count_base, int32_t step_offset,
                                  const mppo_adam_cfg_t* cfg, void* ws, size_t ws_bytes, void* stream) {
  MPPO_REQUIRE(P >= 1 && params && m && v && grad && count_base && cfg && ws, "mppo_clip_adam: null argument");
  MPPO_REQUIRE(!cfg->anneal || (cfg->sched_div >= 1 && cfg->num_updates >= 1), "mppo_clip_adam: anneal needs sched_div, num_updates >= 1");
  if (ws_bytes < mppo_adam_ws_bytes(P)) return fail(MPPO_ENOMEM, "mppo_clip_adam: workspace too small");
  return clip_adam(P, params, m, v, grad, count_base, step_offset, *cfg, static_cast<float*>(ws), static_cast<hipStream_t>(stream));
}

extern "C" int32_t mppo_normal_fill(uint64_t seed, uint64_t stream_id, size_t n, float* out, void* stream) {
  MPPO_REQUIRE(out && n >= 1, "mppo_normal_fill: bad argument");
  return normal_fill_ctr(seed, stream_id, nullptr, n, out, static_cast<hipStream_t>(stream));
}
