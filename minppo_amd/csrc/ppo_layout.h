// ppo_layout.h — flat parameter layout and workspace carving shared by the PPO stages and the engine.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>

#include "../../include/minppo_hip.h"
#include "mppo_common.h"

namespace mppo {

// Offsets (floats) inside the flat parameter / gradient / Adam-moment vectors (include/minppo_hip.h).  Every tensor starts on
// a 16-byte boundary whatever the action dimension is (float4 rows in the fused row pass and the weight-gradient kernel, bf16
// for odd A); the up to three padding words after a tensor hold zeros in all four vectors and stay zero (zero gradient -> zero
// Adam update).  `total` counts the padding; the model has total - npad_words parameters (P of SURVEY 8).
// `nl` hidden layers per MLP (`model.num_layers`, reference train.py:79,82): tensors in the order actor W_0, b_0, ..., W_nl, b_nl
// (W_nl / b_nl: the output layer), log_std, critic W_0 ... b_nl.  The named fields are the two-hidden-layer view the fused kernels
// use: w1 / b1 = layer 0, w2 / b2 = layer 1 (meaningful for nl >= 2), w3 / b3 = the OUTPUT layer whatever nl is.
constexpr int kMaxHidden = 4;
constexpr int kMaxPads = 4 * (kMaxHidden + 1) + 1;
inline int net_layers(const mppo_net_t& n) { return n.num_layers > 0 ? n.num_layers : 2; }
struct ParamLayout {
  int a_w1, a_b1, a_w2, a_b2, a_w3, a_b3, log_std, c_w1, c_b1, c_w2, c_b2, c_w3, c_b3, total;
  int nl, a_w[kMaxHidden + 1], a_b[kMaxHidden + 1], c_w[kMaxHidden + 1], c_b[kMaxHidden + 1];
  int pad_off[kMaxPads], pad_cnt[kMaxPads], npad, npad_words;  // the padding runs (offset, length <= 3)
};
inline ParamLayout param_layout(int O, int A, int H, int nl = 2) {
  ParamLayout L{};
  int o = 0;
  auto take = [&](int n) {
    const int at = o;
    o += n;
    const int pad = (4 - (o & 3)) & 3;
    if (pad) { L.pad_off[L.npad] = o; L.pad_cnt[L.npad] = pad; ++L.npad; L.npad_words += pad; o += pad; }
    return at;
  };
  L.nl = nl;
  for (int l = 0; l <= nl; ++l) { L.a_w[l] = take((l == 0 ? O : H) * (l == nl ? A : H)); L.a_b[l] = take(l == nl ? A : H); }
  L.log_std = take(A);
  for (int l = 0; l <= nl; ++l) { L.c_w[l] = take((l == 0 ? O : H) * (l == nl ? 1 : H)); L.c_b[l] = take(l == nl ? 1 : H); }
  L.total = o;
  L.a_w1 = L.a_w[0]; L.a_b1 = L.a_b[0]; L.a_w2 = L.a_w[nl >= 2 ? 1 : 0]; L.a_b2 = L.a_b[nl >= 2 ? 1 : 0]; L.a_w3 = L.a_w[nl]; L.a_b3 = L.a_b[nl];
  L.c_w1 = L.c_w[0]; L.c_b1 = L.c_b[0]; L.c_w2 = L.c_w[nl >= 2 ? 1 : 0]; L.c_b2 = L.c_b[nl >= 2 ? 1 : 0]; L.c_w3 = L.c_w[nl]; L.c_b3 = L.c_b[nl];
  return L;
}
inline ParamLayout param_layout(const mppo_net_t& n) { return param_layout(n.O, n.A, n.H, net_layers(n)); }

__host__ __device__ inline size_t pad4(size_t n) { return (n + 3) & ~(size_t)3; }

// K-QUAD layout of a [rows][cols] operand of the weight-gradient product (contraction over rows): [rows/4][cols][4], i.e. the
// four consecutive rows of a column are one float4.  The fused row pass (k_fused.hip) writes h1, h2, dZ1, dZ2, dOut and the
// gathered observations this way and k_wgrad.hip reads them; rows are padded to the row pass's 16-row tiles with zeros.
__host__ __device__ inline size_t quad_index(size_t row, size_t col, size_t cols) { return ((row >> 2) * cols + col) * 4 + (row & 3); }
__host__ __device__ inline size_t pad16(size_t n) { return (n + 15) & ~(size_t)15; }

// activations of one forward pass over n rows
struct FwdBufs {
  float *h1a, *h2a, *h1c, *h2c, *mean, *value;  // h1 = hidden layer 0, h2 = the LAST hidden layer (two-hidden-layer view: ha[0], ha[1])
  float *ha[kMaxHidden], *hc[kMaxHidden];        // hidden activations of actor / critic, layer by layer
  int AP;
  // optional (engine rollout of a bf16 network): the bf16 fragment-order weight copies of a gradient workspace whose shadow copies
  // are current (GradBufs::frag); nullptr = convert the float weights in the kernel
  const unsigned short* frag = nullptr;
  size_t frag_net_stride = 0;
};
inline size_t fwd_bufs_floats(const mppo_net_t& net, int n) {
  const size_t AP = pad4((size_t)net.A);
  return 2 * (size_t)net_layers(net) * pad4((size_t)n * net.H) + pad4((size_t)n * AP) + pad4((size_t)n);
}
inline FwdBufs carve_fwd(const mppo_net_t& net, int n, float* ws) {
  FwdBufs f;
  f.AP = (int)pad4((size_t)net.A);
  const size_t nh = pad4((size_t)n * net.H);
  const int nl = net_layers(net);
  for (int l = 0; l < kMaxHidden; ++l) f.ha[l] = f.hc[l] = nullptr;
  for (int l = 0; l < nl; ++l) { f.ha[l] = ws; ws += nh; }
  for (int l = 0; l < nl; ++l) { f.hc[l] = ws; ws += nh; }
  f.h1a = f.ha[0]; f.h2a = f.ha[nl - 1]; f.h1c = f.hc[0]; f.h2c = f.hc[nl - 1];
  f.mean = ws; ws += pad4((size_t)n * f.AP);
  f.value = ws;
  return f;
}

constexpr int kGradKSplitMax = 8;  // slabs reserved in the workspace
inline int grad_ksplit() {  // K-slices of the weight-gradient product; MPPO_KSPLIT overrides for measurements
  static const int v = [] { const char* e = MPPO_EXPERIMENT_ENV("MPPO_KSPLIT"); int k = e ? atoi(e) : 8; return k < 1 ? 1 : (k > kGradKSplitMax ? kGradKSplitMax : k); }();
  return v;
}

// everything one minibatch gradient needs beyond the forward activations
struct GradBufs {
  FwdBufs f;
  float *dout, *dz2a, *dz2c, *dz1a, *dz1c, *partial, *slabs;  // dout [mb, AP+4]: d mean | d value; dz1 = layer 0, dz2 = the LAST hidden layer
  float *dza[kMaxHidden], *dzc[kMaxHidden];                    // d loss / d pre-activation of every hidden layer
  float* xmb;  // [mb, OP]: the minibatch observations, laid out contiguously by the first forward GEMM
  // Second observation buffer (k-quad layout like xmb in the fused path): the engine's row pass of optimizer step s gathers the
  // rows of step s + 1 on workgroups of its own launch (k_fused.hip, XPre) - they depend on the permutation only, not on the
  // parameters - so that step s + 1 starts from a contiguous tile instead of the index -> row chain; xmb and xmb2 alternate.
  float* xmb2;
  // Shadow copy of the second-layer weights, transposed: w2t[net][n][k] = W2_net[k][n] (net 0 actor, 1 critic).  The backward
  // product dZ1 = dZ2 . W2^T then streams its B operand exactly like a forward layer (rows of consecutive outputs) instead of
  // gathering 16-byte pieces of 32 different rows per load.  Written by shadow_refresh() and kept current by clip_adam(); the
  // CALLER knows whether it matches `params` (w2t_valid): the workspace itself carries no state.
  float* w2t;
  bool w2t_valid;
  // bf16 networks only: the hidden-layer weights once more, rounded to bf16 and stored in the order the MFMA pipe consumes them
  // (fragment order: [k stage of 32][32-column wave slab][column tile][lane][8 values], see frag_index): per network W1 (K padded to 32 with
  // zeros), W2 and W2^T.  A lane's operands of a stage are 32 contiguous bytes, a wave's 2 KB: half the bytes of the float
  // weights, no conversion in the GEMM loop.  Same validity as w2t.
  unsigned short* frag;     // nullptr unless net.bf16
  size_t frag_net_stride;   // in bf16 elements: KP*H + 2*H*H
  int ksplit;
  size_t slab_stride;
};
// one pre-gathered minibatch (xmb / xmb2): the observation rows in k-quad layout [mbp / 4][OP][4], and behind them the per-row scalars of the
// loss in the same layout, [mbp / 4][A + 4][4]: columns 0 .. A - 1 the action, A old log_prob, A + 1 advantage, A + 2 old value, A + 3 target
// (k_fused.hip: gathered by the same workgroups as the rows, read by the next step's row tiles as whole float4s - no index -> row chain)
__host__ __device__ inline size_t xquad_obs_floats(int OP, int mb) { return pad4(pad16((size_t)mb) * OP); }
inline size_t xquad_floats(const mppo_net_t& net, int mb) { return xquad_obs_floats(net.OP, mb) + pad4(pad16((size_t)mb) * (net.A + 4)); }
inline size_t grad_bufs_floats(const mppo_net_t& net, int mb) {
  const size_t mbp = pad16((size_t)mb);  // the quad-layout operands are written in whole 16-row tiles
  const size_t AP = pad4((size_t)net.A), nh = pad4(mbp * net.H);
  const size_t P = pad4((size_t)param_layout(net).total);
  const size_t nblk = (size_t)(mb + 3) / 4;  // head kernel: 8 rows per workgroup, 4 with more than 31 action dimensions (fused kernel: 16)
  return fwd_bufs_floats(net, (int)mbp) + pad4(mbp * (AP + 4)) + 2 * (size_t)net_layers(net) * nh + pad4(nblk * (4 + AP)) + (size_t)kGradKSplitMax * P + 2 * xquad_floats(net, mb) + 2 * pad4((size_t)net.H * net.H) +  // (xmb, xmb2)
         (net.bf16 ? pad4((size_t)(((net.O + 31) & ~31) + 2 * net.H) * net.H) : 0);  // bf16 fragments: 2 networks x (KP + 2H) x H halves = that many floats
}
inline GradBufs carve_grad(const mppo_net_t& net, int mb, float* ws) {
  GradBufs g;
  const size_t mbp = pad16((size_t)mb);
  g.f = carve_fwd(net, (int)mbp, ws);
  ws += fwd_bufs_floats(net, (int)mbp);
  const size_t AP = (size_t)g.f.AP, nh = pad4(mbp * net.H);
  g.dout = ws; ws += pad4(mbp * (AP + 4));
  const int nl = net_layers(net);
  for (int l = 0; l < kMaxHidden; ++l) g.dza[l] = g.dzc[l] = nullptr;
  for (int l = nl - 1; l >= 0; --l) { g.dza[l] = ws; ws += nh; g.dzc[l] = ws; ws += nh; }
  g.dz2a = g.dza[nl - 1]; g.dz2c = g.dzc[nl - 1]; g.dz1a = g.dza[0]; g.dz1c = g.dzc[0];
  const size_t nblk = (size_t)(mb + 3) / 4;
  g.partial = ws; ws += pad4(nblk * (4 + AP));
  g.slabs = ws;
  g.xmb = ws + (size_t)kGradKSplitMax * pad4((size_t)param_layout(net).total);
  g.xmb2 = g.xmb + xquad_floats(net, mb);
  g.w2t = g.xmb2 + xquad_floats(net, mb);  // (the shadow copies stay the LAST region of the workspace)
  g.w2t_valid = false;
  g.frag = net.bf16 ? reinterpret_cast<unsigned short*>(g.w2t + 2 * pad4((size_t)net.H * net.H)) : nullptr;
  g.frag_net_stride = (size_t)(((net.O + 31) & ~31) + 2 * net.H) * net.H;
  g.ksplit = grad_ksplit();
  g.slab_stride = pad4((size_t)param_layout(net).total);
  return g;
}

// Pre-gathered observations of the engine's minibatch loop (see GradBufs::xmb2).  cur: the k-quad buffer holding THIS step's rows
// (written by the previous step's launch or by fused_gather_rows); next / idx_next: where the launch gathers the next step's rows
// (idx_next == nullptr: the last step, nothing to gather).
struct XPre { const float* cur; float* next; const int* idx_next; };

// stage launchers (k_ppo.hip)
int32_t policy_forward(const mppo_net_t& net, const float* params, int n, const float* obs, int obs_ld, const FwdBufs& fb, const float* noise, float* action,
                       float* log_prob, float* value, float* mean_out, hipStream_t stream);
// sq_partial (optional): per-workgroup sums of squares of the reduced gradient, consumed by clip_adam(have_sumsq = true)
struct PeerStep;  // peer.h: the ranks' peer-to-peer gradient exchange
// peer (optional, fused path only): `grad` is this rank's exchange buffer; the weight-gradient launch writes it with system-scope stores and signals the peers
int32_t minibatch_grad(const mppo_net_t& net, const float* params, const mppo_batch_t& batch, const int* idx, int mb, const float* adv_stat, float inv_count,
                       const mppo_loss_cfg_t& lc, float* grad, float* loss4, float* sq_partial, const GradBufs& gbuf, hipStream_t stream, const XPre* pre = nullptr,
                       const PeerStep* peer = nullptr);
// shadow (optional): the W2^T copies to keep in step with the parameters (GradBufs::w2t of the workspace the row pass reads)
struct ShadowRef {
  float* w2t; int a_w2, c_w2, H;
  unsigned short* frag; size_t frag_net_stride; int a_w1, c_w1, O;  // bf16 fragments (nullptr for a float network)
  // filled in by clip_adam: the flat workgroups skip the ranges that tile workgroups own (sorted, in float4 units)
  int flat_blocks, nskip; unsigned skip_start4[4], skip_len4[4];
};
// Position (in bf16 elements) of B(k, n) of a [K][N] weight inside its fragment-order copy: stage S = k / 32, wave slab w = n / 32 - a
// block of 1024 elements (2 KB) per (S, w) - and inside the block TILE-MAJOR: column tile tau (n = 32 w + 2 j + tau) is 512 contiguous
// elements, lane = 16 kq + j holds 8 of them, element 4 g + c with k = 32 S + 16 g + 4 kq + c.  One 16-byte load per lane and tile
// therefore reads 1 KB of CONTIGUOUS memory across the wave (eight whole 128-byte lines).  MPPO_FRAG_LANE_MAJOR (A/B builds): round 3's
// order, a lane's sixteen values of both tiles side by side - each of the two loads then touches half of sixteen lines.
#ifdef MPPO_FRAG_LANE_MAJOR
constexpr int kFragLaneElems = 16, kFragTileElems = 8;  // element strides between lanes / between the two tiles of a lane
#else
constexpr int kFragLaneElems = 8, kFragTileElems = 512;
#endif
__host__ __device__ inline size_t frag_index(int k, int n, int N) {
  const int S = k >> 5, kk = k & 31, g = kk >> 4, kq = (kk >> 2) & 3, c = kk & 3;
  const int w = n >> 5, nn = n & 31, j = nn >> 1, tau = nn & 1;
  return ((size_t)S * (N >> 5) + w) * 1024 + (size_t)(16 * kq + j) * kFragLaneElems + (size_t)tau * kFragTileElems + 4 * g + c;
}
// peer (optional): `grad` / `ws` are the reduced gradient and its sums of squares in the exchange buffer; the launch reduces this rank's slice first (peer.h)
int32_t clip_adam(size_t P, float* params, float* m, float* v, const float* grad, const int* count_base, int step_offset, const mppo_adam_cfg_t& cfg, float* ws,
                  bool have_sumsq, hipStream_t stream, const ShadowRef* shadow = nullptr, const PeerStep* peer = nullptr);
int32_t shadow_refresh(const mppo_net_t& net, const float* params, const GradBufs& gbuf, hipStream_t stream);
inline ShadowRef make_shadow_ref(const mppo_net_t& net, const GradBufs& g) {
  const ParamLayout L = param_layout(net);
  ShadowRef r{};
  r.w2t = g.w2t; r.a_w2 = L.a_w2; r.c_w2 = L.c_w2; r.H = net.H;
  r.frag = g.frag; r.frag_net_stride = g.frag_net_stride; r.a_w1 = L.a_w1; r.c_w1 = L.c_w1; r.O = net.O;
  return r;
}
int32_t gae_launch(int T, int N, float gamma, float lam, const float* reward, const float* value, const unsigned char* done, const float* last_val, float* adv,
                   float* target, hipStream_t stream);
// k_fused.hip: row-local forward + backward of one minibatch in a single launch (falls back to the layer-wise path when unsupported)
bool fused_supported(const mppo_net_t& net, const mppo_batch_t& b);
int fused_rows_per_workgroup(const mppo_net_t& net, int mb, bool pre);  // 16, or 32 where that turns two rounds of workgroups into one (k_fused.hip)
bool fused_rollout_supported(const mppo_net_t& net, const float* obs, int obs_ld);
int32_t fused_policy_forward(const mppo_net_t& net, const float* params, int n, const float* obs, int obs_ld, const float* noise, float* action, float* log_prob,
                             float* value, float* mean_out, int AP, hipStream_t stream, const unsigned short* frag = nullptr, size_t frag_net_stride = 0);
int32_t fused_forward_backward(const mppo_net_t& net, const float* params, const mppo_batch_t& batch, const int* idx, int mb, const float* adv_stat,
                               float inv_count, const mppo_loss_cfg_t& lc, const GradBufs& g, hipStream_t stream, const XPre* pre = nullptr);
// rows idx[0 .. mb) of the observations -> k-quad buffer `dst` (whole 16-row tiles, zeros past mb): the first step of an update
int32_t fused_gather_rows(const mppo_net_t& net, const mppo_batch_t& batch, const int* idx, int mb, float* dst, hipStream_t stream);
int32_t perm_fill_keys(unsigned long long seed, unsigned long long stream_id, const int* ctr, int B, unsigned* keys, int* vals, hipStream_t stream);
int32_t normal_fill_ctr(unsigned long long seed, unsigned long long stream_id, const int* ctr, size_t n, float* out, hipStream_t stream);
int32_t permutation_ctr(unsigned long long seed, unsigned long long stream_id, const int* ctr, int B, int* idx, void* ws, size_t ws_bytes, hipStream_t stream);
// E permutations (streams stream_id0 .. stream_id0 + E - 1) into idx[E][B] with one sort; same results as E calls of permutation_ctr
int32_t perm_fill_keys_batch(unsigned long long seed, unsigned long long stream_id0, const int* ctr, int B, int E, unsigned long long* keys, int* vals, hipStream_t stream);
size_t permutation_batch_ws_bytes(int B, int E);
int permutation_max_samples();  // the largest batch (samples per rank) the engine's two-launch sort takes
int32_t permutation_batch_prepare(int B, int E, void* ws, size_t ws_bytes, hipStream_t stream);  // zeroes the two-launch form's counters
void permutation_batch_counters(int B, int E, void* ws, size_t ws_bytes, int** ptr, int* n);  // the words the caller zeroes after every use (nullptr / 0: none)
int32_t permutation_batch_ctr(unsigned long long seed, unsigned long long stream_id0, const int* ctr, int B, int E, int* idx, void* ws, size_t ws_bytes, hipStream_t stream);
// k_rng.hip / k_perm.hip: jax.random-compatible streams (threefry2x32)
int32_t threefry_normal(const unsigned* key2, size_t n, float* out, hipStream_t s);
int32_t threefry_bits(const unsigned* key2, size_t n, unsigned* out, int* iota, hipStream_t s);
int32_t threefry_chain(unsigned* rng2, int T, int E, int rounds, unsigned* act_keys, unsigned* sort_keys, hipStream_t s);
// jax.random.permutation(key, B): `rounds` stable sorts by fresh random bits, sort_keys = [rounds][2] device words
int32_t threefry_permutation(const unsigned* sort_keys, int rounds, int B, int* idx, void* ws, size_t ws_bytes, hipStream_t stream);
inline int threefry_rounds(int B) { return B <= 1 ? 1 : (int)ceil(3.0 * log((double)B) / log(4294967295.0)); }

}  // namespace mppo
