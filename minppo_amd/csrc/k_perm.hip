// k_perm.hip — random permutations of [0, B) as jax.random.permutation makes them (reference minppo/train.py:258): the stable order of
// random 32-bit keys.  Every path sorts with the engine's own two launches (round 6: no library sort is left on the hot path):
//
//   A permutation is the order of the 64-bit values (key << 32 | position), which is what a STABLE sort of (key, position) pairs yields.
//   Launch 1 (perm_scatter_kernel) takes the keys - drawn from Philox in place, or read from an array (JAX's threefry bits) - and scatters
//   the values into NB buckets by the key's top bits: counts per workgroup in LDS first (an LDS atomic hands every value its rank inside the
//   workgroup's share of the bucket), then ONE device atomic per bucket and workgroup reserves the share.  Launch 2 (perm_bucket_kernel), one
//   workgroup per bucket, ranks its few hundred values in LDS (counting rank up to 256 values, bitonic above), finds its place by summing the
//   counters of the buckets before it and writes the payloads: the position itself, or - for the rounds of JAX's multi-round shuffle -
//   the previous round's order at that position.
//
//   NB = 256 up to 131 072 samples, doubling with the batch up to 4 096 (mean load <= 512 of a bucket's 1 024 slots, 22 sigma of headroom):
//   batches up to 2 097 152 samples per rank (209 715 environments x 10 steps).  A bucket that overflows all the same makes the launch
//   mark the word behind the counters; the engine makes the mark sticky (count[3]) and Trainer.check_status raises.
//
//   The counters must be zero before launch 1.  The engine's end-of-update kernel takes them back to zero (so they are zero between
//   updates whatever happens to the update index); the stand-alone entry points and the threefry rounds put a zeroing launch in front.
#include <hip/hip_runtime.h>

#include <cstring>

#include "mppo_common.h"
#include "philox.h"
#include "ppo_layout.h"
#include <wave_ops.h>

namespace mppo {
constexpr int kPermCap = 1024;         // slots per bucket
constexpr int kPermMinBuckets = 256, kPermMaxBuckets = 4096;
constexpr int kPermMeanLoad = 512;     // NB is the smallest power of two with B / NB <= this (standard deviation <= 23)
constexpr int kPermMaxB = kPermMaxBuckets * kPermMeanLoad;

static int perm_buckets(int B) {
  int nb = kPermMinBuckets;
  while (nb < kPermMaxBuckets && (long long)nb * kPermMeanLoad < B) nb <<= 1;
  return nb;
}
static int ilog2(int x) { int b = 0; while ((1 << b) < x) ++b; return b; }
static size_t perm_counter_bytes(int NB, int E) { return align_up((size_t)(E * NB + 64) * 4, 256); }  // E x NB counters + the overflow word
static size_t perm_slot_bytes(int NB, int E) { return (size_t)E * NB * kPermCap * 8; }
int permutation_max_samples() { return kPermMaxB; }
static int32_t perm_check_size(int B) {
  if (B > kPermMaxB) return fail(MPPO_EINVAL, "permutation of %d samples: the engine's sort takes at most %d per rank (4096 buckets of 512)", B, kPermMaxB);
  return MPPO_OK;
}

constexpr int kScatterThreads = 1024;  // x 4 keys: 4096 values per workgroup - one global atomic per bucket reserves the workgroup's share
// PHILOX: key i of job e = word (i & 3) of philox(i >> 2, ctr, stream_id0 + e, seed) - the engine's stream of permutation keys;
// otherwise keys[e][i] (an array of B keys per job)
template <bool PHILOX>
__global__ void __launch_bounds__(kScatterThreads) perm_scatter_kernel(unsigned long long seed, unsigned long long stream_id0, const int* __restrict__ ctr, const unsigned* __restrict__ keys,
                                                                       int B, int NB, int shift, int* __restrict__ cnt_base, unsigned long long* __restrict__ slots) {
  MPPO_DYN_SMEM(smem_raw);
  int* s_mem = reinterpret_cast<int*>(smem_raw);
  int* s_cnt = s_mem;
  int* s_base = s_mem + NB;
  const int t = threadIdx.x, q = blockIdx.x * kScatterThreads + t, e = blockIdx.y, E = gridDim.y;
  int* cnt = cnt_base + e * NB;
  for (int k = t; k < NB; k += kScatterThreads) s_cnt[k] = 0;
  __syncthreads();
  unsigned z[4] = {0u, 0u, 0u, 0u};
  int rank[4] = {0, 0, 0, 0};
  if (4 * q < B) {
    if (PHILOX) {
      const unsigned long long stream_id = stream_id0 + (unsigned long long)e;
      const U4 r = philox4x32((unsigned)q, ctr ? (unsigned)ctr[0] : 0u, (unsigned)stream_id, (unsigned)(stream_id >> 32) ^ 0x5045524Du, (unsigned)seed, (unsigned)(seed >> 32));
      z[0] = r.x; z[1] = r.y; z[2] = r.z; z[3] = r.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (4 * q + k < B) z[k] = keys[(size_t)e * B + 4 * q + k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (4 * q + k < B) rank[k] = atomicAdd(&s_cnt[z[k] >> shift], 1);
  }
  __syncthreads();
  for (int k = t; k < NB; k += kScatterThreads) s_base[k] = s_cnt[k] ? atomicAdd(&cnt[k], s_cnt[k]) : 0;
  __syncthreads();
  if (4 * q < B) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = 4 * q + k;
      if (i < B) {
        const int b = (int)(z[k] >> shift), pos = s_base[b] + rank[k];
        if (pos < kPermCap) slots[((size_t)e * NB + b) * kPermCap + pos] = ((unsigned long long)z[k] << 32) | (unsigned long long)(unsigned)i;
        else cnt_base[E * NB] = 1;  // (22 sigma away; recorded all the same: the engine makes the mark sticky)
      }
    }
  }
}

// one workgroup per (bucket, job): out[job][offset + rank] = payload of the value, the payload being its position or vals_in[job][position]
__global__ void __launch_bounds__(256) perm_bucket_kernel(int B, int NB, const int* __restrict__ cnt_base, const unsigned long long* __restrict__ slots,
                                                          const int* __restrict__ vals_in, int* __restrict__ out) {
  __shared__ unsigned long long s[kPermCap];
  __shared__ int s_part[4];
  const int b = blockIdx.x, e = blockIdx.y, t = threadIdx.x;
  const int* cnt = cnt_base + e * NB;
  // this bucket's count (a uniform read) and its offset: the counts of the buckets before it (thread t adds buckets t, t + 256, ..)
  const int n_all = cnt[b];
  int before = 0;
  for (int k = t; k < b; k += 256) before += cnt[k];
  for (int m = 1; m < 64; m <<= 1) before += __shfl_xor(before, m);
  if ((t & 63) == 0) s_part[t >> 6] = before;
  const int n = n_all < kPermCap ? n_all : kPermCap;
  const unsigned long long* src = slots + ((size_t)e * NB + b) * kPermCap;
  const int* vin = vals_in ? vals_in + (size_t)e * B : nullptr;
  int* dst = out + (size_t)e * B;
  if (n <= 256) {
    // the usual case below 256 values per bucket: every thread holds one value and counts the smaller ones - its place in the bucket.  The
    // values are distinct (the position is part of them): the ranks are a permutation.
    const unsigned long long v = t < n ? src[t] : ~0ull;
    s[t] = v;
    __syncthreads();
    const int off = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    int rk = 0;
#pragma unroll 8
    for (int j = 0; j < n; ++j) rk += (int)(s[j] < v);
    if (t < n) { const int i = (int)(unsigned)v; dst[off + rk] = vin ? vin[i < B ? i : 0] : i; }  // (i < B always; the clamp keeps a corrupted slot from becoming a wild read)
  } else {
    int m2 = 2;
    while (m2 < n) m2 <<= 1;  // elements sorted: the next power of two (uniform)
    for (int k = t; k < m2; k += 256) s[k] = k < n ? src[k] : ~0ull;
    __syncthreads();
    const int off = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    for (int k = 2; k <= m2; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = t; i < m2; i += 256) {
          const int p = i ^ j;
          if (p > i) {
            const unsigned long long a0 = s[i], a1 = s[p];
            const bool up = (i & k) == 0;
            if ((a0 > a1) == up) { s[i] = a1; s[p] = a0; }
          }
        }
        __syncthreads();
      }
    for (int k = t; k < n; k += 256) { const int i = (int)(unsigned)s[k]; dst[off + k] = vin ? vin[i < B ? i : 0] : i; }
  }
}

// the counters of a stand-alone sort are taken to zero by a launch of their own in front of it (a kernel, not a memset node: the threefry rounds are
// captured into the engine's hipGraph like everything else)
__global__ void perm_zero_kernel(int* __restrict__ p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}
static int32_t perm_zero(int* cnt, int n, hipStream_t s) {
  hipLaunchKernelGGL(perm_zero_kernel, dim3(cdiv(n, 1024)), dim3(256), 0, s, cnt, n);
  MPPO_CHECK_LAUNCH("perm_zero_kernel");
  return MPPO_OK;
}

// the two launches for E jobs of B keys each (counters zero on entry)
static int32_t perm_sort_launch(bool philox, unsigned long long seed, unsigned long long stream_id0, const int* ctr, const unsigned* keys, int B, int E, int* cnt,
                                unsigned long long* slots, const int* vals_in, int* out, hipStream_t s) {
  const int NB = perm_buckets(B), shift = 32 - ilog2(NB);
  const dim3 grid(cdiv(cdiv(B, 4), kScatterThreads), E);
  const size_t smem = (size_t)2 * NB * sizeof(int);
  if (philox) hipLaunchKernelGGL(perm_scatter_kernel<true>, grid, dim3(kScatterThreads), smem, s, seed, stream_id0, ctr, keys, B, NB, shift, cnt, slots);
  else hipLaunchKernelGGL(perm_scatter_kernel<false>, grid, dim3(kScatterThreads), smem, s, seed, stream_id0, ctr, keys, B, NB, shift, cnt, slots);
  MPPO_CHECK_LAUNCH("perm_scatter_kernel");
  hipLaunchKernelGGL(perm_bucket_kernel, dim3(NB, E), dim3(256), 0, s, B, NB, cnt, slots, vals_in, out);
  MPPO_CHECK_LAUNCH("perm_bucket_kernel");
  return MPPO_OK;
}
}  // namespace mppo

// workspace of the stand-alone entry points (mppo_permutation, mppo_threefry_permutation): counters | slots | a key array | one order buffer
extern "C" size_t mppo_permutation_ws_bytes(int32_t B) {
  if (B < 1) return 0;
  const int NB = mppo::perm_buckets(B);
  return mppo::perm_counter_bytes(NB, 1) + mppo::perm_slot_bytes(NB, 1) + 2 * mppo::align_up((size_t)B * 4, 256);
}

namespace mppo {
int32_t permutation_ctr(unsigned long long seed, unsigned long long stream_id, const int* ctr, int B, int* idx, void* ws, size_t ws_bytes, hipStream_t s) {
  MPPO_REQUIRE(B >= 1 && idx && ws, "mppo_permutation: bad argument");
  MPPO_TRY(perm_check_size(B));
  if (ws_bytes < mppo_permutation_ws_bytes(B)) return fail(MPPO_ENOMEM, "mppo_permutation: workspace %zu < %zu bytes", ws_bytes, mppo_permutation_ws_bytes(B));
  const int NB = perm_buckets(B);
  int* cnt = static_cast<int*>(ws);
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(ws) + perm_counter_bytes(NB, 1));
  MPPO_TRY(perm_zero(cnt, NB + 1, s));
  return perm_sort_launch(true, seed, stream_id, ctr, nullptr, B, 1, cnt, slots, nullptr, idx, s);
}

size_t permutation_batch_ws_bytes(int B, int E) {
  if (B < 1 || E < 1) return 0;
  const int NB = perm_buckets(B);
  return perm_counter_bytes(NB, E) + perm_slot_bytes(NB, E);
}
// where the bucket counters live and how many words the caller zeroes after every use (the overflow word included)
void permutation_batch_counters(int B, int E, void* ws, size_t ws_bytes, int** ptr, int* n) {
  const bool ok = ws && ws_bytes >= permutation_batch_ws_bytes(B, E) && B <= kPermMaxB;
  *ptr = ok ? static_cast<int*>(ws) : nullptr;
  *n = ok ? E * perm_buckets(B) + 1 : 0;
}
// the counters must be zero before the first use (mppo_engine_reset)
int32_t permutation_batch_prepare(int B, int E, void* ws, size_t ws_bytes, hipStream_t s) {
  if (ws && ws_bytes >= permutation_batch_ws_bytes(B, E)) MPPO_CHECK_HIP(hipMemsetAsync(ws, 0, perm_counter_bytes(perm_buckets(B), E), s));
  return MPPO_OK;
}
// All E epoch permutations of an update in the two launches: block e of the result is exactly what permutation_ctr(stream_id0 + e) produces.
// The counters are zero on entry and are zeroed again by the caller (permutation_batch_counters).
int32_t permutation_batch_ctr(unsigned long long seed, unsigned long long stream_id0, const int* ctr, int B, int E, int* idx, void* ws, size_t ws_bytes,
                              hipStream_t s) {
  MPPO_REQUIRE(B >= 1 && E >= 1 && idx && ws, "permutation_batch: bad argument");
  MPPO_TRY(perm_check_size(B));
  if (ws_bytes < permutation_batch_ws_bytes(B, E)) return fail(MPPO_ENOMEM, "permutation_batch: workspace %zu < %zu bytes", ws_bytes, permutation_batch_ws_bytes(B, E));
  const int NB = perm_buckets(B);
  int* cnt = static_cast<int*>(ws);
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(ws) + perm_counter_bytes(NB, E));
  return perm_sort_launch(true, seed, stream_id0, ctr, nullptr, B, E, cnt, slots, nullptr, idx, s);
}
// jax.random.permutation (`_shuffle`): round r sorts the current order stably by random_bits(sort_keys[r]) - the same two launches with the
// round's threefry bits as keys and the previous round's order as payload; the buffers alternate so that the last round lands in `idx`
int32_t threefry_permutation(const unsigned* sort_keys, int rounds, int B, int* idx, void* ws, size_t ws_bytes, hipStream_t s) {
  MPPO_REQUIRE(B >= 1 && idx && ws && sort_keys && rounds >= 1, "threefry_permutation: bad argument");
  MPPO_TRY(perm_check_size(B));
  if (ws_bytes < mppo_permutation_ws_bytes(B)) return fail(MPPO_ENOMEM, "threefry_permutation: workspace %zu < %zu bytes", ws_bytes, mppo_permutation_ws_bytes(B));
  const int NB = perm_buckets(B);
  const size_t chunk = align_up((size_t)B * 4, 256);
  unsigned char* w = static_cast<unsigned char*>(ws);
  int* cnt = reinterpret_cast<int*>(w);
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(w + perm_counter_bytes(NB, 1));
  unsigned* keys = reinterpret_cast<unsigned*>(w + perm_counter_bytes(NB, 1) + perm_slot_bytes(NB, 1));
  int* tmp = reinterpret_cast<int*>(w + perm_counter_bytes(NB, 1) + perm_slot_bytes(NB, 1) + chunk);
  for (int r = 0; r < rounds; ++r) {
    int* out = ((rounds - 1 - r) % 2 == 0) ? idx : tmp;
    const int* in = r == 0 ? nullptr : (out == idx ? tmp : idx);  // (round 0 starts from the identity)
    MPPO_TRY(threefry_bits(sort_keys + 2 * r, (size_t)B, keys, nullptr, s));
    MPPO_TRY(perm_zero(cnt, NB + 1, s));
    MPPO_TRY(perm_sort_launch(false, 0ull, 0ull, nullptr, keys, B, 1, cnt, slots, in, out, s));
  }
  return MPPO_OK;
}
}  // namespace mppo

extern "C" int32_t mppo_threefry_permutation(const uint32_t* sort_keys, int32_t rounds, int32_t B, int32_t* idx, void* ws, size_t ws_bytes, void* stream) {
  return mppo::threefry_permutation(sort_keys, rounds, B, idx, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int32_t mppo_permutation(uint64_t seed, uint64_t stream_id, int32_t B, int32_t* idx, void* ws, size_t ws_bytes, void* stream) {
  return mppo::permutation_ctr(seed, stream_id, nullptr, B, idx, ws, ws_bytes, static_cast<hipStream_t>(stream));
}
