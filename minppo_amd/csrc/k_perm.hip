// k_perm.hip — random permutation of [0,B): sort of Philox keys (what jax.random.permutation does,
// reference minppo/train.py:258), using rocPRIM's device radix sort for the plain key/value sort.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "mppo_common.h"
#include "philox.h"
#include "ppo_layout.h"

namespace mppo {
static size_t sort_temp_bytes(int B) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, static_cast<unsigned*>(nullptr), static_cast<unsigned*>(nullptr), static_cast<int*>(nullptr),
                            static_cast<int*>(nullptr), (size_t)B, 0, 32, nullptr);
  return bytes;
}
}  // namespace mppo

extern "C" size_t mppo_permutation_ws_bytes(int32_t B) {
  if (B < 1) return 0;
  // keys_in, keys_out, vals_in  +  rocPRIM temporary storage
  return 3 * mppo::align_up((size_t)B * 4, 256) + mppo::align_up(mppo::sort_temp_bytes(B), 256);
}

namespace mppo {
int32_t permutation_ctr(unsigned long long seed, unsigned long long stream_id, const int* ctr, int B, int* idx, void* ws, size_t ws_bytes, hipStream_t s) {
  MPPO_REQUIRE(B >= 1 && idx && ws, "mppo_permutation: bad argument");
  if (ws_bytes < mppo_permutation_ws_bytes(B)) return fail(MPPO_ENOMEM, "mppo_permutation: workspace %zu < %zu bytes", ws_bytes, mppo_permutation_ws_bytes(B));
  const size_t chunk = align_up((size_t)B * 4, 256);
  unsigned char* w = static_cast<unsigned char*>(ws);
  unsigned* keys_in = reinterpret_cast<unsigned*>(w);
  unsigned* keys_out = reinterpret_cast<unsigned*>(w + chunk);
  int* vals_in = reinterpret_cast<int*>(w + 2 * chunk);
  void* temp = w + 3 * chunk;
  size_t temp_bytes = ws_bytes - 3 * chunk;
  MPPO_TRY(perm_fill_keys(seed, stream_id, ctr, B, keys_in, vals_in, s));
  MPPO_CHECK_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, idx, (size_t)B, 0, 32, s));
  return MPPO_OK;
}
static int bits_for(int E) { int b = 0; while ((1 << b) < E) ++b; return b; }
static size_t sort_temp_bytes64(size_t n, int end_bit) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, static_cast<unsigned long long*>(nullptr), static_cast<unsigned long long*>(nullptr), static_cast<int*>(nullptr),
                                  static_cast<int*>(nullptr), n, 0, end_bit, nullptr);
  return bytes;
}
// ---- all E permutations of an update in TWO launches (round 4; the sort above is one key launch + nine rocPRIM launches at the launch
// floor, 70 us of every update).  A permutation of B <= 131072 samples is the order of the 52-bit values (key << 20 | index): exactly what the
// stable sort of (key, index) pairs yields.  Launch 1 draws the Philox keys and scatters the values into 256 buckets by the key's top byte
// (one atomic counter per bucket: the order INSIDE a bucket does not matter, it is sorted next); launch 2, one workgroup per bucket, sorts
// its ~B/256 values in LDS (rank sort, bitonic above 256 values), finds its place by summing the counters of the buckets before it, and writes
// the indices.  The counters are zeroed again by the caller's next launch (permutation_batch_counters: the engine's end-of-update kernel does
// it; a counter that every bucket workgroup decrements through one atomic costs 10 us of contention): they are zero between updates
// whatever the caller does to the update index (a restored checkpoint, a repeated update).
constexpr int kPermBuckets = 256;
constexpr int kPermCap = 1024;  // slots per bucket: the mean is B / 256 <= 512, the standard deviation <= 23
constexpr int kPermIdxBits = 20;  // the index part of a value
constexpr int kPermMaxB = 131072;

static size_t perm_fast_bytes(int E) { return align_up((size_t)(2 * E * kPermBuckets + 64) * 4, 256) + (size_t)E * kPermBuckets * kPermCap * 8; }

constexpr int kScatterThreads = 1024;  // x 4 keys: 4096 values per workgroup = 16 per bucket - one global atomic reserves them all
__global__ void __launch_bounds__(kScatterThreads) perm_scatter_kernel(unsigned long long seed, unsigned long long stream_id0, const int* __restrict__ ctr, int B, int* __restrict__ cnt_base,
                                                                       unsigned long long* __restrict__ slots) {
  // bucket counts of THIS workgroup in LDS first (an LDS atomic hands every value its rank inside the workgroup's share of the bucket), then
  // one global atomic per bucket reserves the workgroup's range: 256 device-scope atomics per workgroup instead of 4096 (the one-atomic-per-value
  // form took 22 us at B = 40 960: 160 contended atomics per counter)
  __shared__ int s_cnt[kPermBuckets], s_base[kPermBuckets];
  const int t = threadIdx.x, q = blockIdx.x * kScatterThreads + t, e = blockIdx.y, E = gridDim.y;
  int* cnt = cnt_base + e * kPermBuckets;
  if (t < kPermBuckets) s_cnt[t] = 0;
  __syncthreads();
  const unsigned long long stream_id = stream_id0 + (unsigned long long)e;
  unsigned z[4] = {0u, 0u, 0u, 0u};
  int rank[4] = {0, 0, 0, 0};
  if (4 * q < B) {
    const U4 r = philox4x32((unsigned)q, (unsigned)ctr[0], (unsigned)stream_id, (unsigned)(stream_id >> 32) ^ 0x5045524Du, (unsigned)seed, (unsigned)(seed >> 32));  // = perm_keys_batch_kernel's keys
    z[0] = r.x; z[1] = r.y; z[2] = r.z; z[3] = r.w;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (4 * q + k < B) rank[k] = atomicAdd(&s_cnt[z[k] >> 24], 1);
  }
  __syncthreads();
  if (t < kPermBuckets) s_base[t] = s_cnt[t] ? atomicAdd(&cnt[t], s_cnt[t]) : 0;
  __syncthreads();
  if (4 * q < B) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = 4 * q + k;
      if (i < B) {
        const int b = (int)(z[k] >> 24), pos = s_base[b] + rank[k];
        if (pos < kPermCap) slots[((size_t)e * kPermBuckets + b) * kPermCap + pos] = ((unsigned long long)z[k] << kPermIdxBits) | (unsigned long long)i;
        else cnt_base[E * kPermBuckets] = 1;  // (cannot happen for B <= 131072 short of a 22-sigma event; recorded all the same)
      }
    }
  }
}

__global__ void __launch_bounds__(256) perm_bucket_kernel(int B, int* __restrict__ cnt_base, const unsigned long long* __restrict__ slots, int* __restrict__ idx) {
  __shared__ unsigned long long s[kPermCap];
  __shared__ int s_part[4];
  const int b = blockIdx.x, e = blockIdx.y, t = threadIdx.x;
  int* cnt = cnt_base + e * kPermBuckets;
  const int mine = cnt[t];  // thread t holds bucket t's count (256 threads = 256 buckets)
  // this bucket's count (a uniform read) and its offset: the counts of the buckets before it
  const int n_all = cnt[b];
  int before = t < b ? mine : 0;
  for (int m = 1; m < 64; m <<= 1) before += __shfl_xor(before, m);
  if ((t & 63) == 0) s_part[t >> 6] = before;
  const int n = n_all < kPermCap ? n_all : kPermCap;
  const unsigned long long* src = slots + ((size_t)e * kPermBuckets + b) * kPermCap;
  if (n <= 256) {
    // the usual case (B / 256 values per bucket on average; more go through the bitonic sort below): every thread holds one value and
    // counts the smaller ones - its place in the bucket.  The values are distinct (the index is part of them): the ranks are a permutation.
    const unsigned long long v = t < n ? src[t] : ~0ull;
    s[t] = v;
    __syncthreads();
    const int off = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    int rk = 0;
#pragma unroll 8
    for (int j = 0; j < n; ++j) rk += (int)(s[j] < v);
    if (t < n) idx[(size_t)e * B + off + rk] = (int)(v & ((1ull << kPermIdxBits) - 1));
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
    for (int k = t; k < n; k += 256) idx[(size_t)e * B + off + k] = (int)(s[k] & ((1ull << kPermIdxBits) - 1));
  }
}

size_t permutation_batch_ws_bytes(int B, int E) {
  if (B < 1 || E < 1) return 0;
  const size_t n = (size_t)B * E;  // keys_in, keys_out (64-bit), vals_in + rocPRIM temporary storage
  const size_t sort_bytes = 2 * align_up(n * 8, 256) + align_up(n * 4, 256) + align_up(sort_temp_bytes64(n, 32 + bits_for(E)), 256);
  return B <= kPermMaxB && perm_fast_bytes(E) > sort_bytes ? perm_fast_bytes(E) : sort_bytes;
}
// where the bucket counters of the two-launch form live (nullptr / 0 when that form is not in use for this shape): the caller zeroes them
// after every use
void permutation_batch_counters(int B, int E, void* ws, size_t ws_bytes, int** ptr, int* n) {
  const bool fast = B <= kPermMaxB && ws && ws_bytes >= perm_fast_bytes(E);
  *ptr = fast ? static_cast<int*>(ws) : nullptr;
  *n = fast ? E * kPermBuckets + 1 : 0;  // (+ the overflow word)
}
// the counters of the two-launch form must be zero before its first use (mppo_engine_reset)
int32_t permutation_batch_prepare(int B, int E, void* ws, size_t ws_bytes, hipStream_t s) {
  if (B <= kPermMaxB && ws && ws_bytes >= perm_fast_bytes(E)) MPPO_CHECK_HIP(hipMemsetAsync(ws, 0, (size_t)(2 * E * kPermBuckets + 64) * 4, s));
  return MPPO_OK;
}
// All E epoch permutations of an update: block e of the result is exactly what permutation_ctr(stream_id0 + e) produces.  B <= kPermMaxB and a
// device counter: the two launches above; otherwise ONE sort of the pairs (epoch << 32 | key, index) of all epochs - the sort is stable and
// the epoch is the most significant part of the key.
int32_t permutation_batch_ctr(unsigned long long seed, unsigned long long stream_id0, const int* ctr, int B, int E, int* idx, void* ws, size_t ws_bytes,
                              hipStream_t s) {
  MPPO_REQUIRE(B >= 1 && E >= 1 && idx && ws, "permutation_batch: bad argument");
  if (ws_bytes < permutation_batch_ws_bytes(B, E)) return fail(MPPO_ENOMEM, "permutation_batch: workspace %zu < %zu bytes", ws_bytes, permutation_batch_ws_bytes(B, E));
  if (B <= kPermMaxB && ctr && ws_bytes >= perm_fast_bytes(E)) {
    int* cnt = static_cast<int*>(ws);
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(ws) + align_up((size_t)(2 * E * kPermBuckets + 64) * 4, 256));
    hipLaunchKernelGGL(perm_scatter_kernel, dim3(cdiv(cdiv(B, 4), kScatterThreads), E), dim3(kScatterThreads), 0, s, seed, stream_id0, ctr, B, cnt, slots);
    MPPO_CHECK_LAUNCH("perm_scatter_kernel");
    hipLaunchKernelGGL(perm_bucket_kernel, dim3(kPermBuckets, E), dim3(256), 0, s, B, cnt, slots, idx);
    MPPO_CHECK_LAUNCH("perm_bucket_kernel");
    return MPPO_OK;
  }
  const size_t n = (size_t)B * E, c8 = align_up(n * 8, 256), c4 = align_up(n * 4, 256);
  unsigned char* w = static_cast<unsigned char*>(ws);
  unsigned long long* keys_in = reinterpret_cast<unsigned long long*>(w);
  unsigned long long* keys_out = reinterpret_cast<unsigned long long*>(w + c8);
  int* vals_in = reinterpret_cast<int*>(w + 2 * c8);
  void* temp = w + 2 * c8 + c4;
  size_t temp_bytes = ws_bytes - (2 * c8 + c4);
  MPPO_TRY(perm_fill_keys_batch(seed, stream_id0, ctr, B, E, keys_in, vals_in, s));
  MPPO_CHECK_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, idx, n, 0, 32 + bits_for(E), s));
  return MPPO_OK;
}
// jax.random.permutation (`_shuffle`): round r sorts the current order stably by random_bits(sort_keys[r]); the buffers alternate
// so that the last round lands in `idx`
int32_t threefry_permutation(const unsigned* sort_keys, int rounds, int B, int* idx, void* ws, size_t ws_bytes, hipStream_t s) {
  MPPO_REQUIRE(B >= 1 && idx && ws && sort_keys && rounds >= 1, "threefry_permutation: bad argument");
  if (ws_bytes < mppo_permutation_ws_bytes(B)) return fail(MPPO_ENOMEM, "threefry_permutation: workspace %zu < %zu bytes", ws_bytes, mppo_permutation_ws_bytes(B));
  const size_t chunk = align_up((size_t)B * 4, 256);
  unsigned char* w = static_cast<unsigned char*>(ws);
  unsigned* keys_in = reinterpret_cast<unsigned*>(w);
  unsigned* keys_out = reinterpret_cast<unsigned*>(w + chunk);
  int* tmp = reinterpret_cast<int*>(w + 2 * chunk);
  void* temp = w + 3 * chunk;
  size_t temp_bytes = ws_bytes - 3 * chunk;
  for (int r = 0; r < rounds; ++r) {
    int* out = ((rounds - 1 - r) % 2 == 0) ? idx : tmp;
    int* in = out == idx ? tmp : idx;
    MPPO_TRY(threefry_bits(sort_keys + 2 * r, (size_t)B, keys_in, r == 0 ? in : nullptr, s));
    MPPO_CHECK_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, in, out, (size_t)B, 0, 32, s));
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
