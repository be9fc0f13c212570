// k_perm.hip — random permutation of [0,B): sort of Philox keys (what jax.random.permutation does,
// reference minppo/train.py:258), using rocPRIM's device radix sort for the plain key/value sort.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "mppo_common.h"
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
size_t permutation_batch_ws_bytes(int B, int E) {
  if (B < 1 || E < 1) return 0;
  const size_t n = (size_t)B * E;  // keys_in, keys_out (64-bit), vals_in + rocPRIM temporary storage
  return 2 * align_up(n * 8, 256) + align_up(n * 4, 256) + align_up(sort_temp_bytes64(n, 32 + bits_for(E)), 256);
}
// All E epoch permutations of an update with ONE sort (a sort is ~7 launches at the launch floor whatever its size): the pairs
// (epoch << 32 | key, index) of all epochs are sorted together; the sort is stable and the epoch is the most significant part
// of the key, so block e of the result is exactly what permutation_ctr(stream_id0 + e) produces.
int32_t permutation_batch_ctr(unsigned long long seed, unsigned long long stream_id0, const int* ctr, int B, int E, int* idx, void* ws, size_t ws_bytes,
                              hipStream_t s) {
  MPPO_REQUIRE(B >= 1 && E >= 1 && idx && ws, "permutation_batch: bad argument");
  if (ws_bytes < permutation_batch_ws_bytes(B, E)) return fail(MPPO_ENOMEM, "permutation_batch: workspace %zu < %zu bytes", ws_bytes, permutation_batch_ws_bytes(B, E));
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
