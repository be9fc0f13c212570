// TEST INFRASTRUCTURE — CPU stand-ins for the pieces of the engine that call vendor device
// libraries (rocPRIM sort, RCCL): same ABI, same results (the radix sort is stable, as is this one).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "mppo_common.h"
#include "ppo_layout.h"

extern "C" size_t mppo_permutation_ws_bytes(int32_t B) { return B < 1 ? 0 : 2 * (size_t)B * 4; }

extern "C" int32_t mppo_permutation(uint64_t seed, uint64_t stream_id, int32_t B, int32_t* idx, void* ws, size_t ws_bytes, void* stream) {
  using namespace mppo;
  MPPO_REQUIRE(B >= 1 && idx && ws, "mppo_permutation: bad argument");
  if (ws_bytes < mppo_permutation_ws_bytes(B)) return fail(MPPO_ENOMEM, "mppo_permutation: workspace too small");
  unsigned* keys = static_cast<unsigned*>(ws);
  int* vals = reinterpret_cast<int*>(keys + B);
  MPPO_TRY(perm_fill_keys(seed, stream_id, B, keys, vals, static_cast<hipStream_t>(stream)));
  std::vector<int> order(B);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return keys[a] < keys[b]; });
  for (int i = 0; i < B; ++i) idx[i] = vals[order[i]];
  return MPPO_OK;
}
