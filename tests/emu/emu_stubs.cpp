// TEST INFRASTRUCTURE — CPU stand-ins for the pieces of the engine that call vendor device
// libraries (rocPRIM sort, RCCL): same ABI, same results (the radix sort is stable, as is this one).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "mppo_common.h"
#include "platform.h"
#include "ppo_layout.h"

extern "C" size_t mppo_permutation_ws_bytes(int32_t B) { return B < 1 ? 0 : 2 * (size_t)B * 4; }

namespace mppo {
int32_t permutation_ctr(unsigned long long seed, unsigned long long stream_id, const int* ctr, int B, int* idx, void* ws, size_t ws_bytes, hipStream_t stream) {
  MPPO_REQUIRE(B >= 1 && idx && ws, "mppo_permutation: bad argument");
  if (ws_bytes < mppo_permutation_ws_bytes(B)) return fail(MPPO_ENOMEM, "mppo_permutation: workspace too small");
  unsigned* keys = static_cast<unsigned*>(ws);
  int* vals = reinterpret_cast<int*>(keys + B);
  MPPO_TRY(perm_fill_keys(seed, stream_id, ctr, B, keys, vals, stream));
  std::vector<int> order(B);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return keys[a] < keys[b]; });
  for (int i = 0; i < B; ++i) idx[i] = vals[order[i]];
  return MPPO_OK;
}

// no RCCL and no hipGraph on the CPU: world_size must be 1, updates run eagerly
struct Comm {};
struct GraphExec {};
int32_t comm_unique_id(void*) { return fail(MPPO_ENCCL, "RCCL is not available in the emulator build"); }
int32_t comm_create(const void*, int, int, Comm**) { return fail(MPPO_ENCCL, "RCCL is not available in the emulator build"); }
void comm_destroy(Comm*) {}
int32_t comm_allreduce_f32(Comm*, float*, size_t, hipStream_t) { return fail(MPPO_ENCCL, "RCCL is not available in the emulator build"); }
int32_t comm_allreduce_f64(Comm*, double*, size_t, hipStream_t) { return fail(MPPO_ENCCL, "RCCL is not available in the emulator build"); }
int32_t graph_begin(hipStream_t) { return fail(MPPO_EHIP, "hipGraph is not available in the emulator build"); }
int32_t graph_end(hipStream_t, GraphExec**) { return fail(MPPO_EHIP, "hipGraph is not available in the emulator build"); }
int32_t graph_launch(GraphExec*, hipStream_t) { return fail(MPPO_EHIP, "hipGraph is not available in the emulator build"); }
void graph_destroy(GraphExec*) {}
}  // namespace mppo

extern "C" int32_t mppo_permutation(uint64_t seed, uint64_t stream_id, int32_t B, int32_t* idx, void* ws, size_t ws_bytes, void* stream) {
  return mppo::permutation_ctr(seed, stream_id, nullptr, B, idx, ws, ws_bytes, static_cast<hipStream_t>(stream));
}
