// wgrad.h — descriptor of the split-K weight-gradient launch (k_wgrad.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace mppo {

constexpr int kWgradMaxProb = 6;

// C[M,N] = A^T . B, A stored [K, lda] (columns [0,M) used), B stored [K, ldb] (columns [0,N) used); the result goes to
// offset off_w of K-chunk slab `c` (row-major [M][N], the Flax kernel layout [in][out]) and the column sums of B to off_b.
struct WgradProb {
  const float* A;
  const float* B;
  int M, N, lda, ldb;
  int acols, bcols;            // readable columns of a row of A / B from the given base pointers (multiples of 4, >= M / N)
  int off_w, off_b;            // offsets (floats) inside the flat gradient / slab; off_b < 0: no bias gradient
  int tiles_m, tiles_n, tile0;  // filled by wgrad_plan
};

struct WgradArgs {
  WgradProb p[kWgradMaxProb];
  int count, K, ksplit, kchunk, ntiles;
  size_t slab_stride;   // floats between K-chunk slabs
  float* slabs;         // [ksplit][slab_stride] partial gradients
  int dbg;  // timing experiments only (MPPO_WGRAD_DBG bit mask): 1 no MFMAs, 2 no stores, 8 no global loads
};

int32_t wgrad_plan(WgradArgs& a, int K);  // tile table + K chunking (a.count, a.p[].{M,N}, a.ksplit set by the caller)
bool wgrad_supported(const WgradArgs& a);
int32_t wgrad_launch(const WgradArgs& a, bool bf16, hipStream_t stream);

}  // namespace mppo
