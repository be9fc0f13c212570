// wgrad.h — descriptor of the weight-gradient launch (k_wgrad.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace mppo {

constexpr int kWgradMaxProb = 6;
constexpr int kSqSlots = 512;  // per-workgroup sums of squares consumed by the clip (adam_kernel adds exactly this many)

// C[M,N] = A^T . B over K minibatch rows.  A and B are K-QUAD operands (ppo_layout.h): element (k, c) at
// ((k/4) * cols + c) * 4 + k%4, with `acols` / `bcols` columns per quad row (>= M / N) as seen from the given base pointers.
// The result goes to the flat gradient at off_w (row-major [M][N]: the Flax kernel layout [in][out]) and the column sums of
// B (the bias gradient) to off_b.
struct WgradProb {
  const float* A;
  const float* B;
  int M, N, acols, bcols, lda, ldb;  // lda / ldb: columns of the whole quad row (the stride between quads, in float4 units)
  int off_w, off_b;
  int tiles_m, tiles_n, tile0;  // filled by wgrad_plan
  int thin_row0, thin_rows;     // a last row band of <= 4 rows gets no tiles: the first row band's workgroups contract it (wgrad_plan)
  // SPLIT-PAIR column order of an operand's quad rows (float networks: h1, h2, dZ1, dZ2 as the row pass stores them): the even columns
  // in the first half of the quad row, the odd ones in the second - column c sits at position (c >> 1) + (c & 1) * (ld / 2).  A lane of
  // the row pass owns two adjacent columns; this way each of its two 16-byte stores covers whole 256-byte runs across the lanes
  // (write-through stores of half lines doubled the bytes written)
  int a_split, b_split;
};

struct WgradArgs {
  WgradProb p[kWgradMaxProb];
  int count, Kq, qwave, ntiles;  // Kq quads in all, qwave quads per wave (a multiple of 8 = one stage)
  float* grad;                   // [P] gradient of the minibatch
  float* sq_partial;             // [kSqSlots] per-workgroup sums of squares of `grad` (optional)
  // log_std gradient + loss scalars (train.py:240-243) from the row pass's per-workgroup partials [nblk][4 + AP]
  int ls_off, A, AP, nblk;
  const float* partial;
  const float* log_std;
  float ent_coef, vf_coef, ent_weight;
  float* loss4;
  int npad, pad_off[13], pad_cnt[13];  // alignment words of the flat layout (ppo_layout.h): written as zeros
  // workgroup id -> its tile, resolved on the host (wgrad_plan): problem | row band << 4 | column band << 12 | (tile 0) << 20.  Tiles that
  // share operand bands sit on the same XCD.  (One scalar load; a tile NUMBER cost the kernel a search through p[].tile0 first:
  // up to `count` dependent scalar loads ahead of the first operand request.)
  // Several ranks (peer.h): bits 24..31 = the gradient slices this workgroup's stores fall into; slice_need[q] = the number of workgroups
  // that store into slice q (wgrad_launch fills both from the same tile table)
  unsigned order[kSqSlots];
  unsigned short slice_need[8];
  int dbg;  // timing experiments only (MPPO_WGRAD_DBG bit mask): 4 launch twice (warm operands), 8 big problems only
};

// upper bound of the launch's workgroups for a two-hidden-layer network (thin bands not discounted): the fused path is only taken
// when it fits the kSqSlots tile table (fused_supported, k_fused.hip); wider observation vectors run the layer-wise kernels
inline int wgrad_tile_bound(int O, int A, int H) {
  const int tn = (H + 31) / 32, th = (H + 31) / 32, to = (O + 31) / 32;
  return 2 * th * tn + 2 * to * tn + th * ((A + 31) / 32) + th;
}
int32_t wgrad_plan(WgradArgs& a, int mb);  // tile table + K ranges (a.count, a.p[].{M,N} set by the caller)
bool wgrad_supported(const WgradArgs& a);
struct PeerStep;  // peer.h
int32_t wgrad_launch(const WgradArgs& a, bool bf16, hipStream_t stream, const PeerStep* peer = nullptr);

}  // namespace mppo
