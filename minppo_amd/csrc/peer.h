// peer.h — the ranks' gradient exchange WITHOUT a collective library: every rank (one process per GPU) owns one
// exchange buffer in fine-grained device memory, exported with hipIpcGetMemHandle and mapped by every peer of the node
// (hipIpcOpenMemHandle: a peer GPU's HBM over xGMI, or - several ranks on ONE GPU, how the path is tested on a one-GPU box - the same
// HBM).  Replaces RCCL's all-reduce of the [P] gradient per optimizer step (the reference has no data parallelism at all:
// minppo/train.py:136,140 vmap only) by a two-hop exchange fused into the kernels that stand on either side of it:
//
//   wgrad_kernel<.., PEER>   writes the rank's local gradient into `pub` of its own buffer with system-scope write-through stores.  The
//                            gradient is cut into G slices (slice q is reduced by rank q); every workgroup knows which slices its
//                            stores fall into (host-side tile table, k_wgrad.hip) and counts itself into those slices' arrival counters;
//                            the workgroup that completes SLICE q raises wg_done[rank] in the buffer of rank q - the only reader of that
//                            slice - as soon as that slice is stored, not when the whole launch has ended (round 4: a fan-in of ~35
//                            arrivals per counter on eight counters instead of 256 on one, and no slice waits for another's stragglers).
//   adam_kernel<PEER>        phase A (the first nA workgroups): wait for every peer's wg_done, PULL slice `rank` of every rank's
//                            `pub` (1/G of the gradient from each of G - 1 peers: all links busy, P/G floats per link), add the G
//                            contributions in rank order, PUSH the reduced slice and its sum of squares into `red` of EVERY rank's
//                            buffer - TAGGED (round 4): every float of `red` travels as an 8-byte pair (value, epoch), two pairs per
//                            16-byte store.  Phase B (every workgroup): load its own floats of `red` in its OWN buffer until their tags
//                            carry this step's epoch, then clip_by_global_norm + adam (train.py:115-124,248).  No flag follows the
//                            data, so the pusher does not wait for its stores to be acknowledged before it may say so (that drain was a
//                            device-to-device round trip on the critical path of every step, and the flag one more one-way trip):
//                            what a reader needs is in the very bytes it reads - the low-latency protocol of the collective
//                            libraries, for the 1 MB that crosses the links 128 times per update.  (Ranks that SHARE a GPU keep
//                            the drain and the red_done flags for their one-wave wait kernels: nothing large may spin there.)
//
// A slice is reduced ONCE, by its owner, and broadcast: the replicas see bit-identical gradients by construction.  Two hops of
// latency and 2 P/G floats per link and step (RCCL's ring: 2 (G-1) hops); no extra launch, no host involvement, capturable in the
// hipGraph like any kernel.  Flags are epochs (optimizer steps since the buffers were created, a device word advanced once per
// update): nothing is ever reset, a flag can be waited for with >=.  Reuse is safe with single buffers: a rank reaches wgrad of
// step s + 1 only after its adam of step s, which waited for every owner's reduction of step s, i.e. for the last reader of
// every `pub`; and an owner pushes step s + 1 into `red` only after every peer's wgrad of step s + 1, i.e. after every reader of
// step s's `red` has finished.  Every spin is bounded (time limit -> error word, the kernel runs to its end with garbage).
#pragma once
#include <wave_ops.h>

#include "wgrad.h"

namespace mppo {

constexpr int kPeerMaxRanks = 8;
constexpr int kPeerThreads = 256;  // threads of a workgroup that reduces a piece (= adam_kernel's)

struct PeerHdr {
  int wg_done[kPeerMaxRanks][16];   // [q][0]: epoch at which slice `this rank` of rank q's local gradient was complete in q's `pub` (one 64-byte line per writer)
  int adv_done[kPeerMaxRanks][16];  // [q][0]: update epoch of rank q's published advantage sums
  int red_done[kSqSlots];           // [q * nA + b]: epoch of piece b of slice q in `red` (raised only for ranks that share a GPU: see above)
  // fan-in counters of this rank's own gradient-writing launches, one line each: [q][0] (q < world) the workgroups that store into
  // slice q, [kPeerMaxRanks][0] a launch whose workgroups all store everywhere (publish kernel).  The last arriver takes the counter
  // back to 0: the next launch's workgroups arrive only after this launch has ended (stream order), nothing wraps, and the two kinds
  // of launch may alternate.
  int arrive[kPeerMaxRanks + 1][16];
  int error;                        // != 0: a wait ran into its time limit (the run is invalid)
  int epoch[2];                     // optimizer steps / updates completed since creation (peer_advance_kernel)
  int error_info[4];                // the first wait that timed out: kind (1 wg_done, 2 red_done, 3 adv_done), index, epoch waited for, value seen
  int probe[kPeerMaxRanks];         // [q]: the count of the latency probe's flags rank q has stored here (peer_latency: a ping-pong of one word between two ranks)
  int pad[33];
};
static_assert(sizeof(PeerHdr) % 256 == 0, "the regions behind the header stay 256-byte aligned");

struct PeerView {  // kernel argument: the G exchange buffers as mapped in THIS process
  unsigned char* base[kPeerMaxRanks];
  int rank, world;
  int nA, K;    // pieces per slice; float4 per thread and piece (piece = 256 K float4)
  int P4, S4;   // float4 in the gradient / in a slice
  unsigned pub_off, red_off, adv_off;  // byte offsets of pub [4 P4] floats, red [4 P4 + kSqSlots] (value, epoch) PAIRS, adv [n] doubles
  unsigned long long limit_ticks;      // time limit of one wait, 100 MHz ticks
  int poll_rmw;                        // experiment (MPPO_PEER_POLL_RMW=1): poll with a system-scope atomic OR of 0 instead of a load
};

// what a launch needs to take part: the view, where the epoch lives, and the optimizer step inside the update.
// mode (host side only): 0 fused - the Adam launch reduces, waits and applies (ranks on distinct GPUs); 1 split - two launches, each
// waiting for itself; 2 shared - the ranks share a GPU: one-wave wait kernels, nothing else waits (see clip_adam, k_ppo.hip)
struct PeerStep { PeerView v; const int* epoch; int step; int mode; };

__device__ __forceinline__ PeerHdr* peer_hdr(const PeerView& v, int q) { return reinterpret_cast<PeerHdr*>(v.base[q]); }

// waits until *flag has reached `epoch` (wrap-safe); gives up at the time limit and records it.  Once a wait of this rank has timed
// out no later one waits at all: a dead peer costs ONE time limit, not one per optimizer step.
__device__ __forceinline__ void peer_wait(const int* flag, int epoch, PeerHdr* me, unsigned long long limit, int kind, int index, int rmw = 0) {
  unsigned long long t0 = 0;
  unsigned spins = 0;
  int seen;
  while ((int)((seen = rmw ? sys_poll_rmw(flag) : sys_load_i32(flag)) - epoch) < 0) {
    if ((spins++ & 127u) == 0) {
      if (sys_load_i32(&me->error)) return;
      const unsigned long long now = realtime_ticks();
      if (t0 == 0) t0 = now;
      if (now - t0 > limit) {
        if (agent_fetch_add(&me->error, 1) == 0) { me->error_info[0] = kind; me->error_info[1] = index; me->error_info[2] = epoch; me->error_info[3] = seen; }
        return;
      }
    }
    spin_pause();
  }
}

// End of a workgroup that wrote part of this rank's local gradient into `pub` (system-scope stores): called by EVERY thread of EVERY
// workgroup after its last store.  `mask`: the slices this workgroup's stores fall into, need[q]: how many workgroups of the launch
// store into slice q (both from the host's tile table).  Whoever completes slice q tells rank q, the slice's only reader.
__device__ __forceinline__ void peer_publish_done(const PeerView& v, int epoch, unsigned mask, const unsigned short* need) {
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    PeerHdr* me = peer_hdr(v, v.rank);
    int old[kPeerMaxRanks];
#pragma unroll
    for (int q = 0; q < kPeerMaxRanks; ++q)  // (all the adds are requested before the first result is looked at)
      if (q < v.world && ((mask >> q) & 1u)) old[q] = agent_fetch_add(&me->arrive[q][0], 1);
#pragma unroll
    for (int q = 0; q < kPeerMaxRanks; ++q)
      if (q < v.world && ((mask >> q) & 1u) && old[q] + 1 == (int)need[q]) {
        agent_fetch_add(&me->arrive[q][0], -(int)need[q]);
        if (q != v.rank) sys_store_i32(&peer_hdr(v, q)->wg_done[v.rank][0], epoch);
      }
  }
}

// the same for a launch whose workgroups store all over the gradient (peer_publish_kernel): one counter, the last of the `nblocks`
// workgroups tells every peer
__device__ __forceinline__ void peer_publish_all_done(const PeerView& v, int epoch, int nblocks) {
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    PeerHdr* me = peer_hdr(v, v.rank);
    if (agent_fetch_add(&me->arrive[kPeerMaxRanks][0], 1) + 1 == nblocks) {
      agent_fetch_add(&me->arrive[kPeerMaxRanks][0], -nblocks);
#pragma unroll
      for (int q = 0; q < kPeerMaxRanks; ++q)
        if (q < v.world && q != v.rank) sys_store_i32(&peer_hdr(v, q)->wg_done[v.rank][0], epoch);
    }
  }
}

// ---- the tagged reduced gradient: float k of `red` is the pair at byte 8 k = (value, bits of the epoch it belongs to) ----
__device__ __forceinline__ bool peer_tag_ok(float tag, int epoch) { return (int)(__builtin_bit_cast(int, tag) - epoch) >= 0; }

// bounded wait shared by the two tagged loads below: returns false when the wait must end (error word set by somebody, or time is up)
struct PeerSpin {
  unsigned long long t0 = 0;
  unsigned spins = 0;
  __device__ __forceinline__ bool keep_waiting(PeerHdr* me, unsigned long long limit, int index, int epoch, float seen_tag) {
    if ((spins++ & 127u) == 0) {
      if (sys_load_i32(&me->error)) return false;
      const unsigned long long now = realtime_ticks();
      if (t0 == 0) t0 = now;
      if (now - t0 > limit) {
        if (agent_fetch_add(&me->error, 1) == 0) { me->error_info[0] = 2; me->error_info[1] = index; me->error_info[2] = epoch; me->error_info[3] = __builtin_bit_cast(int, seen_tag); }
        return false;
      }
    }
    spin_pause();
    return true;
  }
};

// floats [i, i + 4) of the reduced gradient (i a multiple of 4) once all four carry `epoch`; `red` = this rank's own `red` region (wave-uniform)
__device__ __forceinline__ float4 peer_load_reduced4(const PeerView& v, const float* red, size_t i, int epoch) {
  PeerHdr* me = peer_hdr(v, v.rank);
  PeerSpin sp;
  float4 lo, hi;
  for (;;) {
    lo = sys_load_f4(red, i * 8);
    hi = sys_load_f4(red, i * 8 + 16);
    if ((int)peer_tag_ok(lo.y, epoch) & (int)peer_tag_ok(lo.w, epoch) & (int)peer_tag_ok(hi.y, epoch) & (int)peer_tag_ok(hi.w, epoch)) break;
    if (!sp.keep_waiting(me, v.limit_ticks, (int)(i >> 2), epoch, lo.y)) break;
  }
  return make_float4(lo.x, lo.z, hi.x, hi.z);
}
// one tagged float (the sums of squares behind the gradient: slot k is float 4 P4 + k)
__device__ __forceinline__ float peer_load_reduced1(const PeerView& v, const float* red, size_t k, int epoch) {
  PeerHdr* me = peer_hdr(v, v.rank);
  PeerSpin sp;
  float2 pr;
  for (;;) {
    pr = sys_load_f2(red + 2 * k);
    if (peer_tag_ok(pr.y, epoch)) break;
    if (!sp.keep_waiting(me, v.limit_ticks, (int)k, epoch, pr.y)) break;
  }
  return pr.x;
}

// Phase A of the fused exchange: workgroup b < nA (kPeerThreads threads) reduces piece b of slice `rank` and broadcasts it, tagged.
// flags: also drain the stores and raise red_done (ranks sharing a GPU: their one-wave wait kernel polls those words).
__device__ __forceinline__ void peer_reduce_piece(const PeerView& v, int epoch, int b, bool wait = true, bool flags = false) {
  __shared__ float s_sq[kPeerThreads / 64];
  const int t = threadIdx.x;
  PeerHdr* me = peer_hdr(v, v.rank);
  const float ef = __builtin_bit_cast(float, epoch);
  if (wait && t < v.world && t != v.rank) peer_wait(&me->wg_done[t][0], epoch, me, v.limit_ticks, 1, t, v.poll_rmw);
  __syncthreads();
  float sq = 0.f;
  for (int k = 0; k < v.K; ++k) {
    const int s4 = (b * v.K + k) * kPeerThreads + t, i4 = v.rank * v.S4 + s4;  // float4 index in the slice / in the gradient
    if (s4 < v.S4 && i4 < v.P4) {
      // the G contributions are added in rank order (one fixed summation chain), four loads in flight at a time (the kernel that
      // hosts this is held to 64 registers, see adam_kernel)
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int h = 0; h < kPeerMaxRanks; h += 4) {
        if (h < v.world) {
          float4 g[4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (h + q < v.world) g[q] = sys_load_f4(v.base[h + q] + v.pub_off, (size_t)i4 * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (h + q < v.world) {
              if (h + q == 0) a = g[0];
              else { a.x += g[q].x; a.y += g[q].y; a.z += g[q].z; a.w += g[q].w; }
            }
        }
      }
      const float4 lo = make_float4(a.x, ef, a.y, ef), hi = make_float4(a.z, ef, a.w, ef);
#pragma unroll
      for (int q = 0; q < kPeerMaxRanks; ++q)
        if (q < v.world) {
          sys_store_f4(v.base[q] + v.red_off, (size_t)i4 * 32, lo);
          sys_store_f4(v.base[q] + v.red_off, (size_t)i4 * 32 + 16, hi);
        }
      sq += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    }
  }
  sq = wave_sum(sq);
  if ((t & 63) == 0) s_sq[t >> 6] = sq;
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
    for (int w = 0; w < kPeerThreads / 64; ++w) s += s_sq[w];
#pragma unroll
    for (int q = 0; q < kPeerMaxRanks; ++q)
      if (q < v.world) sys_store_f2(reinterpret_cast<float*>(v.base[q] + v.red_off) + 2 * ((size_t)4 * v.P4 + v.rank * v.nA + b), make_float2(s, ef));
  }
  if (flags) {
    drain_stores();
    __syncthreads();
    if (t == 0) {
#pragma unroll
      for (int q = 0; q < kPeerMaxRanks; ++q)
        if (q < v.world) sys_store_i32(&peer_hdr(v, q)->red_done[v.rank * v.nA + b], epoch);
    }
  }
}

// ---- host side (k_peer.hip) ----
struct PeerComm;
int32_t peer_wait_launch(const PeerStep& ps, int kind, hipStream_t s);  // one wave waits for every peer's gradient (1) / every reduced piece (2)
int32_t peer_create(int rank, int world, size_t P, size_t adv_doubles, PeerComm** out, void* handle64);
int32_t peer_connect(PeerComm* c, const void* handles, int shared_device);  // world x 64 bytes, rank order; shared_device: several ranks on one GPU
bool peer_connected(const PeerComm* c);
bool peer_has_local(const PeerComm* c);  // some peer's buffer belongs to an engine of THIS process (found in the registry, not opened through hipIpc)
void peer_destroy(PeerComm* c);
PeerStep peer_step(const PeerComm* c, int step);
int peer_mode(const PeerComm* c);
float* peer_pub(const PeerComm* c);                 // this rank's local gradient [P] (what the weight-gradient launch writes)
const float* peer_red(const PeerComm* c);           // the reduced gradient [P] + kSqSlots sums of squares as (value, epoch) pairs: float k at [2 k] (what Adam reads)
int32_t peer_publish(const PeerComm* c, const float* grad, size_t P, int step, hipStream_t s);  // a gradient computed elsewhere -> pub + signal
int32_t peer_allreduce_f64(const PeerComm* c, double* buf, size_t n, hipStream_t s);            // in-place sum over the ranks, once per update
int32_t peer_advance(const PeerComm* c, int steps, hipStream_t s);                              // end of an update
int32_t peer_status(const PeerComm* c, int32_t* timed_out, int32_t* info8);
// one-way latency of a system-scope flag between this rank and `other` (both ranks call it at the same time, one as initiator): `iters` round
// trips of one word through the two exchange buffers' headers, timed by the initiator's kernel on the 100 MHz clock; synchronises
int32_t peer_latency(PeerComm* c, int other, int iters, int initiator, hipStream_t s, double* one_way_us);
unsigned long long peer_set_limit_ms(PeerComm* c, double ms);  // time limit of a wait; returns the previous one in the same unit (ms <= 0: leave it, just return it)                      // synchronises the device; info8 (optional): what timed out + counters

}  // namespace mppo
