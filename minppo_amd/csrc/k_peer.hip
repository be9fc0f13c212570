// k_peer.hip — host side of the ranks' peer-to-peer exchange (peer.h): the exchange buffer (fine-grained device memory, exported /
// imported through hipIpc handles), and the small kernels around the two fused ones: the once-per-update all-reduce of the advantage
// sums, the publish kernel for gradients that were not written by wgrad_kernel<.., PEER>, the epoch counter.
#include "peer.h"

#include <array>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "mppo_common.h"

namespace mppo {

struct PeerComm {
  int rank, world;
  size_t P, adv_doubles, bytes;
  unsigned char* mine;
  void* mapped[kPeerMaxRanks];
  bool local[kPeerMaxRanks];  // a peer engine of THIS process: its buffer is used through its own pointer, nothing to unmap
  std::array<unsigned char, 64> handle;
  bool connected;
  int mode;  // PeerStep::mode
  PeerView view;
  int probe_count[kPeerMaxRanks] = {};  // flags exchanged with rank q by the latency probe so far (peer_latency)
};

// Exchange buffers created in this process, by exported handle: hipIpcOpenMemHandle refuses a handle of the opening process, so several
// engines in ONE process (one per GPU driven by a single host process, or - how eight ranks are run on a one-GPU box whose process limit is
// six - two ranks per process) reach each other's buffers through the pointers themselves.
static std::mutex g_reg_mu;
static std::map<std::array<unsigned char, 64>, void*>& registry() { static std::map<std::array<unsigned char, 64>, void*> m; return m; }

// The waits of the shared-GPU form as kernels of ONE wave: whatever else the GPU has to run for a peer rank finds room beside them.
__global__ void __launch_bounds__(64) peer_wait_kernel(PeerView v, const int* epoch_base, int step, int kind) {
  const int epoch = epoch_base[0] + step + 1, t = threadIdx.x;
  PeerHdr* me = peer_hdr(v, v.rank);
  if (kind == 1) {
    if (t < v.world && t != v.rank) peer_wait(&me->wg_done[t][0], epoch, me, v.limit_ticks, 1, t, v.poll_rmw);
  } else {
    for (int s = t; s < v.world * v.nA; s += 64) peer_wait(&me->red_done[s], epoch, me, v.limit_ticks, 2, s, v.poll_rmw);
  }
}

// [E*M*2] float64 advantage sums (train.py:235 needs the statistics of the GLOBAL minibatch): one workgroup publishes this
// rank's sums, tells the peers, waits for theirs and adds all G vectors in rank order (every rank computes the identical sum).
__global__ void __launch_bounds__(256) peer_allreduce_f64_kernel(PeerView v, double* buf, int n) {
  PeerHdr* me = peer_hdr(v, v.rank);
  const int t = threadIdx.x, epoch = me->epoch[1] + 1;
  double* mine = reinterpret_cast<double*>(v.base[v.rank] + v.adv_off);
  for (int i = t; i < n; i += blockDim.x) sys_store_f64(mine + i, buf[i]);
  drain_stores();
  __syncthreads();
  if (t == 0) {
#pragma unroll
    for (int q = 0; q < kPeerMaxRanks; ++q)
      if (q < v.world && q != v.rank) sys_store_i32(&peer_hdr(v, q)->adv_done[v.rank][0], epoch);
  }
  if (t < v.world && t != v.rank) peer_wait(&me->adv_done[t][0], epoch, me, v.limit_ticks, 3, t);
  __syncthreads();
  for (int i = t; i < n; i += blockDim.x) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < kPeerMaxRanks; ++q)
      if (q < v.world) s += sys_load_f64(reinterpret_cast<const double*>(v.base[q] + v.adv_off) + i);
    buf[i] = s;
  }
}

// a gradient computed by kernels that do not know about the exchange (the layer-wise path) -> pub, then the completion signal
__global__ void __launch_bounds__(256) peer_publish_kernel(PeerView v, const float* __restrict__ grad, const int* epoch_base, int step) {
  const int epoch = epoch_base[0] + step + 1;
  const int nthr = gridDim.x * blockDim.x;
  for (int i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < v.P4; i4 += nthr)
    sys_store_f4(v.base[v.rank] + v.pub_off, (size_t)i4 * 16, *reinterpret_cast<const float4*>(grad + 4 * (size_t)i4));
  peer_publish_all_done(v, epoch, (int)gridDim.x);
}

__global__ void peer_advance_kernel(PeerView v, int steps) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { PeerHdr* me = peer_hdr(v, v.rank); me->epoch[0] += steps; me->epoch[1] += 1; }
}

static unsigned long long limit_ticks() {
  const char* e = getenv("MPPO_PEER_TIMEOUT_MS");
#ifdef MPPO_EMU
  const double ms = e ? atof(e) : 600000.0;  // emulated peers are slow
#else
  const double ms = e ? atof(e) : 60000.0;  // host-side skew between ranks (a checkpoint on a slow disk, a paused process) must not end a run
#endif
  return (unsigned long long)(ms * 1e5);
}

int32_t peer_create(int rank, int world, size_t P, size_t adv_doubles, PeerComm** out, void* handle64) {
  MPPO_REQUIRE(world >= 2 && world <= kPeerMaxRanks && rank >= 0 && rank < world, "peer exchange: %d ranks (2 .. %d on one node)", world, kPeerMaxRanks);
  MPPO_REQUIRE(P % 4 == 0 && P >= 4, "peer exchange: the flat gradient must be a whole number of float4 (P = %zu)", P);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI passes hipIpc handles as 64 bytes");
  PeerComm* c = new PeerComm();
  c->rank = rank; c->world = world; c->P = P; c->adv_doubles = adv_doubles; c->connected = false;
  for (int q = 0; q < kPeerMaxRanks; ++q) c->mapped[q] = nullptr;
  PeerView& v = c->view;
  memset(&v, 0, sizeof(v));
  v.rank = rank; v.world = world; v.P4 = (int)(P / 4); v.S4 = (v.P4 + world - 1) / world;
  // pieces: 256 K float4 each, K as small as the G * nA <= kSqSlots flag / sum-of-squares slots allow
  v.K = 1;
  while (world * ((v.S4 + kPeerThreads * v.K - 1) / (kPeerThreads * v.K)) > kSqSlots) ++v.K;
  v.nA = (v.S4 + kPeerThreads * v.K - 1) / (kPeerThreads * v.K);
  v.pub_off = (unsigned)sizeof(PeerHdr);
  v.red_off = (unsigned)align_up(v.pub_off + P * 4, 256);
  v.adv_off = (unsigned)align_up(v.red_off + (P + kSqSlots) * 8, 256);  // (tagged: 8 bytes per float)
  c->bytes = align_up(v.adv_off + adv_doubles * 8, 256);
  v.limit_ticks = limit_ticks();
  { const char* e = MPPO_EXPERIMENT_ENV("MPPO_PEER_POLL_RMW"); v.poll_rmw = e && e[0] == '1'; }
  // fine-grained device memory: coherent for accesses from other agents (what RCCL allocates for its own peer buffers);
  // MPPO_PEER_ALLOC=uncached | plain select the other two kinds hipIpc can export (measurements)
  const char* kind = MPPO_EXPERIMENT_ENV("MPPO_PEER_ALLOC");
  hipError_t e;
  if (kind && !strcmp(kind, "plain")) e = hipMalloc(reinterpret_cast<void**>(&c->mine), c->bytes);
  else e = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->mine), c->bytes, kind && !strcmp(kind, "uncached") ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
  if (e != hipSuccess) { delete c; return fail(MPPO_EHIP, "peer exchange: allocating %zu bytes of exchange memory failed: %s", c->bytes, hipGetErrorString(e)); }
  e = hipMemset(c->mine, 0, c->bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipIpcGetMemHandle(static_cast<hipIpcMemHandle_t*>(handle64), c->mine);
  if (e != hipSuccess) { (void)hipFree(c->mine); delete c; return fail(MPPO_EHIP, "peer exchange: exporting the exchange buffer failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(e)); }
  c->mapped[rank] = c->mine;
  v.base[rank] = c->mine;
  for (int q = 0; q < kPeerMaxRanks; ++q) c->local[q] = false;
  memcpy(c->handle.data(), handle64, 64);
  { std::lock_guard<std::mutex> lk(g_reg_mu); registry()[c->handle] = c->mine; }
  *out = c;
  return MPPO_OK;
}

int32_t peer_wait_launch(const PeerStep& ps, int kind, hipStream_t s) {
  hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, s, ps.v, ps.epoch, ps.step, kind);
  MPPO_CHECK_LAUNCH("peer_wait_kernel");
  return MPPO_OK;
}

int32_t peer_connect(PeerComm* c, const void* handles, int shared_device) {
  MPPO_REQUIRE(c && handles, "peer_connect: null argument");
  MPPO_REQUIRE(!c->connected, "peer_connect: already connected");
  const hipIpcMemHandle_t* h = static_cast<const hipIpcMemHandle_t*>(handles);
  for (int q = 0; q < c->world; ++q) {
    if (q == c->rank) continue;
    void* p = nullptr;
    hipIpcMemHandle_t hq;
    memcpy(&hq, &h[q], sizeof(hq));
    {
      std::array<unsigned char, 64> key;
      memcpy(key.data(), &h[q], 64);
      std::lock_guard<std::mutex> lk(g_reg_mu);
      auto it = registry().find(key);
      if (it != registry().end()) { p = it->second; c->local[q] = true; }
    }
    if (c->local[q]) {
#ifndef MPPO_EMU
      // an engine of this process on ANOTHER GPU: its memory is reached once peer access is on (same GPU: nothing to do)
      hipPointerAttribute_t at;
      int dev = -1;
      if (hipPointerGetAttributes(&at, p) == hipSuccess && hipGetDevice(&dev) == hipSuccess && at.device != dev) {
        const hipError_t pe = hipDeviceEnablePeerAccess(at.device, 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return fail(MPPO_EHIP, "peer exchange: peer access to GPU %d (rank %d, same process) failed: %s", at.device, q, hipGetErrorString(pe));
        (void)hipGetLastError();
      }
#endif
    } else {
      const hipError_t e = hipIpcOpenMemHandle(&p, hq, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) return fail(MPPO_EHIP, "peer exchange: mapping rank %d's exchange buffer failed: %s", q, hipGetErrorString(e));
    }
    c->mapped[q] = p;
    c->view.base[q] = static_cast<unsigned char*>(p);
  }
  c->connected = true;
  // fused unless the ranks share a GPU; MPPO_PEER_MODE=fused | split | shared overrides (measurements, tests)
  c->mode = shared_device ? 2 : 0;
  if (const char* e = getenv("MPPO_PEER_MODE")) c->mode = !strcmp(e, "shared") ? 2 : !strcmp(e, "split") ? 1 : !strcmp(e, "fused") ? 0 : c->mode;
  return MPPO_OK;
}

bool peer_connected(const PeerComm* c) { return c && c->connected; }
bool peer_has_local(const PeerComm* c) {
  if (!c) return false;
  for (int q = 0; q < c->world; ++q)
    if (q != c->rank && c->local[q]) return true;
  return false;
}

void peer_destroy(PeerComm* c) {
  if (!c) return;
  (void)hipDeviceSynchronize();
  for (int q = 0; q < c->world; ++q)
    if (q != c->rank && c->mapped[q] && !c->local[q]) (void)hipIpcCloseMemHandle(c->mapped[q]);
  { std::lock_guard<std::mutex> lk(g_reg_mu); registry().erase(c->handle); }
  (void)hipFree(c->mine);
  delete c;
}

PeerStep peer_step(const PeerComm* c, int step) { return PeerStep{c->view, &reinterpret_cast<const PeerHdr*>(c->mine)->epoch[0], step, c->mode}; }
int peer_mode(const PeerComm* c) { return c->mode; }
float* peer_pub(const PeerComm* c) { return reinterpret_cast<float*>(c->mine + c->view.pub_off); }
const float* peer_red(const PeerComm* c) { return reinterpret_cast<const float*>(c->mine + c->view.red_off); }

int32_t peer_publish(const PeerComm* c, const float* grad, size_t P, int step, hipStream_t s) {
  MPPO_REQUIRE(c && c->connected && P == c->P && (reinterpret_cast<uintptr_t>(grad) & 15) == 0, "peer_publish: bad argument");
  hipLaunchKernelGGL(peer_publish_kernel, dim3(cdiv((long)c->view.P4, 256)), dim3(256), 0, s, c->view, grad, &reinterpret_cast<const PeerHdr*>(c->mine)->epoch[0], step);
  MPPO_CHECK_LAUNCH("peer_publish_kernel");
  return MPPO_OK;
}

int32_t peer_allreduce_f64(const PeerComm* c, double* buf, size_t n, hipStream_t s) {
  MPPO_REQUIRE(c && c->connected && n <= c->adv_doubles, "peer_allreduce_f64: %zu values exceed the exchange buffer's %zu", n, c ? c->adv_doubles : (size_t)0);
  hipLaunchKernelGGL(peer_allreduce_f64_kernel, dim3(1), dim3(256), 0, s, c->view, buf, (int)n);
  MPPO_CHECK_LAUNCH("peer_allreduce_f64_kernel");
  return MPPO_OK;
}

int32_t peer_advance(const PeerComm* c, int steps, hipStream_t s) {
  hipLaunchKernelGGL(peer_advance_kernel, dim3(1), dim3(64), 0, s, c->view, steps);
  MPPO_CHECK_LAUNCH("peer_advance_kernel");
  return MPPO_OK;
}

// latency probe: `iters` round trips of one flag between two ranks (the initiator stores first); one lane per rank, bounded waits
__global__ void peer_probe_kernel(PeerView v, int other, int iters, int initiator, int base, unsigned long long* ticks_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  PeerHdr* me = peer_hdr(v, v.rank);
  PeerHdr* peer = peer_hdr(v, other);
  // base: the flags this pair has exchanged before (counted on the host by both sides: every probe is symmetric) - NOT read from the flag
  // word, which the other side may have advanced already
  const unsigned long long t0 = realtime_ticks();
  for (int i = 1; i <= iters; ++i) {
    if (initiator) sys_store_i32(&peer->probe[v.rank], base + i);
    peer_wait(&me->probe[other], base + i, me, v.limit_ticks, 4, other);
    if (!initiator) sys_store_i32(&peer->probe[v.rank], base + i);
  }
  *ticks_out = realtime_ticks() - t0;
}

int32_t peer_latency(PeerComm* c, int other, int iters, int initiator, hipStream_t s, double* one_way_us) {
  MPPO_REQUIRE(c && c->connected && one_way_us && other >= 0 && other < c->world && other != c->rank && iters >= 1, "peer_latency: bad argument");
  unsigned long long* dev = nullptr;
  MPPO_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&dev), sizeof(unsigned long long)));
  hipLaunchKernelGGL(peer_probe_kernel, dim3(1), dim3(64), 0, s, c->view, other, iters, initiator, c->probe_count[other], dev);
  c->probe_count[other] += iters;
  hipError_t he = hipGetLastError();
  if (he == hipSuccess) he = hipStreamSynchronize(s);
  unsigned long long ticks = 0;
  if (he == hipSuccess) he = hipMemcpy(&ticks, dev, sizeof(ticks), hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  if (he != hipSuccess) return fail(MPPO_EHIP, "peer_latency: %s", hipGetErrorString(he));
  *one_way_us = (double)ticks / 100.0 / (double)iters / 2.0;  // 100 MHz ticks, two one-way trips per iteration
  return MPPO_OK;
}

unsigned long long peer_set_limit_ms(PeerComm* c, double ms) {
  const unsigned long long old_ms = c->view.limit_ticks / 100000ull;
  if (ms > 0.0) c->view.limit_ticks = (unsigned long long)(ms * 1e5);
  return old_ms;
}

int32_t peer_status(const PeerComm* c, int32_t* timed_out, int32_t* info8) {
  MPPO_REQUIRE(c && timed_out, "peer_status: null argument");
  MPPO_CHECK_HIP(hipDeviceSynchronize());
  PeerHdr h;
  MPPO_CHECK_HIP(hipMemcpy(&h, c->mine, sizeof(h), hipMemcpyDeviceToHost));
  *timed_out = h.error;
  if (const char* dbg = getenv("MPPO_PEER_DEBUG"); dbg && dbg[0] == '1') {  // the whole header, for post-mortems of a wait that gave up
    fprintf(stderr, "[peer rank %d] error %d info %d %d %d %d epoch %d %d nA %d K %d\n  arrive:", c->rank, h.error, h.error_info[0], h.error_info[1], h.error_info[2],
            h.error_info[3], h.epoch[0], h.epoch[1], c->view.nA, c->view.K);
    for (int q = 0; q <= kPeerMaxRanks; ++q) fprintf(stderr, " %d", h.arrive[q][0]);
    fprintf(stderr, "\n  wg_done:");
    for (int q = 0; q < c->world; ++q) fprintf(stderr, " %d", h.wg_done[q][0]);
    fprintf(stderr, "\n  adv_done:");
    for (int q = 0; q < c->world; ++q) fprintf(stderr, " %d", h.adv_done[q][0]);
    fprintf(stderr, "\n  red_done:");
    for (int k = 0; k < c->world * c->view.nA; ++k) fprintf(stderr, " %d", h.red_done[k]);
    fprintf(stderr, "\n");
  }
  if (info8) {
    for (int k = 0; k < 4; ++k) info8[k] = h.error_info[k];
    info8[4] = h.epoch[0]; info8[5] = h.epoch[1]; info8[6] = 0; info8[7] = c->view.nA;
    for (int q = 0; q <= kPeerMaxRanks; ++q) info8[6] += h.arrive[q][0];  // workgroups counted into an unfinished slice (0 between launches)
  }
  return MPPO_OK;
}

}  // namespace mppo
