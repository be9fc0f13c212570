// peer_probe.hip — prototype and hardware check of the engine's peer-to-peer gradient all-reduce (csrc/k_peer.hip) between G
// PROCESSES that share ONE MI355X (RCCL refuses two ranks on one device; hipIpc mappings do not care).
//
// Per optimizer step every rank (process) runs, from a hipGraph:
//   K1  "publish" (stands where wgrad_kernel stands): 256 workgroups write the rank's [P] gradient into its exchange buffer with
//       system-scope write-through stores; the LAST workgroup to finish (local arrival counter) stores the step's epoch into the
//       wg_done[rank] word of every peer.
//   K2  "reduce + apply" (stands where adam_kernel stands): phase A - the first nA workgroups wait for every peer's wg_done, pull
//       slice `rank` of every rank's gradient through the hipIpc mappings, add in rank order, and push the reduced slice (+ the slice's
//       sums of squares) into EVERY rank's `red` buffer, then raise red_done[rank * nA + b] there; phase B - every workgroup waits
//       for all G * nA red_done words and reads the complete reduced gradient from its own `red` buffer (the probe checks every word
//       against the closed form instead of applying Adam).
// Values are small integers, so sums are exact and every word can be checked.  The chain is timed against the same two kernels
// with the exchange switched off (mode "local"): the difference is what the exchange costs per step on one device.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/peer_probe.bin tools/peer_probe.hip
//   HSA_ENABLE_IPC_MODE_LEGACY=0 tools/peer_probe.bin <ranks 1..8> <alloc: 0 hipMalloc | 1 fine-grained | 2 uncached> [steps per graph] [replays] [poll mode 0|1|2] [sleep units of 32 x 64 clocks] [delay us: every 8th step one rank is that late]
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[rank %d] HIP error '%s' at line %d: %s\n", g_rank, hipGetErrorString(e_), __LINE__, #x); fflush(stdout); _exit(3); } } while (0)

static int g_rank = -1;
constexpr int kMaxRanks = 8;
constexpr int kSlots = 512;

struct Shared {  // host-side rendezvous between the processes (anonymous shared mapping created before fork)
  std::atomic<int> arrive[8];
  hipIpcMemHandle_t handle[kMaxRanks];
  double us_per_step[kMaxRanks][2];
  unsigned errors[kMaxRanks];
  int failed[kMaxRanks];
};

static void host_barrier(Shared* sh, int which, int world) {
  sh->arrive[which].fetch_add(1);
  const time_t t0 = time(nullptr);
  while (sh->arrive[which].load() < world) {
    usleep(200);
    if (time(nullptr) - t0 > 120) { printf("[rank %d] host barrier %d timed out\n", g_rank, which); fflush(stdout); _exit(4); }
  }
}

// ---- exchange buffer of one rank -------------------------------------------------------------------------------------------------
struct Hdr {
  int wg_done[kMaxRanks][16];   // [q][0]: epoch of rank q's last complete K1 (one 64-byte line per writer)
  int red_done[kSlots];         // [q * nA + b]: epoch of the reduced piece b of slice q
  int arrive;                   // local fan-in of K1's workgroups (monotonic)
  int error;                    // a spin ran into its time limit
  int pad[14];
  int ready[32];                // poll mode 1: epoch up to which the collector workgroup has seen every piece (own line)
  int red_count[32];            // poll mode 2: pieces that have arrived in this rank's `red` (remote atomic adds; own line)
};
struct Peers {
  unsigned char* base[kMaxRanks];  // exchange buffers as mapped in THIS process (own: the allocation itself)
  int rank, world, nA, P4, S4;     // P4 float4 in the gradient, S4 float4 per slice
  int poll_mode, sleep_arg;        // 0 every workgroup polls every flag | 1 one collector workgroup + a ready word | 2 arrival counter (remote atomics)
  size_t pub_off, red_off;         // byte offsets of pub [4 P4] and red [4 P4 + kSlots]
};

typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef int i32x4n __attribute__((ext_vector_type(4)));
__device__ void raw_store_f32x4(f32x4n data, i32x4n rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
__device__ f32x4n raw_load_f32x4(i32x4n rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ __forceinline__ i32x4n make_rsrc(const void* base) {
  union { i32x4n v; struct { const void* p; unsigned n; unsigned f; } s; } u;
  u.s.p = base; u.s.n = 0x7FFFFFFFu; u.s.f = 0x00020000;
  return u.v;
}
constexpr int kSys = 17;  // sc0 | sc1: system scope (write-through store / cache-bypassing load)
__device__ __forceinline__ void sys_store4(void* base, size_t byte_off, float4 v) { raw_store_f32x4(f32x4n{v.x, v.y, v.z, v.w}, make_rsrc(base), (int)byte_off, 0, kSys); }
__device__ __forceinline__ float4 sys_load4(const void* base, size_t byte_off) { const f32x4n q = raw_load_f32x4(make_rsrc(base), (int)byte_off, 0, kSys); return make_float4(q.x, q.y, q.z, q.w); }
__device__ __forceinline__ int sys_load_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void sys_store_i32(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// waits until *flag has reached `epoch` (wrap-safe); gives up after ~2 s and records it (every wave leaves the kernel)
__device__ __forceinline__ bool spin_until(const int* flag, int epoch, int* error, int sleep_arg = 2) {
  unsigned long long t0 = 0;
  unsigned spins = 0;
  while ((int)(sys_load_i32(flag) - epoch) < 0) {
    if (sys_load_i32(error)) return false;
    for (int k = 0; k < sleep_arg; ++k) __builtin_amdgcn_s_sleep(32);
    if ((++spins & 255u) == 0) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t0 == 0) t0 = now;
      if (now - t0 > 200000000ull) { sys_store_i32(error, 1); return false; }  // 100 MHz: 2 s
    }
  }
  return true;
}

__device__ __forceinline__ float grad_value(int rank, int epoch, int i) { return (float)(((i * 7 + epoch * 3) & 63) + rank); }

__global__ void __launch_bounds__(512) publish_kernel(Peers p, const int* epoch_base, int step, int exchange) {
  const int epoch = epoch_base[0] + step + 1;
  unsigned char* me = p.base[p.rank];
  Hdr* hdr = reinterpret_cast<Hdr*>(me);
  const int nthr = gridDim.x * blockDim.x;
  for (int i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < p.P4; i4 += nthr)
    sys_store4(me + p.pub_off, (size_t)i4 * 16, make_float4(grad_value(p.rank, epoch, 4 * i4), grad_value(p.rank, epoch, 4 * i4 + 1), grad_value(p.rank, epoch, 4 * i4 + 2),
                                                                grad_value(p.rank, epoch, 4 * i4 + 3)));
  if (!exchange) return;
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(&hdr->arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == epoch * (int)gridDim.x)  // the last workgroup of this step's launch: every workgroup's stores are acknowledged
#pragma unroll
      for (int q = 0; q < kMaxRanks; ++q)
        if (q < p.world && q != p.rank) sys_store_i32(&reinterpret_cast<Hdr*>(p.base[q])->wg_done[p.rank][0], epoch);
  }
}

__global__ void __launch_bounds__(256) reduce_apply_kernel(Peers p, const int* epoch_base, int step, int exchange, unsigned* errors) {
  __shared__ float s_red[4];
  const int epoch = epoch_base[0] + step + 1;
  unsigned char* me = p.base[p.rank];
  Hdr* hdr = reinterpret_cast<Hdr*>(me);
  const int t = threadIdx.x, b = blockIdx.x;
  if (exchange && b < p.nA) {
    // ---- phase A: piece b of slice `rank` ----
    if (t < p.world && t != p.rank) spin_until(&hdr->wg_done[t][0], epoch, &hdr->error, p.sleep_arg);
    __syncthreads();
    const int i4 = p.rank * p.S4 + b * 256 + t;
    const bool on = b * 256 + t < p.S4 && i4 < p.P4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) {
      float4 g[kMaxRanks];
#pragma unroll
      for (int q = 0; q < kMaxRanks; ++q)
        if (q < p.world) g[q] = sys_load4(p.base[q] + p.pub_off, (size_t)i4 * 16);
#pragma unroll
      for (int q = 0; q < kMaxRanks; ++q)
        if (q < p.world) { acc.x += g[q].x; acc.y += g[q].y; acc.z += g[q].z; acc.w += g[q].w; }
#pragma unroll
      for (int q = 0; q < kMaxRanks; ++q)
        if (q < p.world) sys_store4(p.base[q] + p.red_off, (size_t)i4 * 16, acc);
    }
    float sq = acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
    for (int m = 1; m < 64; m <<= 1) sq += __shfl_xor(sq, m);
    if ((t & 63) == 0) s_red[t >> 6] = sq;
    __syncthreads();
    if (t == 0) {
      const float s = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
#pragma unroll
      for (int q = 0; q < kMaxRanks; ++q)
        if (q < p.world) __hip_atomic_store(reinterpret_cast<float*>(p.base[q] + p.red_off) + (size_t)4 * p.P4 + p.rank * p.nA + b, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    drain_stores();
    __syncthreads();
    if (t == 0) {
#pragma unroll
      for (int q = 0; q < kMaxRanks; ++q)
        if (q < p.world) {
          if (p.poll_mode == 2) __hip_atomic_fetch_add(&reinterpret_cast<Hdr*>(p.base[q])->red_count[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          else sys_store_i32(&reinterpret_cast<Hdr*>(p.base[q])->red_done[p.rank * p.nA + b], epoch);
        }
    }
  }
  // ---- phase B: the whole reduced gradient is in my own `red` buffer once every piece's flag has arrived ----
  if (exchange) {
    if (p.poll_mode == 0) {
      if (t < 64)
        for (int s = t; s < p.world * p.nA; s += 64) spin_until(&hdr->red_done[s], epoch, &hdr->error, p.sleep_arg);
    } else if (p.poll_mode == 1) {
      if (b == (int)gridDim.x - 1) {  // the collector: the only workgroup that reads the flags
        if (t < 64)
          for (int s = t; s < p.world * p.nA; s += 64) spin_until(&hdr->red_done[s], epoch, &hdr->error, p.sleep_arg);
        __syncthreads();
        if (t == 0) sys_store_i32(&hdr->ready[0], epoch);
      } else if (t == 0) {
        spin_until(&hdr->ready[0], epoch, &hdr->error, p.sleep_arg);
      }
    } else {
      if (t == 0) spin_until(&hdr->red_count[0], epoch * p.world * p.nA, &hdr->error, p.sleep_arg);
    }
    __syncthreads();
  }
  unsigned bad = 0;
  const int nthr = gridDim.x * blockDim.x;
  for (int i4 = b * blockDim.x + t; i4 < p.P4; i4 += nthr) {
    const float4 v = exchange ? sys_load4(me + p.red_off, (size_t)i4 * 16) : sys_load4(me + p.pub_off, (size_t)i4 * 16);
    float want[4];
    for (int c = 0; c < 4; ++c) {
      float s = 0.f;
      if (exchange) for (int q = 0; q < p.world; ++q) s += grad_value(q, epoch, 4 * i4 + c);
      else s = grad_value(p.rank, epoch, 4 * i4 + c);
      want[c] = s;
    }
    bad += (v.x != want[0]) + (v.y != want[1]) + (v.z != want[2]) + (v.w != want[3]);
  }
  if (exchange && b == 0 && t < 64) {  // the sums of squares of the pieces: every slot must hold its piece's value (checked loosely: > 0)
    for (int s = t; s < p.world * p.nA; s += 64) {
      const float v = __hip_atomic_load(reinterpret_cast<const float*>(me + p.red_off) + (size_t)4 * p.P4 + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      bad += !(v > 0.f);
    }
  }
  if (bad) atomicAdd(errors, bad);
}

__global__ void delay_kernel(unsigned long long ticks) {  // one rank is late: the others wait that long
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

__global__ void advance_kernel(int* epoch_base, int steps) { if (threadIdx.x == 0 && blockIdx.x == 0) epoch_base[0] += steps; }

static int g_poll_mode = 0, g_sleep_arg = 1, g_delay_us = 0;
static int run_rank(Shared* sh, int rank, int world, int alloc_mode, int steps, int replays) {
  g_rank = rank;
  CK(hipSetDevice(0));
  const int P = 250140, P4 = P / 4;
  const int S4 = (P4 + world - 1) / world, nA = (S4 + 255) / 256;
  const size_t pub_off = sizeof(Hdr), red_off = pub_off + (size_t)P4 * 16, bytes = red_off + (size_t)P4 * 16 + kSlots * 4;
  unsigned char* mine = nullptr;
  if (alloc_mode == 0) CK(hipMalloc(reinterpret_cast<void**>(&mine), bytes));
  else CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&mine), bytes, alloc_mode == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
  CK(hipMemset(mine, 0, bytes));
  CK(hipDeviceSynchronize());
  if (world > 1) CK(hipIpcGetMemHandle(&sh->handle[rank], mine));
  host_barrier(sh, 0, world);
  Peers p{};
  p.rank = rank; p.world = world; p.nA = nA; p.P4 = P4; p.S4 = S4; p.pub_off = pub_off; p.red_off = red_off;
  p.poll_mode = g_poll_mode; p.sleep_arg = g_sleep_arg;
  for (int q = 0; q < world; ++q) {
    if (q == rank) { p.base[q] = mine; continue; }
    void* ptr = nullptr;
    CK(hipIpcOpenMemHandle(&ptr, sh->handle[q], hipIpcMemLazyEnablePeerAccess));
    p.base[q] = static_cast<unsigned char*>(ptr);
  }
  host_barrier(sh, 1, world);
  if (rank == 0) { printf("ranks %d, alloc mode %d: exchange buffers mapped (%zu bytes each, nA = %d pieces per slice)\n", world, alloc_mode, bytes, nA); fflush(stdout); }
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int* epoch_base; unsigned* errors;
  CK(hipMalloc(&epoch_base, 4)); CK(hipMalloc(&errors, 4));
  CK(hipMemset(epoch_base, 0, 4)); CK(hipMemset(errors, 0, 4));
  CK(hipDeviceSynchronize());
  const int grid2 = 245;
  for (int exchange = 1; exchange >= 0; --exchange) {
    if (world == 1 && exchange) { sh->us_per_step[rank][1] = 0; continue; }
    hipGraph_t g; hipGraphExec_t x;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int st = 0; st < steps; ++st) {
      if (g_delay_us > 0 && (st % 8) == 0 && ((st / 8) % world) == rank) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s, (unsigned long long)g_delay_us * 100ull);
      hipLaunchKernelGGL(publish_kernel, dim3(256), dim3(512), 0, s, p, epoch_base, st, exchange);
      hipLaunchKernelGGL(reduce_apply_kernel, dim3(grid2), dim3(256), 0, s, p, epoch_base, st, exchange, errors);
    }
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, s, epoch_base, steps);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
    host_barrier(sh, 2 + 2 * (1 - exchange), world);
    CK(hipGraphLaunch(x, s));  // warm-up
    CK(hipStreamSynchronize(s));
    host_barrier(sh, 3 + 2 * (1 - exchange), world);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < replays; ++r) CK(hipGraphLaunch(x, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    sh->us_per_step[rank][exchange] = (double)ms * 1e3 / ((double)steps * replays);
    CK(hipGraphExecDestroy(x)); CK(hipGraphDestroy(g));
  }
  unsigned herr = 0;
  CK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost));
  Hdr hh;
  CK(hipMemcpy(&hh, mine, sizeof(Hdr), hipMemcpyDeviceToHost));
  sh->errors[rank] = herr;
  sh->failed[rank] = hh.error;
  host_barrier(sh, 6, world);
  for (int q = 0; q < world; ++q) if (q != rank) CK(hipIpcCloseMemHandle(p.base[q]));
  host_barrier(sh, 7, world);
  CK(hipFree(mine));
  return 0;
}

int main(int argc, char** argv) {
  const int world = argc > 1 ? atoi(argv[1]) : 2, alloc_mode = argc > 2 ? atoi(argv[2]) : 1;
  const int steps = argc > 3 ? atoi(argv[3]) : 128, replays = argc > 4 ? atoi(argv[4]) : 10;
  g_poll_mode = argc > 5 ? atoi(argv[5]) : 0; g_sleep_arg = argc > 6 ? atoi(argv[6]) : 1; g_delay_us = argc > 7 ? atoi(argv[7]) : 0;
  if (world < 1 || world > kMaxRanks) { printf("ranks 1..8\n"); return 2; }
  Shared* sh = static_cast<Shared*>(mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0));
  if (sh == MAP_FAILED) { printf("mmap failed\n"); return 2; }
  memset(sh, 0, sizeof(Shared));
  pid_t pids[kMaxRanks];
  for (int r = 0; r < world; ++r) {  // fork BEFORE anything touches the GPU: every child initialises HIP for itself
    pids[r] = fork();
    if (pids[r] == 0) _exit(run_rank(sh, r, world, alloc_mode, steps, replays));
  }
  int rc = 0;
  for (int r = 0; r < world; ++r) {
    int status = 0;
    waitpid(pids[r], &status, 0);
    if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) rc = 1;
  }
  unsigned errs = 0; int failed = 0;
  double ex = 0, lo = 0;
  for (int r = 0; r < world; ++r) { errs += sh->errors[r]; failed += sh->failed[r]; ex = ex > sh->us_per_step[r][1] ? ex : sh->us_per_step[r][1]; lo = lo > sh->us_per_step[r][0] ? lo : sh->us_per_step[r][0]; }
  printf("poll mode %d sleep %d delay %d us | ", g_poll_mode, g_sleep_arg, g_delay_us);
  printf("ranks %d alloc %d steps/graph %d replays %d: exit %s, wrong words %u, spin time-outs %d | us per step (max over ranks): exchange %.2f, local-only %.2f, difference %.2f\n",
         world, alloc_mode, steps, replays, rc ? "FAILED" : "ok", errs, failed, ex, lo, ex - lo);
  return rc || errs || failed;
}
