// TEST INFRASTRUCTURE — CPU stand-ins for the pieces of the engine that call vendor device
// libraries (rocPRIM sort, RCCL): same ABI, same results (the radix sort is stable, as is this one; the
// all-reduce is a sum over rank processes through POSIX shared memory).
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <ctime>
#include <numeric>
#include <vector>

#include "mppo_common.h"
#include "platform.h"
#include "ppo_layout.h"

// ---- the engine's own "device" allocations: shared memory segments that other rank processes can map (hip_runtime.h) ----
#include <map>
#include <string>
namespace {
struct ShmAlloc { std::string name; size_t bytes; bool owner; };
std::map<void*, ShmAlloc>& shm_allocs() { static std::map<void*, ShmAlloc> m; return m; }
void* shm_map(const char* name, size_t bytes, bool create) {
  const int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
  if (fd < 0) return nullptr;
  if (create && ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name); return nullptr; }
  void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  return m == MAP_FAILED ? nullptr : m;
}
}  // namespace
hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) {
  static std::atomic<unsigned> serial{0};
  char name[64];
  snprintf(name, sizeof(name), "/mppo_emu_x_%d_%u", (int)getpid(), serial.fetch_add(1));
  void* m = shm_map(name, n, true);
  if (!m) return hipErrorInvalidValue;
  shm_allocs()[m] = ShmAlloc{name, n, true};
  *p = m;
  return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) { return hipExtMallocWithFlags(p, n, 0); }
hipError_t hipFree(void* p) {
  auto it = shm_allocs().find(p);
  if (it == shm_allocs().end()) return hipErrorInvalidValue;
  munmap(p, it->second.bytes);
  if (it->second.owner) shm_unlink(it->second.name.c_str());
  shm_allocs().erase(it);
  return hipSuccess;
}
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) {
  auto it = shm_allocs().find(p);
  if (it == shm_allocs().end() || !it->second.owner) return hipErrorInvalidValue;
  memset(h, 0, sizeof(*h));
  snprintf(h->reserved, 48, "%s", it->second.name.c_str());
  const unsigned long long n = it->second.bytes;
  memcpy(h->reserved + 48, &n, 8);
  return hipSuccess;
}
hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) {
  h.reserved[47] = 0;
  unsigned long long n = 0;
  memcpy(&n, h.reserved + 48, 8);
  void* m = shm_map(h.reserved, (size_t)n, false);
  if (!m) return hipErrorInvalidValue;
  shm_allocs()[m] = ShmAlloc{h.reserved, (size_t)n, false};
  *p = m;
  return hipSuccess;
}
hipError_t hipIpcCloseMemHandle(void* p) { return hipFree(p); }

namespace mppo {
// RCCL's place is taken by an all-reduce through POSIX shared memory between the rank PROCESSES of a CPU test
// (tests/test_distributed.py): same call sites in engine.hip, same semantics (in-place sum, every rank gets the
// identical result: slots are added in rank order).  The 128-byte "unique id" carries the segment name.
struct ShmHdr {
  std::atomic<int> ready, arrive, gen;
  int world;
  size_t cap;
};
struct Comm {
  int rank, world;
  ShmHdr* hdr;
  unsigned char* slots;
  size_t map_bytes;
  char name[96];
};
constexpr size_t kSlotBytes = 4u << 20;  // per rank: gradients of up to 1 M parameters

int32_t comm_unique_id(void* id128) {
  static std::atomic<unsigned> serial{0};
  memset(id128, 0, 128);
  snprintf(static_cast<char*>(id128), 96, "/mppo_emu_%d_%lld_%u", (int)getpid(), (long long)time(nullptr), serial.fetch_add(1));
  return MPPO_OK;
}

static int32_t shm_barrier(Comm* c) {
  ShmHdr* h = c->hdr;
  const int g = h->gen.load();
  if (h->arrive.fetch_add(1) + 1 == c->world) {
    h->arrive.store(0);
    h->gen.fetch_add(1);
    return MPPO_OK;
  }
  const time_t t0 = time(nullptr);
  while (h->gen.load() == g) {
    sched_yield();
    if (time(nullptr) - t0 > 300) return fail(MPPO_ENCCL, "emulator all-reduce: a peer rank did not arrive within 300 s");
  }
  return MPPO_OK;
}

int32_t comm_create(const void* id128, int rank, int world, Comm** out) {
  MPPO_REQUIRE(world >= 1 && world <= 64 && rank >= 0 && rank < world, "emulator communicator: bad rank %d / world %d", rank, world);
  Comm* c = new Comm();
  c->rank = rank; c->world = world;
  memcpy(c->name, id128, 96);
  c->name[95] = 0;
  c->map_bytes = 4096 + (size_t)world * kSlotBytes;
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return fail(MPPO_ENCCL, "emulator communicator: cannot create %s", (const char*)id128); }
  } else {
    const time_t t0 = time(nullptr);
    for (;;) {
      fd = shm_open(c->name, O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= c->map_bytes) break;
      if (fd >= 0) close(fd);
      if (time(nullptr) - t0 > 120) { delete c; return fail(MPPO_ENCCL, "emulator communicator: %s never appeared", (const char*)id128); }
      usleep(2000);
    }
  }
  void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { delete c; return fail(MPPO_ENCCL, "emulator communicator: mmap failed"); }
  c->hdr = static_cast<ShmHdr*>(m);
  c->slots = static_cast<unsigned char*>(m) + 4096;
  if (rank == 0) {
    c->hdr->arrive.store(0); c->hdr->gen.store(0); c->hdr->world = world; c->hdr->cap = kSlotBytes;
    c->hdr->ready.store(1);
  } else {
    const time_t t0 = time(nullptr);
    while (c->hdr->ready.load() != 1) {
      usleep(1000);
      if (time(nullptr) - t0 > 120) { munmap(m, c->map_bytes); delete c; return fail(MPPO_ENCCL, "emulator communicator: rank 0 never initialised the segment"); }
    }
  }
  const int32_t r = shm_barrier(c);  // everyone has the segment mapped: rank 0 may unlink the name
  if (rank == 0) shm_unlink(c->name);
  if (r != MPPO_OK) { munmap(m, c->map_bytes); delete c; return r; }
  *out = c;
  return MPPO_OK;
}

void comm_destroy(Comm* c) {
  if (!c) return;
  munmap(c->hdr, c->map_bytes);
  delete c;
}

template <typename T>
static int32_t shm_allreduce(Comm* c, T* buf, size_t n) {
  MPPO_REQUIRE(c, "all-reduce without a communicator");
  MPPO_REQUIRE(n * sizeof(T) <= kSlotBytes, "emulator all-reduce: %zu bytes exceed the slot size", n * sizeof(T));
  memcpy(c->slots + (size_t)c->rank * kSlotBytes, buf, n * sizeof(T));
  MPPO_TRY(shm_barrier(c));
  for (size_t i = 0; i < n; ++i) {
    T s = reinterpret_cast<const T*>(c->slots)[i];
    for (int r = 1; r < c->world; ++r) s += reinterpret_cast<const T*>(c->slots + (size_t)r * kSlotBytes)[i];
    buf[i] = s;
  }
  return shm_barrier(c);  // nobody overwrites a slot that a peer is still reading
}
int32_t comm_allreduce_f32(Comm* c, float* buf, size_t n, hipStream_t) { return shm_allreduce(c, buf, n); }
int32_t comm_allreduce_f64(Comm* c, double* buf, size_t n, hipStream_t) { return shm_allreduce(c, buf, n); }

// no hipGraph on the CPU: updates run eagerly
struct GraphExec {};
int32_t graph_begin(hipStream_t, bool) { return fail(MPPO_EHIP, "hipGraph is not available in the emulator build"); }
int32_t graph_end(hipStream_t, GraphExec**) { return fail(MPPO_EHIP, "hipGraph is not available in the emulator build"); }
int32_t graph_launch(GraphExec*, hipStream_t) { return fail(MPPO_EHIP, "hipGraph is not available in the emulator build"); }
void graph_destroy(GraphExec*) {}
}  // namespace mppo


