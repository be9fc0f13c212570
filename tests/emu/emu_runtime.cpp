// TEST INFRASTRUCTURE — fiber scheduler of the CPU SIMT emulator (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

namespace emu {
dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
unsigned char* g_dyn_smem = nullptr;
double g_xchg[4096];
const void* g_kernel_ptr = nullptr;

static const char* kernel_name() {
  Dl_info info;
  return (g_kernel_ptr && dladdr(g_kernel_ptr, &info) && info.dli_sname) ? info.dli_sname : "?";
}

namespace {
struct Fiber {
  ucontext_t ctx;
  std::vector<char> stack;
  bool done = false;
  dim3 tidx;
};
ucontext_t g_sched;
Fiber* g_cur = nullptr;
const std::function<void()>* g_body = nullptr;
constexpr size_t kStack = 256 * 1024;

void trampoline() {
  (*g_body)();
  g_cur->done = true;
  swapcontext(&g_cur->ctx, &g_sched);
}
}  // namespace

void barrier() { swapcontext(&g_cur->ctx, &g_sched); }

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  const size_t nthreads = (size_t)block.x * block.y * block.z;
  if (nthreads == 0 || nthreads > 1024) { fprintf(stderr, "emu: bad block size\n"); abort(); }
  g_gridDim = grid;
  g_blockDim = block;
  g_body = &body;
  // guard zones around the dynamic LDS block and a canary at the low end of every fiber stack: an out-of-range LDS access
  // (silently dropped by the hardware!) or a stack overflow aborts with a message instead of corrupting the heap
  constexpr size_t kGuard = 4096;
  std::vector<unsigned char> smem_store(shmem + 16 + 2 * kGuard);
  unsigned char* const smem_lo = smem_store.data();
  unsigned char* const smem_base = smem_lo + kGuard;
  unsigned char* const smem_hi = smem_base + shmem + 16;
  std::vector<Fiber> fibers(nthreads);
  for (auto& f : fibers) { f.stack.resize(kStack); memset(f.stack.data(), 0xA5, 256); }
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_blockIdx = dim3(bx, by, bz);
        memset(smem_base, 0xFF, shmem + 16);  // LDS starts as NaNs: uninitialised reads show up
        memset(smem_lo, 0x5C, kGuard);
        memset(smem_hi, 0x5C, kGuard);
        g_dyn_smem = smem_base;
        size_t t = 0;
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx, ++t) {
              Fiber& f = fibers[t];
              f.done = false;
              f.tidx = dim3(tx, ty, tz);
              getcontext(&f.ctx);
              f.ctx.uc_stack.ss_sp = f.stack.data();
              f.ctx.uc_stack.ss_size = kStack;
              f.ctx.uc_link = &g_sched;
              makecontext(&f.ctx, trampoline, 0);
            }
        size_t live = nthreads;
        while (live) {
          size_t arrived = 0, finished = 0;
          for (auto& f : fibers) {
            if (f.done) continue;
            g_cur = &f;
            g_threadIdx = f.tidx;
            swapcontext(&g_sched, &f.ctx);
            if (f.done) ++finished; else ++arrived;
          }
          live -= finished;
          // (threads that return early while others wait at a barrier are tolerated, as on the hardware)
        }
        for (size_t i = 0; i < kGuard; ++i)
          if (smem_lo[i] != 0x5C || smem_hi[i] != 0x5C) {
            fprintf(stderr, "emu: %s block (%u,%u,%u) wrote outside its %zu bytes of dynamic LDS (guard byte %zu %s the block)\n", kernel_name(), bx, by, bz,
                    shmem, i, smem_lo[i] != 0x5C ? "below" : "above");
            abort();
          }
        for (auto& f : fibers)
          for (int i = 0; i < 256; ++i)
            if ((unsigned char)f.stack[i] != 0xA5) { fprintf(stderr, "emu: %s: fiber stack overflow (%zu bytes are not enough)\n", kernel_name(), kStack); abort(); }
      }
  g_body = nullptr;
}
}  // namespace emu
