// TEST INFRASTRUCTURE — fiber scheduler of the CPU SIMT emulator (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>

namespace emu {
dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
unsigned char* g_dyn_smem = nullptr;
double g_xchg[4096];

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
  std::vector<unsigned char> smem(shmem + 16);
  std::vector<Fiber> fibers(nthreads);
  for (auto& f : fibers) f.stack.resize(kStack);
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_blockIdx = dim3(bx, by, bz);
        memset(smem.data(), 0xFF, smem.size());  // LDS starts as NaNs: uninitialised reads show up
        g_dyn_smem = smem.data();
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
      }
  g_body = nullptr;
}
}  // namespace emu
