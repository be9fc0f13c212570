// platform.hip — RCCL communicator (gradient / statistics all-reduce over xGMI) and hipGraph
// capture / replay of the update's launch sequence.
#include "platform.h"

#include <rccl/rccl.h>

#include <cstring>

#include "mppo_common.h"

namespace mppo {

struct Comm { ncclComm_t comm; int rank, world; };
struct GraphExec { hipGraph_t graph; hipGraphExec_t exec; };

#define MPPO_CHECK_NCCL(expr)                                                                         \
  do {                                                                                                \
    ncclResult_t _r = (expr);                                                                         \
    if (_r != ncclSuccess) return fail(MPPO_ENCCL, "%s failed: %s", #expr, ncclGetErrorString(_r));   \
  } while (0)

static_assert(sizeof(ncclUniqueId) == 128, "the ABI passes the RCCL unique id as 128 bytes");

int32_t comm_unique_id(void* id128) {
  ncclUniqueId id;
  MPPO_CHECK_NCCL(ncclGetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return MPPO_OK;
}

int32_t comm_create(const void* id128, int rank, int world, Comm** out) {
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  Comm* c = new Comm{nullptr, rank, world};
  ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { delete c; return fail(MPPO_ENCCL, "ncclCommInitRank failed: %s", ncclGetErrorString(r)); }
  *out = c;
  return MPPO_OK;
}

void comm_destroy(Comm* c) {
  if (!c) return;
  ncclCommDestroy(c->comm);
  delete c;
}

int32_t comm_allreduce_f32(Comm* c, float* buf, size_t n, hipStream_t s) {
  MPPO_REQUIRE(c, "all-reduce without a communicator");
  MPPO_CHECK_NCCL(ncclAllReduce(buf, buf, n, ncclFloat32, ncclSum, c->comm, s));
  return MPPO_OK;
}

int32_t comm_allreduce_f64(Comm* c, double* buf, size_t n, hipStream_t s) {
  MPPO_REQUIRE(c, "all-reduce without a communicator");
  MPPO_CHECK_NCCL(ncclAllReduce(buf, buf, n, ncclFloat64, ncclSum, c->comm, s));
  return MPPO_OK;
}

int32_t graph_begin(hipStream_t s, bool with_collectives) {
  MPPO_CHECK_HIP(hipStreamBeginCapture(s, with_collectives ? hipStreamCaptureModeRelaxed : hipStreamCaptureModeThreadLocal));
  return MPPO_OK;
}

int32_t graph_end(hipStream_t s, GraphExec** out) {
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(s, &g);
  if (e != hipSuccess || !g) { (void)hipGetLastError(); return fail(MPPO_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e)); }
  hipGraphExec_t x = nullptr;
  e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(g); (void)hipGetLastError(); return fail(MPPO_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
  *out = new GraphExec{g, x};
  return MPPO_OK;
}

int32_t graph_launch(GraphExec* g, hipStream_t s) {
  MPPO_CHECK_HIP(hipGraphLaunch(g->exec, s));
  return MPPO_OK;
}

void graph_destroy(GraphExec* g) {
  if (!g) return;
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
}

}  // namespace mppo
