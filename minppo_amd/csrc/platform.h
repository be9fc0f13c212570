// platform.h — the two runtime services the engine needs besides kernel launches:
// RCCL collectives over xGMI and hipGraph capture / replay.  (platform.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace mppo {

struct Comm;
int32_t comm_unique_id(void* id128);
int32_t comm_create(const void* id128, int rank, int world, Comm** out);
void comm_destroy(Comm* c);
int32_t comm_allreduce_f32(Comm* c, float* buf, size_t n, hipStream_t s);   // in-place sum
int32_t comm_allreduce_f64(Comm* c, double* buf, size_t n, hipStream_t s);  // in-place sum

struct GraphExec;
int32_t graph_begin(hipStream_t s, bool with_collectives);  // with_collectives: relaxed capture mode (RCCL calls runtime APIs while enqueueing)
int32_t graph_end(hipStream_t s, GraphExec** out);
int32_t graph_launch(GraphExec* g, hipStream_t s);
void graph_destroy(GraphExec* g);

}  // namespace mppo
