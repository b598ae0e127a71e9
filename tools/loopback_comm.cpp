// An all-gather that plays the OTHER ranks of an RCCL communicator on one GPU (tests/util.py LoopbackComm, experiments/misc/
// msm_allgather_pipeline.py): jj_ctx_set_comm takes the address of jj_loopback_all_gather in place of ncclAllGather and a LoopbackComm* in
// place of the ncclComm_t.  Like RCCL's it is stream-ordered and returns at once: slot `rank` of the receive buffer takes what the
// calling rank sends, the other slots the records the caller prepared for the other ranks (device memory, `count` bytes apart).  Several
// prepared sets are used in turn, one per call, so that several jj_msm_allgather_begin jobs with different terms can be in flight.
//   hipcc -O2 -shared -fPIC -o tools/libloopback_comm.so tools/loopback_comm.cpp
#include <hip/hip_runtime.h>

struct LoopbackComm {
  int rank, world;
  unsigned calls, ring;
  const void* others[64];
};

extern "C" {
__attribute__((visibility("default"))) LoopbackComm* jj_loopback_create(int rank, int world) {
  LoopbackComm* c = new LoopbackComm();
  c->rank = rank; c->world = world; c->calls = 0; c->ring = 0;
  return c;
}
// the records of all `world` ranks for the calls k, k + ring, ... (slot `rank` of the set is ignored)
__attribute__((visibility("default"))) int jj_loopback_set(LoopbackComm* c, unsigned k, const void* records_dev) {
  if (!c || k >= 64) return 1;
  c->others[k] = records_dev;
  if (k + 1 > c->ring) c->ring = k + 1;
  return 0;
}
__attribute__((visibility("default"))) void jj_loopback_rewind(LoopbackComm* c) { if (c) c->calls = 0; }
__attribute__((visibility("default"))) unsigned jj_loopback_calls(const LoopbackComm* c) { return c ? c->calls : 0; }
__attribute__((visibility("default"))) void jj_loopback_destroy(LoopbackComm* c) { delete c; }
// the signature of ncclAllGather (sendbuff, recvbuff, sendcount, datatype, comm, stream); datatype: bytes only
__attribute__((visibility("default"))) int jj_loopback_all_gather(const void* send, void* recv, size_t count, int datatype, void* comm, hipStream_t stream) {
  LoopbackComm* c = (LoopbackComm*)comm;
  if (!c || !c->ring || datatype != 1) return 4;                    // ncclInvalidArgument
  const void* others = c->others[c->calls++ % c->ring];
  if (hipMemcpyAsync(recv, others, count * (size_t)c->world, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
  if (hipMemcpyAsync((char*)recv + (size_t)c->rank * count, send, count, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
  return 0;
}
}
