// Multi-rank MSM at the C level: one process per GPU, the records of partial window sums exchanged with RCCL.
//
//   hipcc -O2 -Iinclude examples/msm_rccl.cpp -Ljubjub_amd/lib -ljubjub_hip -L/opt/rocm/lib -lrccl \
//         -Wl,-rpath,$PWD/jubjub_amd/lib -Wl,-rpath,/opt/rocm/lib -o examples/msm_rccl
//   # one process per GPU; the ncclUniqueId travels through a file (any rendezvous will do: MPI, a socket, torch.distributed ...)
//   for r in 0 1 2 3 4 5 6 7; do RANK=$r WORLD_SIZE=8 LOCAL_RANK=$r JJ_ID_FILE=/tmp/jj_id ./examples/msm_rccl 1048576 & done; wait
//
// What it computes: sum_i points[i] * scalars[i] over n terms (the reference's `iter.map(|(p, k)| p * k).sum()`,
// /root/reference/src/lib.rs:183-193 + 873-879), the terms cut into contiguous shards, one per rank.  Three ways, which must agree:
//   A  jj_ctx_set_comm + jj_msm_allgather                      the whole exchange behind one call
//   B  jj_msm_partial -> ncclAllGather -> one D2H -> jj_msm_combine   the same steps spelled out (what A does inside)
//   C  jj_msm_allgather with the WINDOW partition              every rank holds all terms and reduces windows g, g + G, ...
//   D  jj_msm_allgather_begin / jj_msm_finish, three in flight    a stream of MSMs: gather, fold and host tail of one beside the kernels of the next
// Inputs are the library's counter-based generators over GLOBAL term indices (jj_synth_scalars / jj_random_points), so every rank
// builds its shard without moving data and a checker can rebuild the batch (tests/test_gpu_dist.py compares rank 0's line with the oracle).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <time.h>
#include <unistd.h>
#include <vector>

#include "jubjub_hip.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s: %s\n", rank, #x, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCL(x) do { ncclResult_t e_ = (x); if (e_ != ncclSuccess) { fprintf(stderr, "rank %d: %s: %s\n", rank, #x, ncclGetErrorString(e_)); return 1; } } while (0)
#define JJ(x) do { int e_ = (x); if (e_ != JJ_OK) { fprintf(stderr, "rank %d: %s: %d (%s)\n", rank, #x, e_, jj_last_error(ctx)); return 1; } } while (0)

static const uint64_t SEED = 0x4A55424A5542ull, POINT_SEED = SEED ^ 0x9E3779B97F4A7C15ull;   // bench.py's streams
static int env_int(const char* k, int dflt) { const char* v = getenv(k); return v ? atoi(v) : dflt; }
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

// rank 0 creates the id and publishes it with an atomic rename; the others wait for the file
static int exchange_id(ncclUniqueId* id, int rank, int world, const char* path) {
  if (rank == 0) {
    if (ncclGetUniqueId(id) != ncclSuccess) return 1;
    if (world == 1) return 0;
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(id, 1, sizeof *id, f) != sizeof *id) return 1;
    fclose(f);
    return rename(tmp.c_str(), path) != 0;
  }
  for (int tries = 0; tries < 600; tries++) {
    FILE* f = fopen(path, "rb");
    if (f) { const size_t got = fread(id, 1, sizeof *id, f); fclose(f); if (got == sizeof *id) return 0; }
    usleep(100 * 1000);
  }
  return 1;
}

int main(int argc, char** argv) {
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), device = env_int("JJ_DEVICE", env_int("LOCAL_RANK", 0));
  const size_t n = argc > 1 ? (size_t)strtoull(argv[1], nullptr, 10) : 50000;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const char* id_file = getenv("JJ_ID_FILE") ? getenv("JJ_ID_FILE") : "/tmp/jj_msm_rccl_id";
  jj_ctx* ctx = nullptr;

  HIP(hipSetDevice(device));
  ncclUniqueId id;
  if (exchange_id(&id, rank, world, id_file)) { fprintf(stderr, "rank %d: could not exchange the ncclUniqueId through %s\n", rank, id_file); return 1; }
  ncclComm_t comm;
  NCCL(ncclCommInitRank(&comm, world, id, rank));
  if (rank == 0 && world > 1) unlink(id_file);

  JJ(jj_ctx_create(device, &ctx));
  JJ(jj_ctx_set_comm(ctx, comm, rank, world, (void*)&ncclAllGather));

  // this rank's shard [lo, hi) of the n terms, generated on the device from the global indices; and (for C) the whole batch
  const size_t lo = n / world * rank + ((size_t)rank < n % world ? rank : n % world), cnt = n / world + ((size_t)rank < n % world ? 1 : 0);
  void *d_s, *d_p, *d_sall, *d_pall, *d_rec, *d_all;
  HIP(hipMalloc(&d_s, 32 * (cnt + 1))); HIP(hipMalloc(&d_p, 64 * (cnt + 1)));
  HIP(hipMalloc(&d_sall, 32 * (n + 1))); HIP(hipMalloc(&d_pall, 64 * (n + 1)));
  HIP(hipMalloc(&d_rec, JJ_MSM_PARTIAL_BYTES)); HIP(hipMalloc(&d_all, (size_t)world * JJ_MSM_PARTIAL_BYTES));
  JJ(jj_synth_scalars(ctx, cnt, SEED, lo, d_s));
  JJ(jj_random_points(ctx, cnt, POINT_SEED, lo, 0, d_p, nullptr));
  JJ(jj_synth_scalars(ctx, n, SEED, 0, d_sall));
  JJ(jj_random_points(ctx, n, POINT_SEED, 0, 0, d_pall, nullptr));

  // ---- A: the exchange behind one call (term partition)
  uint8_t a[64], b[64], c[64];
  JJ(jj_msm_allgather(ctx, cnt, d_s, d_p, 0, a));
  const double t0 = now();
  for (int r = 0; r < reps; r++) JJ(jj_msm_allgather(ctx, cnt, d_s, d_p, 0, a));
  const double per_call = (now() - t0) / (reps > 0 ? reps : 1);

  // ---- B: the same steps spelled out, on a stream of the caller
  hipStream_t s;
  HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  JJ(jj_ctx_set_stream(ctx, s));                                                        // the record is produced on s ...
  JJ(jj_msm_partial(ctx, cnt, d_s, d_p, 0, 1, d_rec));                                  // (device pointer: asynchronous)
  NCCL(ncclAllGather(d_rec, d_all, JJ_MSM_PARTIAL_BYTES, ncclUint8, comm, s));          // ... gathered on s over xGMI ...
  std::vector<uint8_t> h_all((size_t)world * JJ_MSM_PARTIAL_BYTES);
  HIP(hipMemcpyAsync(h_all.data(), d_all, h_all.size(), hipMemcpyDeviceToHost, s));     // ... and copied to the host ONCE
  HIP(hipStreamSynchronize(s));
  JJ(jj_msm_combine((size_t)world, h_all.data(), b));                                   // one host tail: window sums, Horner, one inversion
  JJ(jj_ctx_use_own_stream(ctx));

  // ---- C: window partition (every rank passes ALL terms)
  JJ(jj_msm_allgather(ctx, n, d_sall, d_pall, 1, c));

  // ---- D: a stream of MSMs (here the same one `reps` times), three jobs in flight; every rank begins the same jobs in the same order
  uint8_t d[64];
  bool ad = true;
  const double t1 = now();
  {
    std::vector<jj_msm_job*> pend;
    for (int r = 0; r < reps + 3; r++) {
      if (r < reps) { jj_msm_job* j = nullptr; JJ(jj_msm_allgather_begin(ctx, cnt, d_s, d_p, 0, &j)); pend.push_back(j); }
      if (pend.size() == 3 || (r >= reps && !pend.empty())) { JJ(jj_msm_finish(pend.front(), d)); pend.erase(pend.begin()); ad = ad && memcmp(a, d, 64) == 0; }
    }
  }
  const double per_job = (now() - t1) / (reps > 0 ? reps : 1);

  const bool ab = memcmp(a, b, 64) == 0, ac = memcmp(a, c, 64) == 0;
  if (rank == 0) {
    printf("msm_rccl: ranks=%d n=%zu result=", world, n);
    for (int i = 0; i < 64; i++) printf("%02x", a[i]);
    printf("\nmsm_rccl: allgather == partial+ncclAllGather+combine: %s; term partition == window partition: %s\n", ab ? "ok" : "MISMATCH", ac ? "ok" : "MISMATCH");
    printf("msm_rccl: %.3f ms per jj_msm_allgather (%zu terms per rank, %d ranks)\n", per_call * 1e3, cnt, world);
    printf("msm_rccl: %.3f ms per MSM with three jj_msm_allgather_begin jobs in flight: %s\n", per_job * 1e3, ad ? "ok" : "MISMATCH");
  }
  JJ(jj_ctx_set_comm(ctx, nullptr, 0, 1, nullptr));
  jj_ctx_destroy(ctx);
  (void)hipStreamDestroy(s);
  for (void* p : {d_s, d_p, d_sall, d_pall, d_rec, d_all}) (void)hipFree(p);
  NCCL(ncclCommDestroy(comm));
  return (ab && ac && ad) ? 0 : 1;
}
