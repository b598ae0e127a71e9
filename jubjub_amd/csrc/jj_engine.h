// Internal header of libjubjub_hip.so: the context, its workspaces and the helpers the library's translation units share.
//   jj_pipeline.hip   context, streams, staging of host arguments, the host-buffer pipeline's machinery, page-locked host memory
//   jj_abi.hip        the batch entry points (fields, points, ladders, fixed-base, codec, generators) and their kernels (jj_kernels.h)
//   jj_msm.hip        multi-scalar multiplication: planner, jobs, records, the multi-rank exchange; kernels in jj_msm_kernels.h
//   jj_multi.hip      jj_multi_*: several devices of one node from one process
// There is no CPU fallback: without a gfx950 device jj_ctx_create fails with JJ_ERR_NODEVICE.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <mutex>
#include <thread>
#include <condition_variable>
#include <atomic>
#include <functional>
#include <dlfcn.h>
#include <unistd.h>

#include "../../include/jubjub_hip.h"
#include "jj_kernels.h"      // device helpers and argument types; the __global__ kernels the including translation unit asked for (JJ_KERNELS_*)
#include "jj_host_tail.h"

using namespace jj;

#define JJ_VERSION 100  /* 0.1.0 */
#define JJ_API extern "C" __attribute__((visibility("default")))

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct jj_table {
  u32* dev = nullptr;      // entries x ANIELS_WORDS
  int window_bits = FB_W;  // 7: signed comb in LDS (k_fixedbase_comb: 8 teeth, the default); 6: LDS-staged window table (k_fixedbase); 8..16: table gathered from L2 / Infinity Cache (k_fixedbase_gather)
  int device = -1;         // the table lives in this device's memory: only contexts of the same device may use it
  FbParams fp;
  FbxParams fx;            // composite table (several bases with short scalars, layout of k_fixedbase): fx.nb > 0
  jj_table() { memset(&fx, 0, sizeof fx); }
};

struct WorkSet { DevBuf ext, scratch, tables, cursor; };
// A few host threads that copy between a caller's pageable array and the context's page-locked staging buffers while the GPU works on the
// neighbouring chunk (the host-buffer pipeline's bounce path).  One job at a time: copy(dst, src, bytes) cuts the range into page-aligned
// slices, the pool's threads and the caller each take slices until none is left.
class HostCopyPool {
 public:
  explicit HostCopyPool(int nthreads) {
    for (int t = 0; t < nthreads; t++) th_.emplace_back([this]() { worker(); });
  }
  ~HostCopyPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; gen_++; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void copy(void* dst, const void* src, size_t bytes) {
    if (bytes < ((size_t)4 << 20) || th_.empty()) { memcpy(dst, src, bytes); return; }
    {
      std::lock_guard<std::mutex> lk(mu_);
      dst_ = (uint8_t*)dst; src_ = (const uint8_t*)src; bytes_ = bytes;
      slice_ = std::max<size_t>((size_t)1 << 20, ((bytes / (4 * (th_.size() + 1))) + 4095) & ~(size_t)4095);
      next_.store(0); pending_ = (int)th_.size(); gen_++;
    }
    cv_.notify_all();
    run_slices();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this]() { return pending_ == 0; });
  }
 private:
  void run_slices() {
    for (;;) {
      const size_t lo = next_.fetch_add(slice_);
      if (lo >= bytes_) return;
      memcpy(dst_ + lo, src_ + lo, std::min(slice_, bytes_ - lo));
    }
  }
  void worker() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&]() { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
      }
      run_slices();
      { std::lock_guard<std::mutex> lk(mu_); if (--pending_ == 0) done_cv_.notify_one(); }
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  uint8_t* dst_ = nullptr; const uint8_t* src_ = nullptr; size_t bytes_ = 0, slice_ = 1;
  std::atomic<size_t> next_{0};
  int pending_ = 0; uint64_t gen_ = 0; bool stop_ = false;
};
struct jj_ctx;
// One MSM pipeline of a context: its own workspaces, and for lanes >= 1 its own streams.  Lane 0 runs on the context's launch
// stream (jj_msm, host-array jobs); device-pointer jobs of jj_msm_begin alternate over the lanes, so that the dependent chains at
// the end of one MSM (a few hundred wavefronts) overlap the sort and accumulation of the next -- what several contexts on one
// device give (profiles/r3_msm_concurrency.txt), without the caller having to run several.
struct MsmLane {
  hipStream_t stream = nullptr;                     // lane 0: the context's launch stream, filled in at every use; other lanes: owned
  hipEvent_t ready_ev = nullptr;
  DevBuf buf[8], ctl, bigpart, seg, rec, bins;        // ... | totals + cursors per (slot, coarse bin) of the two-pass sort, two parities (bins_parity: the half the next pass uses)
  int bins_parity = 0;             // kprime, niels, offsets, idx, buckets, heads/records, -, tile counts | counters + lists | big-bucket partials | segments | record
  bool owned = false;
};
constexpr int MSM_LANES_MAX = 4;
constexpr int MSM_SMALL_BLK_MAX = 64;          // = MSM_TREE_QUADS (jj_msm_kernels.h): workgroups per window whose partial sums the window's last workgroup folds
// window-count override (JJ_MSM_WINDOWS): fewer than 16 windows means windows of 17+ bits, i.e. more than 128 coarse bins of 256 buckets
// per window -- beyond the LDS arrays of k_msm_part_hist / k_msm_part_scatter (and bucket arrays of hundreds of MB)
constexpr int MSM_WINDOWS_MIN = 16, MSM_WINDOWS_MAX = 36;
struct jj_msm_job {
  jj_ctx* c = nullptr;
  hipEvent_t ev = nullptr;
  uint8_t* host = nullptr;      // page-locked: nrec records, REC_MAX_BYTES apart
  size_t cap = 0;
  size_t nrec = 0;
  // a job of jj_msm_allgather_begin: this rank's record followed by the gathered ones, in device memory the JOB owns (a lane's next
  // job must not overwrite what a finish may still have to copy)
  void* gdev = nullptr; size_t gdev_cap = 0;
  int gathered = 0;             // records the all_gather delivers (0: a local job)
  bool folded = false;          // the fold kernel was queued: host holds the ONE folded record, or a zero header if the layouts differ
};

struct jj_ctx {
  std::recursive_mutex mu;       // every entry point locks its context: calls from several host threads are serialised
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t order_ev = nullptr;   // orders a newly selected launch stream after the work queued on the previous one
  int cus = 0, clock_khz = 0, wave = 64;
  std::string err;
  // staging for host-pointer arguments (inputs 0..3, outputs 0..1) and kernel workspaces
  DevBuf in[4], out[2], okb, ws_tmp[4], sqrt_tabs;      // (the workspaces of the MSM live in its lanes)
  // kernel workspaces of the batch entry points: extended SoA, normalisation scratch, var-base window tables, the waves' work cursor.
  // Every launch helper goes through `ws`; the host-buffer pipeline points it at the set of the chunk's slot (its two slots run on
  // their own compute streams, so that the kernels of neighbouring chunks overlap), everything else uses ws0.
  WorkSet ws0;
  WorkSet* ws = &ws0;
  SqrtTables sqrt_tables{nullptr, nullptr};
  int msm_segments = -1;         // bucket accumulation: 1 = length-sorted segments, 0 = fixed chunks + fix-up, -1 = segments from MSM_LARGE_MIN (147 456) terms
                                 // (2-4 % faster there, slower below: more launches) (JJ_MSM_ACCUM=segments|chunks)
  int msm_seg_len = 0;           // segment length override (JJ_MSM_SEG_LEN; 0 = twice the mean bucket of the widest windows, clamped to [32, 1024])
  int msm_chunk = 0;             // accumulation chunk override (option msm_chunk; 0 = from msm_chunk_waves)
  int msm_sort_blocks_per_cu = 2; // one-pass sort (k_msm_front2 / k_msm_scatter2): (part, slot) blocks per CU the parts are cut for (option msm_sort_blocks_per_cu, 1..4)
  int msm_chunk_waves = 2;       // chunked accumulation: rounds of one-wave-per-SIMD workgroups per CU the launch is sized for (option msm_chunk_waves, 1..8)
  int msm_reduce_chunk = 0;      // bucket-reduce chunk length (0 = from the bucket count, see msm_enqueue_pippenger; JJ_MSM_REDUCE_CHUNK, a power of two)
  int msm_l1_rows = -1;          // two-level bucket reduce (k_msm_reduce_l1 / _l2): rows R of the bucket matrix a level-1 lane sums (a power of two, 2..64); 0 = one level (k_msm_reduce_fold);
                                 // -1 = from the bucket count (msm_enqueue_pippenger; JJ_MSM_REDUCE_L1)
  int msm_l2_chunk = 0;          // elements per level-2 quad (0 = the shortest for which the workgroups fit one per CU; JJ_MSM_REDUCE_L2_CHUNK, a power of two)
  bool msm_fused_hist = true;    // two-pass sort: coarse histogram inside the conversion kernel + atomic run reservation (JJ_MSM_SORT_HIST=separate: k_msm_part_hist / _plan, round 4)
  int msm_two_pass = -1;         // counting sort in two passes (coarse bin, then low 8 bits): always above 4096 buckets per window, never below; at exactly 4096: 0 = one pass, else two (JJ_MSM_SORT=1pass|2pass)
  int msm_pass_log2 = 24;        // terms per Pippenger pass (JJ_MSM_PASS_LOG2 overrides; for tests)
  // optional per-call kernel timing (HIP events on the launch stream): e0 | main kernel | e1 | normalise tail | e2
  // pipelined host-buffer path: copy streams + two device slots (caller buffers are page-locked in place)
  struct Pipe {
    hipStream_t h2d = nullptr, d2h = nullptr, cs[2] = {nullptr, nullptr};   // copy streams; one compute stream per slot
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr}, ev_start = nullptr, ev_tail = nullptr;
    DevBuf din[2], dout[2];
    WorkSet wset;                          // kernel workspaces of slot 1 (slot 0 uses the context's ws0)
    bool ready = false;
  } pipe;
  int dec_c_mid = 8;                     // decoder, batches of 2^20 .. 2^21 - 1 encodings (the host pipeline's chunk): encodings per lane of the shared inversion.  8 = two waves
                                         // per SIMD: 444 M/s against 431 with 16 (one wave per SIMD) and 396 with 4 (profiles/r4_pcie_inclusive.txt); JJ_DEC_C_MID = 8 | 16.
                                         // (The normalisation kernel stays at 16 there: 8 and 4 measured slower, same file.)
  // Pageable caller memory: bounce (default) = the chunks pass through page-locked staging buffers of the context, copied by a few host
  // threads beside the GPU's work -- no registration of the caller's memory, so a result array the caller has just allocated costs only its
  // page faults, spread over the copy threads (2^24 fixed-base units into a new 1 GB array: 99 ms with in-place page-locking, of which the
  // kernel's serial page faults and pinning are 68 ms); register = page-lock the caller's arrays in place for the call (no CPU copies; as
  // fast when the same arrays come back call after call, the runtime caches the pinning).  JJ_PIPE_PAGEABLE=bounce|register
  bool pipe_bounce = true;
  int pipe_copy_threads = 0;             // threads of the bounce path's copy pool (JJ_PIPE_COPY_THREADS; 0 = min(8, hardware threads / 2))
  HostCopyPool* copy_pool = nullptr;
  uint8_t* stage_in[3] = {nullptr, nullptr, nullptr}; uint8_t* stage_out[3] = {nullptr, nullptr, nullptr}; size_t stage_in_cap = 0, stage_out_cap = 0;   // three slots: the host runs two chunks ahead of its copies out
  hipEvent_t ev_stage[3] = {nullptr, nullptr, nullptr};     // chunk k's copy out of the device has reached stage_out[k % 3]
  bool pipe_ramp = true;                 // host-buffer pipeline: first and last chunk a quarter of the others (JJ_PIPE_RAMP=0: uniform)
  bool pipe_prefault = true;             // pageable result arrays are touched by several threads before they are page-locked (JJ_PIPE_PREFAULT=0: off)
  int pipe_mode = 1;                     // compute streams of the host-buffer pipeline (JJ_PIPE_STREAMS):
                                         //   1  all kernels of all chunks on one stream;
                                         //   2  the chunks of the two slots on two streams (own workspaces): 1.7x SLOWER for the fixed-base and decoder pipelines
                                         //      (2^24 units 31.4 -> 53.6 ms: two kernels that each fill the CUs time-share them), experiment knob only;
                                         //   3  the first kernel of every chunk (ladder / comb / decoder) on one stream, the kernels that follow it (normalisation,
                                         //      flag kernels) on a second one, so that the latency-bound tail of chunk k runs beside the main kernel of chunk k + 1:
                                         //      equal to mode 1 within noise (fixed-base 514-524 against 530 M/s, var-base 0.91 against 0.89-0.90 of device-resident).
                                         //   Modes 2 and 3 need GPU_MAX_HW_QUEUES >= 8: with HIP's default of 4 hardware queues per process the fifth stream in use
                                         //   shares a queue with another one and the copies serialise behind the kernels (1.7x slower, profiles/r4_pcie_inclusive.txt).
  hipStream_t pipe_tail = nullptr;       // mode 3, inside a pipelined call: the stream pipe_to_tail() moves the chunk's remaining launches to
  hipEvent_t pipe_tail_ev = nullptr;
  size_t pipe_chunk = 0;                 // elements per pipeline chunk: 0 = per entry point (pipe_chunk_for), else JJ_PIPE_CHUNK_LOG2
  // MSM jobs (jj_msm_begin / jj_msm_finish): free list of page-locked record buffers + events
  std::vector<jj_msm_job*> job_pool;
  MsmLane lanes[MSM_LANES_MAX + 1];   // [0]: the context's launch stream (synchronous calls, host-staged inputs); [1 ..]: streams of their own for the jobs in flight
  int msm_lanes = 3;             // lanes that device-pointer jobs of jj_msm_begin / jj_msm_allgather_begin alternate over (option msm_lanes, 1..4; round 6: three -- with three or more jobs in flight 2^17 terms 0.245 -> 0.221 ms per MSM, 2^18 0.385 -> 0.370, 2^20 equal, four lanes no better: profiles/r6_msm_lanes.txt; memory per lane in use; 1: every job on the context's stream)
  unsigned next_lane = 0;
  uint8_t host_out[8][64];       // results on their way to a device pointer (ring: the copies are asynchronous)
  int host_out_next = 0;
  bool msm_host_split = true;    // host arrays of 2^19 terms and more: two passes, the second half's copy beside the first half's kernels (JJ_MSM_HOST_SPLIT=0: off)
  int msm_small_blk = 4;         // small-batch path: at most this many 64-quad workgroups per window (JJ_MSM_SMALL_BLK, 1..64; 4 x 64 windows = one per CU)
  int msm_windows = 0;           // number of windows W (0 = from n; JJ_MSM_WINDOWS, 16..36: the two-pass sort holds at most 128 coarse bins per window, i.e. windows of at most 16 bits)
  int msm_small_max = 1 << 14;   // batches up to this size take the two-launch small-batch path (JJ_MSM_SMALL_MAX; 0 = never)
  bool torsion_ladder = false;   // subgroup test: false = Tate pairing (k_torsion_free), true = multiply by r (reference definition)
  bool fb_const_time = true;     // fixed-base window select: true = lane-staged + ds_bpermute shuffle, false = per-lane LDS gather
  int fb_default_kind = 7;       // what window_bits = 0 means: 7 = signed comb (32 additions + 3 doublings), 6 = signed 6-bit windows (43 additions); JJ_FIXEDBASE_DEFAULT
  bool vb_default_ct = true;     // jj_varbase_mul / _compressed run the constant-time ladder (always, in the shipped library; -DJJ_EXPERIMENTS probe builds read JJ_VARBASE_DEFAULT=vartime for A/B runs of the table ladder)
  int vb_ct_window = 3;          // constant-time ladder: signed window width, 3 (k_varbase_ct3) or 2 (k_varbase_ct); JJ_VB_CT_WINDOW
  int vb_quad_max = 32768;       // batches up to this size run one scalar-mul per quad of lanes (JJ_VB_QUAD_MAX; 0 = never)
  int fb_gather_blocks_per_cu = 3;   // wide-window fixed-base kernel: resident blocks of 256 per CU (JJ_FB_GATHER_BLOCKS_PER_CU)
  int vb_blocks_per_cu = 2;      // var-base ladder: resident blocks of 256 per CU (the ladder holds ~190 VGPRs: 2 waves per SIMD); JJ_VB_BLOCKS_PER_CU
  // multi-rank MSM exchange (jj_ctx_set_comm / jj_msm_allgather): the caller's RCCL communicator, ncclAllGather of the library that made it
  typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
  void* comm = nullptr; int comm_rank = 0, comm_nranks = 1; AllGatherFn all_gather = nullptr;
  DevBuf gather_dev, poison_dev; uint8_t* gather_host = nullptr; size_t gather_host_cap = 0;      // poison_dev: the all-zero record a failing rank gathers (msm_post_poison)
  bool msm_front1 = true;          // one-pass sort: conversion + tile histograms / plan + scatter in two launches (k_msm_front1, k_msm_scatter1); false: the four launches of rounds 2-5
  bool msm_front1_lds_set = false;
  bool msm_acc_lds = true;         // chunked accumulation: the slot's bucket offsets staged in LDS (k_msm_accumulate<true>); false: read from memory (rounds 2-5)
  bool msm_hist_lds_set = false;   // k_msm_convert_hist's LDS carve-out was requested on this context's device
  bool msm_fold_dev = true;      // gathered records are folded window by window on the device before ONE record goes to the host tail (JJ_MSM_FOLD=host: every record is copied and the host adds them)
  int msm_fold_min = 8;          // ... from this many records (JJ_MSM_FOLD_MIN, 2..4096): at 8 the two paths cost the same (57 us per call, profiles/r5_msm_partition_cost.txt), beyond it the host path grows by ~2.3 us per record while the fold stays put
  // result pool (jj_result_acquire / _release): page-locked result buffers owned by the context, handed out and taken back, so that a caller whose API
  // returns a NEW result per call (every batch function of the reference: `-> Vec<..>`, src/lib.rs:541-627, 1084-1107) pays neither the page faults of a
  // fresh array nor a registration per call
  struct PoolBuf { uint8_t* p; size_t cap; bool in_use; };
  std::vector<PoolBuf> result_pool;
  size_t result_pool_keep = (size_t)4 << 30;      // released buffers are kept for reuse while the pool holds at most this many bytes (JJ_RESULT_POOL_MB)
  bool profile = false;
  struct Rec { hipEvent_t e0, e1, e2; };
  std::vector<Rec> recs;
  size_t rec_used = 0;
};

#define HIPCHK(ctx, call)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      char b_[256];                                                                         \
      snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      (ctx)->err = b_;                                                                      \
      return JJ_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)

// entry of every API function: serialise the host threads that share this context, select its device
#define JJ_ENTER(ctx) std::lock_guard<std::recursive_mutex> jj_lock_((ctx)->mu); HIPCHK(ctx, hipSetDevice((ctx)->device))

// ---- shared by the translation units (defined in jj_pipeline.hip unless noted; hidden visibility: only the JJ_API entry points are exported)
constexpr size_t BOUNCE_MIN_BYTES = (size_t)16 << 20;     // one staging slot
// Pageable arrays of 1 MB and more never reach the runtime: from that size (GPU_PINNED_MIN_XFER_SIZE) hipMemcpy page-locks the CALLER's pages
// for the transfer -- for a copy to the device, read-only -- and keeps such ranges cached.  Caller arrays on the C heap share their first and
// last page with their neighbours: a later transfer (or registration) that WRITES through such a page met the cached read-only mapping once in
// ~2000 rounds of tests/soak_host.py ("Memory access fault by GPU ... Write access to a read-only page").  Through the context's own page-locked
// slots the GPU never touches caller pages at all (in every mode: JJ_PIPE_PAGEABLE=register page-locks caller arrays itself, read-write, for the
// pipelined entry points only).  Smaller arrays go through the runtime's staging buffer, which does not page-lock them either.
constexpr size_t BOUNCE_THRESHOLD = (size_t)1 << 20;
// The library page-locks CALLER memory itself in two places only -- JJ_PIPE_PAGEABLE=register and the whole-batch registration of jj_multi_*
// -- and only ranges that OWN THEIR PAGES (owns_its_pages: both ends page-aligned; round 4 went by size alone -- 64 MB and more, "above the C
// library's mmap threshold" -- which holds for glibc malloc only and still shares the first and last page of an unaligned mapping) of at least
// 1 MB (below that the staging copy is cheaper than the two system calls).  Everything else takes the staging slots.
constexpr size_t REGISTER_MIN_BYTES = (size_t)1 << 20;
bool owns_its_pages(const void* p, size_t bytes);
struct OutRef { void* user; void* dev; size_t bytes; bool host; };
struct HostIn { const void* p; size_t elem; };
struct HostOut { void* p; size_t elem; };
static inline SoA soa_of(DevBuf& b, size_t n) { SoA s; s.base = (u32*)b.p; s.n = n; return s; }
static inline unsigned blocks_for(size_t n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }
static const uint8_t AFFINE_IDENTITY_BYTES[64] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1};
void prof_mark(jj_ctx* c, int which);
int switch_stream(jj_ctx* c, hipStream_t s);
int ensure(jj_ctx* c, DevBuf& b, size_t bytes);
bool is_device_ptr(const void* p);
bool is_pinned_host(const void* p, size_t bytes);
int host_to_dev_bounced(jj_ctx* c, void* dev, const void* host, size_t bytes, hipStream_t stream = nullptr, size_t* seq = nullptr);
int dev_to_host_bounced(jj_ctx* c, void* host, const void* dev, size_t bytes);
int stage_in(jj_ctx* c, int slot, const void* p, size_t bytes, const void** dev);
int stage_out(jj_ctx* c, DevBuf& buf, void* p, size_t bytes, OutRef* o);
int finish_out(jj_ctx* c, const OutRef& o, bool* need_sync);
int finish(jj_ctx* c, bool need_sync);
int pipe_prepare(jj_ctx* c, size_t in_bytes, size_t out_bytes);
bool all_host(std::initializer_list<const void*> ptrs);
void prefault_parallel(void* p, size_t bytes);
size_t pipe_chunk_for(const jj_ctx* c, size_t n, int pref_log2, size_t quantum = 0);
std::vector<size_t> pipe_chunk_bounds(size_t n, size_t CH, size_t quantum, bool ramp);
size_t msm_host_pass_terms(size_t n, int pass_log2, bool split);
int pipe_to_tail(jj_ctx* c);
int stage_ensure(jj_ctx* c, size_t in_bytes, size_t out_bytes);
int stage_in_drain(jj_ctx* c, size_t seq);
int jj_batch_init(jj_ctx* c);                          // jj_abi.hip: LDS carve-outs of the fixed-base kernels, square-root tables (called by jj_ctx_create)
int msm_begin_locked(jj_ctx* c, size_t n, const void* scalars, const void* points, int part_w0, int part_stride, bool spread, jj_msm_job** out);   // jj_msm.hip
void msm_job_put(jj_ctx* c, jj_msm_job* j);            // jj_msm.hip

template <int NIN, int NOUT, class Body>
static int run_pipelined(jj_ctx* c, size_t n, size_t CH, const HostIn (&in)[NIN], const HostOut (&out)[NOUT], Body body, size_t quantum = 0) {
  size_t in_stride = 0, out_stride = 0;
  for (int k = 0; k < NIN; k++) in_stride += in[k].elem;
  for (int k = 0; k < NOUT; k++) out_stride += out[k].elem;
  int rc = pipe_prepare(c, in_stride * CH, out_stride * CH); if (rc) return rc;
#ifdef JJ_EXPERIMENTS
  const bool dbg = getenv("JJ_PIPE_DEBUG") != nullptr;
#else
  const bool dbg = false;
#endif
  timespec ts0, ts1, ts2, ts3; clock_gettime(CLOCK_MONOTONIC, &ts0);
  // Memory that is page-locked already (jj_host_alloc / hipHostMalloc, or registered by the caller -- jj_multi_* registers the whole
  // batch once before it cuts it into per-device shards, whose boundaries are not page-aligned) is copied from and to as it is.
  // Pageable arrays go through the context's staging buffers (bounce, the default) or are page-locked in place for this call.
  bool pin_in[NIN], pin_out[NOUT], any_bounce = false;
  void* locked[NIN + NOUT]; int nlocked = 0; bool ok = true;
  for (int k = 0; k < NIN; k++) pin_in[k] = is_pinned_host(in[k].p, n * in[k].elem);
  for (int k = 0; k < NOUT; k++) pin_out[k] = is_pinned_host(out[k].p, n * out[k].elem);
  if (!c->pipe_bounce) {
    // JJ_PIPE_PAGEABLE=register: arrays that own their pages (both ends page-aligned) are page-locked in place for this call; all others take the
    // staging slots like in the default mode (see REGISTER_MIN_BYTES)
    for (int k = 0; k < NIN && ok; k++) {
      if (pin_in[k] || n * in[k].elem < REGISTER_MIN_BYTES || !owns_its_pages(in[k].p, n * in[k].elem)) continue;
      if (hipHostRegister(const_cast<void*>(in[k].p), n * in[k].elem, hipHostRegisterDefault) == hipSuccess) { locked[nlocked++] = const_cast<void*>(in[k].p); pin_in[k] = true; } else ok = false;
    }
    for (int k = 0; k < NOUT && ok; k++) {
      if (pin_out[k] || n * out[k].elem < REGISTER_MIN_BYTES || !owns_its_pages(out[k].p, n * out[k].elem)) continue;
      if (c->pipe_prefault) prefault_parallel(out[k].p, n * out[k].elem);
      if (hipHostRegister(out[k].p, n * out[k].elem, hipHostRegisterDefault) == hipSuccess) { locked[nlocked++] = out[k].p; pin_out[k] = true; } else ok = false;
    }
  }
  for (int k = 0; k < NIN; k++) any_bounce |= !pin_in[k];
  for (int k = 0; k < NOUT; k++) any_bounce |= !pin_out[k];
  auto unlock = [&]() { for (int k = 0; k < nlocked; k++) (void)hipHostUnregister(locked[k]); };
  if (!ok) { (void)hipGetLastError(); unlock(); return 1; }
  if (any_bounce && (rc = stage_ensure(c, in_stride * CH, out_stride * CH))) { unlock(); return rc; }
  clock_gettime(CLOCK_MONOTONIC, &ts1);
  jj_ctx::Pipe& P = c->pipe;
  hipStream_t saved = c->stream;
  // One compute stream for all chunks (default); modes 2 and 3: see jj_ctx::pipe_mode.  The compute streams start after the work
  // already queued on the context's launch stream.
  const int mode = c->pipe_mode;
  const bool two = mode == 2;
  hipStream_t cs[2] = {mode == 1 ? c->own_stream : P.cs[0], two ? P.cs[1] : (mode == 1 ? c->own_stream : P.cs[0])};     // main stream of slot 0 / 1
  WorkSet* wsets[2] = {&c->ws0, mode == 1 ? &c->ws0 : &P.wset};
  c->pipe_tail = mode == 3 ? P.cs[1] : nullptr;
  c->pipe_tail_ev = P.ev_tail;
  // Chunk schedule: chunks of CH units, except that the first and the last one are a quarter of that when the batch has at least four
  // chunks -- the copy in of the first chunk and the copy out of the last one are the two transfers nothing overlaps
  // (2^24 fixed-base units, chunks of 2^20: 0.6 ms + 1.3 ms of 31.5 ms; JJ_PIPE_RAMP=0: uniform chunks).
  const std::vector<size_t> bounds = pipe_chunk_bounds(n, CH, quantum, c->pipe_ramp);       // chunk k = [bounds[k], bounds[k + 1])
  const size_t nchunks = bounds.size() - 1;
  // bounce path: the host stages chunk k in and queues it, THEN moves the results of chunk k - 2 from their staging slot to the caller's array
  // (waiting for that chunk's copy out of the device): it stays two chunks ahead of the GPU, which therefore never waits for a host copy.
  // Staging slots rotate over three (a chunk's slot is free again when the chunk three before it has been copied out, which happened one
  // iteration earlier), device slots over two as in the page-locked case.
  auto copy_out = [&](size_t k) -> hipError_t {
    const int g = (int)(k % 3); const size_t lo = bounds[k], cn = bounds[k + 1] - lo;
    const hipError_t e = hipEventSynchronize(c->ev_stage[g]);
    if (e != hipSuccess) return e;
    size_t off = 0;
    for (int j = 0; j < NOUT; j++) {
      if (!pin_out[j]) c->copy_pool->copy((uint8_t*)out[j].p + lo * out[j].elem, c->stage_out[g] + off, cn * out[j].elem);
      off += CH * out[j].elem;
    }
    return hipSuccess;
  };
  rc = JJ_OK;
  #define PIPE_CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { c->err = std::string(#call) + " failed: " + hipGetErrorString(e_); rc = JJ_ERR_HIP; goto done; } } while (0)
  PIPE_CHK(hipEventRecord(P.ev_start, saved));
  PIPE_CHK(hipStreamWaitEvent(cs[0], P.ev_start, 0));
  if (cs[1] != cs[0]) PIPE_CHK(hipStreamWaitEvent(cs[1], P.ev_start, 0));
  if (mode == 3) PIPE_CHK(hipStreamWaitEvent(P.cs[1], P.ev_start, 0));
  for (size_t k = 0; k < nchunks; k++) {
    const int s = (int)(k & 1); const size_t lo = bounds[k], cn = bounds[k + 1] - lo;
    const int g = (int)(k % 3);
    const void* din[NIN]; void* dout[NOUT];
    size_t off = 0;
    if (k >= 2) PIPE_CHK(hipStreamWaitEvent(P.h2d, P.ev_done[s], 0));            // slot's previous kernels have consumed din[s]
    for (int j = 0; j < NIN; j++) {
      din[j] = (uint8_t*)P.din[s].p + off;
      const uint8_t* src = (const uint8_t*)in[j].p + lo * in[j].elem;
      if (!pin_in[j]) { c->copy_pool->copy(c->stage_in[g] + off, src, cn * in[j].elem); src = c->stage_in[g] + off; }      // (its last reader, the copy in of chunk k - 3, finished before that chunk's results were waited for)
      PIPE_CHK(hipMemcpyAsync((uint8_t*)P.din[s].p + off, src, cn * in[j].elem, hipMemcpyHostToDevice, P.h2d));
      off += CH * in[j].elem;
    }
    PIPE_CHK(hipEventRecord(P.ev_in[s], P.h2d));
    c->stream = cs[s]; c->ws = wsets[s];
    PIPE_CHK(hipStreamWaitEvent(c->stream, P.ev_in[s], 0));
    if (k >= 2) PIPE_CHK(hipStreamWaitEvent(c->stream, P.ev_out[s], 0));         // slot's previous results have left dout[s]
    off = 0;
    for (int j = 0; j < NOUT; j++) { dout[j] = (uint8_t*)P.dout[s].p + off; off += CH * out[j].elem; }
    if ((rc = body(cn, din, dout))) goto done;
    PIPE_CHK(hipEventRecord(P.ev_done[s], c->stream));
    PIPE_CHK(hipStreamWaitEvent(P.d2h, P.ev_done[s], 0));
    off = 0;
    for (int j = 0; j < NOUT; j++) {
      uint8_t* dst = pin_out[j] ? (uint8_t*)out[j].p + lo * out[j].elem : c->stage_out[g] + off;
      PIPE_CHK(hipMemcpyAsync(dst, (uint8_t*)P.dout[s].p + off, cn * out[j].elem, hipMemcpyDeviceToHost, P.d2h));
      off += CH * out[j].elem;
    }
    PIPE_CHK(hipEventRecord(P.ev_out[s], P.d2h));
    if (any_bounce) { PIPE_CHK(hipEventRecord(c->ev_stage[g], P.d2h)); if (k >= 2) PIPE_CHK(copy_out(k - 2)); }
  }
  clock_gettime(CLOCK_MONOTONIC, &ts2);
  if (any_bounce) { if (nchunks >= 2) PIPE_CHK(copy_out(nchunks - 2)); PIPE_CHK(copy_out(nchunks - 1)); }
  PIPE_CHK(hipStreamSynchronize(P.d2h));
  PIPE_CHK(hipGetLastError());
  clock_gettime(CLOCK_MONOTONIC, &ts3);
  if (dbg) {
    auto ms = [](const timespec& a, const timespec& b) { return (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6; };
    fprintf(stderr, "[jj pipe] n=%zu chunk=%zu chunks=%zu stream mode=%d %s: prepare %.2f ms, loop %.2f ms, drain %.2f ms\n", n, CH, nchunks, mode,
            any_bounce ? "bounce" : (nlocked ? "registered in place" : "page-locked"), ms(ts0, ts1), ms(ts1, ts2), ms(ts2, ts3));
  }
done:
  #undef PIPE_CHK
  if (rc != JJ_OK) { (void)hipStreamSynchronize(P.h2d); (void)hipStreamSynchronize(cs[0]); (void)hipStreamSynchronize(cs[1]); if (P.cs[1]) (void)hipStreamSynchronize(P.cs[1]); (void)hipStreamSynchronize(P.d2h); }
  c->pipe_tail = nullptr;
  // every chunk's kernels finished before its copy out did, and all copies were waited for (or, on error, every stream was drained):
  // nothing of this call is in flight any more, the context returns to the stream and workspaces it came with
  c->stream = saved; c->ws = &c->ws0;
  unlock();
  return rc;
}
