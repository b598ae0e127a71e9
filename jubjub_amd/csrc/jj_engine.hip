// libjubjub_hip.so — host side of the C ABI declared in include/jubjub_hip.h.
// Owns the device context (stream, staging + workspace buffers) and launches the kernels in jj_kernels.h.
// There is no CPU fallback: without a gfx950 device jj_ctx_create fails with JJ_ERR_NODEVICE.
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
#include "jj_kernels.h"
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
  DevBuf buf[8], ctl, bigpart, seg, rec;             // kprime, niels, offsets, idx, buckets, heads/records, -, tile counts | counters + lists | big-bucket partials | segments | record
  bool owned = false;
};
constexpr int MSM_LANES_MAX = 4;
// window-count override (JJ_MSM_WINDOWS): fewer than 16 windows means windows of 17+ bits, i.e. more than 128 coarse bins of 256 buckets
// per window -- beyond the LDS arrays of k_msm_part_hist / k_msm_part_scatter (and bucket arrays of hundreds of MB)
constexpr int MSM_WINDOWS_MIN = 16, MSM_WINDOWS_MAX = 36;
struct jj_msm_job {
  jj_ctx* c = nullptr;
  hipEvent_t ev = nullptr;
  uint8_t* host = nullptr;      // page-locked: nrec records, REC_MAX_BYTES apart
  size_t cap = 0;
  size_t nrec = 0;
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
  int msm_chunk = 0;             // accumulation chunk override (JJ_MSM_CHUNK; 0 = scale with n)
  int msm_reduce_chunk = 0;      // bucket-reduce chunk length (0 = from the bucket count, see msm_enqueue_pippenger; JJ_MSM_REDUCE_CHUNK, a power of two)
  int msm_l1_rows = -1;          // two-level bucket reduce (k_msm_reduce_l1 / _l2): rows R of the bucket matrix a level-1 lane sums (a power of two, 2..64); 0 = one level (k_msm_reduce_fold);
                                 // -1 = from the bucket count (msm_enqueue_pippenger; JJ_MSM_REDUCE_L1)
  int msm_l2_chunk = 0;          // elements per level-2 quad (0 = the shortest for which the workgroups fit one per CU; JJ_MSM_REDUCE_L2_CHUNK, a power of two)
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
  MsmLane lanes[MSM_LANES_MAX];
  int msm_lanes = 2;             // lanes that device-pointer jobs of jj_msm_begin alternate over (JJ_MSM_LANES, 1..4; memory per lane in use)
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
  int vb_quad_max = 32768;       // batches up to this size run one scalar-mul per quad of lanes (JJ_VB_QUAD_MAX; 0 = never)
  int fb_gather_blocks_per_cu = 3;   // wide-window fixed-base kernel: resident blocks of 256 per CU (JJ_FB_GATHER_BLOCKS_PER_CU)
  int vb_blocks_per_cu = 2;      // var-base ladder: resident blocks of 256 per CU (the ladder holds ~190 VGPRs: 2 waves per SIMD); JJ_VB_BLOCKS_PER_CU
  // multi-rank MSM exchange (jj_ctx_set_comm / jj_msm_allgather): the caller's RCCL communicator, ncclAllGather of the library that made it
  typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
  void* comm = nullptr; int comm_rank = 0, comm_nranks = 1; AllGatherFn all_gather = nullptr;
  DevBuf gather_dev; uint8_t* gather_host = nullptr; size_t gather_host_cap = 0;
  bool profile = false;
  struct Rec { hipEvent_t e0, e1, e2; };
  std::vector<Rec> recs;
  size_t rec_used = 0;
};

static void prof_mark(jj_ctx* c, int which) {
  if (!c->profile) return;
  if (which == 0) {
    if (c->rec_used == c->recs.size()) {
      jj_ctx::Rec r;
      if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess || hipEventCreate(&r.e2) != hipSuccess) return;
      c->recs.push_back(r);
    }
    (void)hipEventRecord(c->recs[c->rec_used].e0, c->stream);
  } else if (c->rec_used < c->recs.size()) {
    if (which == 1) (void)hipEventRecord(c->recs[c->rec_used].e1, c->stream);
    else { (void)hipEventRecord(c->recs[c->rec_used].e2, c->stream); c->rec_used++; }
  }
}

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

static int switch_stream(jj_ctx* c, hipStream_t s);
static int ensure(jj_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return JJ_OK;
  if (b.p) { HIPCHK(c, hipDeviceSynchronize()); HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }   // growth only; the buffer may be in use on any of the context's streams
  size_t want = std::max(bytes, (size_t)4096);
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) { c->err = std::string("hipMalloc failed: ") + hipGetErrorString(e); b.p = nullptr; return JJ_ERR_NOMEM; }
  b.cap = want;
  return JJ_OK;
}

static bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

static bool is_pinned_host(const void* p, size_t bytes);
static int host_to_dev_bounced(jj_ctx* c, void* dev, const void* host, size_t bytes, hipStream_t stream = nullptr, size_t* seq = nullptr);
static int dev_to_host_bounced(jj_ctx* c, void* host, const void* dev, size_t bytes);
constexpr size_t BOUNCE_MIN_BYTES = (size_t)16 << 20;     // one staging slot
// Pageable arrays of 1 MB and more never reach the runtime: from that size (GPU_PINNED_MIN_XFER_SIZE) hipMemcpy page-locks the CALLER's pages
// for the transfer -- for a copy to the device, read-only -- and keeps such ranges cached.  Caller arrays on the C heap share their first and
// last page with their neighbours: a later transfer (or registration) that WRITES through such a page met the cached read-only mapping once in
// ~2000 rounds of tests/soak_host.py ("Memory access fault by GPU ... Write access to a read-only page").  Through the context's own page-locked
// slots the GPU never touches caller pages at all (in every mode: JJ_PIPE_PAGEABLE=register page-locks caller arrays itself, read-write, for the
// pipelined entry points only).  Smaller arrays go through the runtime's staging buffer, which does not page-lock them either.
constexpr size_t BOUNCE_THRESHOLD = (size_t)1 << 20;
// The library page-locks CALLER memory itself in two places only -- JJ_PIPE_PAGEABLE=register and the whole-batch registration of jj_multi_*
// -- and only arrays of 64 MB and more: those are mappings of their own (the C library's mmap threshold never exceeds 32 MB), while smaller
// arrays sit on the C heap between other objects, whose pages a registration would hand to the GPU as well.  Both GPU faults of the soak
// (above) were writes into heap-sized result arrays (the decoder's `ok` bytes, 256 KB and 1 MB) registered in place.
constexpr size_t REGISTER_MIN_BYTES = (size_t)64 << 20;
// Resolves an input pointer: device pointers pass through (must be 16-byte aligned), host data is copied into a
// staging buffer (large pageable arrays through the page-locked staging slots, see host_to_dev_bounced).
static int stage_in(jj_ctx* c, int slot, const void* p, size_t bytes, const void** dev) {
  if (bytes == 0) { *dev = nullptr; return JJ_OK; }
  if (!p) { c->err = "null input pointer"; return JJ_ERR_INVALID; }
  if (is_device_ptr(p)) {
    if (((uintptr_t)p & 15u) != 0) { c->err = "device pointers must be 16-byte aligned"; return JJ_ERR_INVALID; }
    *dev = p; return JJ_OK;
  }
  int rc = ensure(c, c->in[slot], bytes); if (rc) return rc;
  if (bytes >= BOUNCE_THRESHOLD && !is_pinned_host(p, bytes)) { if ((rc = host_to_dev_bounced(c, c->in[slot].p, p, bytes))) return rc; }
  else HIPCHK(c, hipMemcpyAsync(c->in[slot].p, p, bytes, hipMemcpyHostToDevice, c->stream));
  *dev = c->in[slot].p; return JJ_OK;
}
struct OutRef { void* user; void* dev; size_t bytes; bool host; };
static int stage_out(jj_ctx* c, DevBuf& buf, void* p, size_t bytes, OutRef* o) {
  o->user = p; o->bytes = bytes;
  if (bytes == 0) { o->dev = nullptr; o->host = false; return JJ_OK; }
  if (!p) { c->err = "null output pointer"; return JJ_ERR_INVALID; }
  if (is_device_ptr(p)) {
    if (((uintptr_t)p & 15u) != 0) { c->err = "device pointers must be 16-byte aligned"; return JJ_ERR_INVALID; }
    o->dev = p; o->host = false; return JJ_OK;
  }
  int rc = ensure(c, buf, bytes); if (rc) return rc;
  o->dev = buf.p; o->host = true; return JJ_OK;
}
static int finish_out(jj_ctx* c, const OutRef& o, bool* need_sync) {
  if (o.host) {
    if (o.bytes >= BOUNCE_THRESHOLD && !is_pinned_host(o.user, o.bytes)) { const int rc = dev_to_host_bounced(c, o.user, o.dev, o.bytes); if (rc) return rc; }
    else if (o.bytes) HIPCHK(c, hipMemcpyAsync(o.user, o.dev, o.bytes, hipMemcpyDeviceToHost, c->stream));
    *need_sync = true;
  }
  return JJ_OK;
}
static int finish(jj_ctx* c, bool need_sync) {
  HIPCHK(c, hipGetLastError());
  if (need_sync) HIPCHK(c, hipStreamSynchronize(c->stream));
  return JJ_OK;
}
static inline unsigned blocks_for(size_t n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }

// ---------------------------------------------------------------------------------------------------- host-buffer pipeline
// When every array argument is a host pointer and the batch is large, the caller's buffers are page-locked in place
// (hipHostRegister: ~1 ms per 100 MB, measured) and the batch is cut into chunks that flow over two copy streams
// while the kernels of the neighbouring chunk run:  H2D (h2d stream) -> kernels (compute stream) -> D2H (d2h stream),
// two device slots, all ordering by events (no host synchronisation inside the loop, no CPU bounce copies).
// If registration fails (e.g. overlapping or already registered buffers) the caller falls back to plain staging.
struct HostIn { const void* p; size_t elem; };
struct HostOut { void* p; size_t elem; };
static int pipe_prepare(jj_ctx* c, size_t in_bytes, size_t out_bytes) {
  jj_ctx::Pipe& P = c->pipe;
  if (!P.ready) {
    HIPCHK(c, hipStreamCreateWithFlags(&P.h2d, hipStreamNonBlocking));
    HIPCHK(c, hipStreamCreateWithFlags(&P.d2h, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&P.ev_start, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&P.ev_tail, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) {
      if (c->pipe_mode != 1) HIPCHK(c, hipStreamCreateWithFlags(&P.cs[i], hipStreamNonBlocking));   // (a process's streams share a few hardware queues: none is created unless used)
      HIPCHK(c, hipEventCreateWithFlags(&P.ev_in[i], hipEventDisableTiming));
      HIPCHK(c, hipEventCreateWithFlags(&P.ev_done[i], hipEventDisableTiming));
      HIPCHK(c, hipEventCreateWithFlags(&P.ev_out[i], hipEventDisableTiming));
    }
    P.ready = true;
  }
  for (int i = 0; i < 2; i++) {
    int rc;
    if ((rc = ensure(c, P.din[i], in_bytes))) return rc;
    if ((rc = ensure(c, P.dout[i], out_bytes))) return rc;
  }
  return JJ_OK;
}
// both ends of [p, p + bytes) lie in page-locked host memory known to the runtime
static bool is_pinned_host(const void* p, size_t bytes) {
  if (!p || !bytes) return false;
  for (const uint8_t* q : {(const uint8_t*)p, (const uint8_t*)p + bytes - 1}) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a.type != hipMemoryTypeHost) return false;
  }
  return true;
}
static bool all_host(std::initializer_list<const void*> ptrs) { for (const void* p : ptrs) if (!p || is_device_ptr(p)) return false; return true; }

// A result array the caller has just allocated (calloc / vec![0; n] / np.empty) has no pages yet: page-locking it makes the kernel fault
// every page in, one after the other, inside hipHostRegister -- 12 ms per 100 MB on the box measured (profiles/r4_pcie_probe.txt: 123 ms
// for the 1 GB result of a 2^24-unit fixed-base call, four times the call's own 31 ms).  Touching one byte per page from several
// threads first (read and write back the same value: the array's contents, if any, stay) spreads the faults over the cores.
static void prefault_parallel(void* p, size_t bytes) {
  if (bytes < ((size_t)32 << 20)) return;
  const unsigned hw = std::thread::hardware_concurrency();
  const int T = (int)std::min<size_t>(std::min<unsigned>(hw ? hw : 4, 16), bytes >> 24);
  if (T < 2) return;
  std::vector<std::thread> th;
  const size_t per = ((bytes / T) + 4095) & ~(size_t)4095;
  for (int t = 0; t < T; t++)
    th.emplace_back([=]() {
      volatile uint8_t* q = (volatile uint8_t*)p;
      const size_t lo = (size_t)t * per, hi = std::min(bytes, lo + per);
      for (size_t o = lo; o < hi; o += 4096) q[o] = q[o];
    });
  for (auto& x : th) x.join();
}
// body(cn, dev_in[k], dev_out[k]) must enqueue the chunk's kernels on c->stream.
// Returns JJ_OK, an error, or +1 when the buffers could not be page-locked (caller uses the staging path).
// Chunk length of the host-buffer pipeline for a batch of n units (0: the batch is too small to be cut, it is staged whole).
// `pref_log2` is what the entry point measured as its best chunk at its BASELINE size (profiles/r4_pcie_inclusive.txt: 2^20 for the
// fixed-base kernels, 2^21 for the decoder -- shorter chunks pay the shared inversion of their normalisation over too few points and
// leave the decoder one wave per SIMD, longer ones pay the unoverlapped first copy in and last copy out, which the short first / last
// chunk only softens; 2^18 for the var-base ladder, whose kernel time dwarfs its copies); smaller batches are cut in four, down to
// 2^16 units per chunk.
// `quantum`: the kernel's lane count when every lane takes ceil(chunk / lanes) units in a grid-stride loop (the fixed-base kernels: one
// workgroup per CU): a chunk that is not a multiple of it leaves most lanes idle during the last round -- 2^20 units over 196 608 lanes are
// 5.33 per lane, i.e. the time of 6 (-11 %) -- so the chunk and the short first / last chunk are rounded to multiples of it.
static size_t pipe_chunk_for(const jj_ctx* c, size_t n, int pref_log2, size_t quantum = 0) {
  if (c->pipe_chunk) return n >= 2 * c->pipe_chunk ? c->pipe_chunk : 0;
  size_t ch = (size_t)1 << pref_log2;
  while (ch > ((size_t)1 << 16) && n < 4 * ch) ch >>= 1;
  if (n < 4 * ch) return 0;
  if (quantum && ch >= 2 * quantum) ch = ((ch + quantum / 2) / quantum) * quantum;
  return ch;
}
// Chunk schedule of a pipelined host batch (n >= 1 units, chunks of CH): chunk k = [bounds[k], bounds[k + 1]).  With `ramp` the first and
// the last chunk are a quarter of CH when the batch has at least four chunks of at least 2^18 units (rounded to whole `quantum`s, the
// kernel's lanes per round, when CH is a multiple of it); no chunk is longer than CH + the edge.  Exported as jj_plan_host_chunks for
// the CPU-side tests.
static std::vector<size_t> pipe_chunk_bounds(size_t n, size_t CH, size_t quantum, bool ramp) {
  std::vector<size_t> bounds;
  size_t edge = (ramp && n >= 4 * CH && CH >= ((size_t)1 << 18)) ? CH / 4 : 0;
  if (edge && quantum && CH % quantum == 0) edge = std::max(quantum, (edge / quantum) * quantum);      // whole rounds of the kernel's lanes
  size_t lo = 0;
  bounds.push_back(0);
  if (edge) { lo = edge; bounds.push_back(lo); }
  while (n - lo > CH + edge) { lo += CH; bounds.push_back(lo); }
  if (edge && n - lo > edge) { lo = n - edge; bounds.push_back(lo); }
  bounds.push_back(n);
  return bounds;
}
JJ_API int jj_plan_host_chunks(size_t n, size_t chunk, size_t quantum, int ramp, size_t* bounds, size_t cap, size_t* count) {
  if (!n || !chunk || !count || (cap && !bounds)) return JJ_ERR_INVALID;
  const std::vector<size_t> b = pipe_chunk_bounds(n, chunk, quantum, ramp != 0);
  *count = b.size();
  if (b.size() > cap) return bounds ? JJ_ERR_INVALID : JJ_OK;        // cap = 0: the count only
  std::copy(b.begin(), b.end(), bounds);
  return JJ_OK;
}
// Terms per pass of an MSM over HOST arrays (msm_begin_locked): 2^pass_log2 terms at most; arrays of 2^19 terms and more are cut into two
// to eight passes of at least 2^18 terms (a multiple of 64; more passes when eight would exceed 2^pass_log2 terms each) so that the copy of
// a pass overlaps the kernels of the pass before.
static size_t msm_host_pass_terms(size_t n, int pass_log2, bool split) {
  size_t PASS = (size_t)1 << pass_log2;
  if (split && n >= ((size_t)1 << 19)) {
    const size_t passes = std::min<size_t>(8, std::max<size_t>(2, n >> 19));
    PASS = std::min(PASS, (((n + passes - 1) / passes) + 63) & ~(size_t)63);      // (arrays beyond eight full passes: more passes of 2^pass_log2 terms)
  }
  return PASS;
}
JJ_API int jj_plan_msm_host_passes(size_t n, int pass_log2, int split, size_t* pass_terms, size_t* passes) {
  if (!pass_terms || !passes || pass_log2 < 10 || pass_log2 > 24) return JJ_ERR_INVALID;
  *pass_terms = msm_host_pass_terms(n, pass_log2, split != 0);
  *passes = n ? (n + *pass_terms - 1) / *pass_terms : 0;
  return JJ_OK;
}
// Inside a pipelined call in stream mode 3: the launches that follow go to the tail stream, ordered after what the chunk has queued on
// its main stream so far.  A no-op everywhere else.
static int pipe_to_tail(jj_ctx* c) {
  if (!c->pipe_tail || c->stream == c->pipe_tail) return JJ_OK;
  HIPCHK(c, hipEventRecord(c->pipe_tail_ev, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->pipe_tail, c->pipe_tail_ev, 0));
  c->stream = c->pipe_tail;
  return JJ_OK;
}
// page-locked staging of the bounce path: three slots each way, grown on demand
static int stage_ensure(jj_ctx* c, size_t in_bytes, size_t out_bytes) {
  auto grow = [&](uint8_t* (&buf)[3], size_t& cap, size_t want) -> int {
    if (want <= cap) return JJ_OK;
    for (int i = 0; i < 3; i++) {
      if (buf[i]) (void)hipHostFree(buf[i]);
      buf[i] = nullptr;
      if (hipHostMalloc((void**)&buf[i], want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); cap = 0; c->err = "hipHostMalloc(staging) failed"; return JJ_ERR_NOMEM; }
    }
    cap = want;
    return JJ_OK;
  };
  int rc;
  if ((rc = grow(c->stage_in, c->stage_in_cap, in_bytes))) return rc;
  if ((rc = grow(c->stage_out, c->stage_out_cap, out_bytes))) return rc;
  for (int i = 0; i < 3; i++) if (!c->ev_stage[i]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_stage[i], hipEventDisableTiming));
  if (!c->copy_pool) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int T = c->pipe_copy_threads ? c->pipe_copy_threads : (int)std::min<unsigned>(8, std::max<unsigned>(2, hw / 2));
    c->copy_pool = new HostCopyPool(T - 1);          // the calling thread copies too
  }
  return JJ_OK;
}
// A large pageable array of an entry point that is not pipelined (the inputs of an MSM, the operands of a batched field or point
// operation): hipMemcpyAsync from pageable memory goes through the runtime's own single-threaded staging (3 - 30 GB/s measured,
// profiles/r4_pcie_probe.txt); here the copy pool fills page-locked staging slots while the previous slot's DMA runs.
static int host_to_dev_bounced(jj_ctx* c, void* dev, const void* host, size_t bytes, hipStream_t stream, size_t* seq) {
  // seq: a slot counter the caller keeps over SEVERAL arrays (and drains once with stage_in_drain): the last slots' DMA of one array then
  // runs beside the host copy of the next array's first slots, instead of being waited for between the arrays
  if (!stream) stream = c->stream;
  // slots of whole MB, 16 MB at most; a caller that keeps `seq` over several arrays has copies in flight between them, so its slots must not
  // be re-allocated on the way: full-size slots from the start
  const size_t CHB = seq ? BOUNCE_MIN_BYTES : std::min(BOUNCE_MIN_BYTES, (bytes + 0xfffff) & ~(size_t)0xfffff);
  int rc = stage_ensure(c, std::max(CHB, c->stage_in_cap), c->stage_out_cap); if (rc) return rc;
  size_t k0 = 0;
  size_t& k = seq ? *seq : k0;
  for (size_t lo = 0; lo < bytes; lo += CHB, k++) {
    const int g = (int)(k % 3); const size_t cn = std::min(CHB, bytes - lo);
    if (k >= 3) HIPCHK(c, hipEventSynchronize(c->ev_stage[g]));          // the slot's previous DMA has read it
    c->copy_pool->copy(c->stage_in[g], (const uint8_t*)host + lo, cn);
    HIPCHK(c, hipMemcpyAsync((uint8_t*)dev + lo, c->stage_in[g], cn, hipMemcpyHostToDevice, stream));
    HIPCHK(c, hipEventRecord(c->ev_stage[g], stream));
  }
  if (!seq) for (size_t j = (k > 3 ? k - 3 : 0); j < k; j++) HIPCHK(c, hipEventSynchronize(c->ev_stage[j % 3]));   // the slots are free for the next user
  return JJ_OK;
}
static int stage_in_drain(jj_ctx* c, size_t seq) {
  for (size_t j = (seq > 3 ? seq - 3 : 0); j < seq; j++) HIPCHK(c, hipEventSynchronize(c->ev_stage[j % 3]));
  return JJ_OK;
}
static int dev_to_host_bounced(jj_ctx* c, void* host, const void* dev, size_t bytes) {
  const size_t CHB = std::min(BOUNCE_MIN_BYTES, (bytes + 0xfffff) & ~(size_t)0xfffff);
  int rc = stage_ensure(c, c->stage_in_cap, std::max(CHB, c->stage_out_cap)); if (rc) return rc;
  const size_t nch = (bytes + CHB - 1) / CHB;
  auto drain = [&](size_t k) -> hipError_t {
    const hipError_t e = hipEventSynchronize(c->ev_stage[k % 3]);
    if (e != hipSuccess) return e;
    c->copy_pool->copy((uint8_t*)host + k * CHB, c->stage_out[k % 3], std::min(CHB, bytes - k * CHB));
    return hipSuccess;
  };
  for (size_t k = 0; k < nch; k++) {
    HIPCHK(c, hipMemcpyAsync(c->stage_out[k % 3], (const uint8_t*)dev + k * CHB, std::min(CHB, bytes - k * CHB), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_stage[k % 3], c->stream));
    if (k >= 2) HIPCHK(c, drain(k - 2));
  }
  if (nch >= 2) HIPCHK(c, drain(nch - 2));
  HIPCHK(c, drain(nch - 1));
  return JJ_OK;
}
template <int NIN, int NOUT, class Body>
static int run_pipelined(jj_ctx* c, size_t n, size_t CH, const HostIn (&in)[NIN], const HostOut (&out)[NOUT], Body body, size_t quantum = 0) {
  size_t in_stride = 0, out_stride = 0;
  for (int k = 0; k < NIN; k++) in_stride += in[k].elem;
  for (int k = 0; k < NOUT; k++) out_stride += out[k].elem;
  int rc = pipe_prepare(c, in_stride * CH, out_stride * CH); if (rc) return rc;
  const bool dbg = getenv("JJ_PIPE_DEBUG") != nullptr;
  timespec ts0, ts1, ts2, ts3; clock_gettime(CLOCK_MONOTONIC, &ts0);
  // Memory that is page-locked already (jj_host_alloc / hipHostMalloc, or registered by the caller -- jj_multi_* registers the whole
  // batch once before it cuts it into per-device shards, whose boundaries are not page-aligned) is copied from and to as it is.
  // Pageable arrays go through the context's staging buffers (bounce, the default) or are page-locked in place for this call.
  bool pin_in[NIN], pin_out[NOUT], any_bounce = false;
  void* locked[NIN + NOUT]; int nlocked = 0; bool ok = true;
  for (int k = 0; k < NIN; k++) pin_in[k] = is_pinned_host(in[k].p, n * in[k].elem);
  for (int k = 0; k < NOUT; k++) pin_out[k] = is_pinned_host(out[k].p, n * out[k].elem);
  if (!c->pipe_bounce) {
    // JJ_PIPE_PAGEABLE=register: arrays of REGISTER_MIN_BYTES and more are page-locked in place for this call; smaller ones take the staging
    // slots like in the default mode (see REGISTER_MIN_BYTES)
    for (int k = 0; k < NIN && ok; k++) {
      if (pin_in[k] || n * in[k].elem < REGISTER_MIN_BYTES) continue;
      if (hipHostRegister(const_cast<void*>(in[k].p), n * in[k].elem, hipHostRegisterDefault) == hipSuccess) { locked[nlocked++] = const_cast<void*>(in[k].p); pin_in[k] = true; } else ok = false;
    }
    for (int k = 0; k < NOUT && ok; k++) {
      if (pin_out[k] || n * out[k].elem < REGISTER_MIN_BYTES) continue;
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

// HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A context uses three streams in its host-
// buffer pipeline (compute, copy in, copy out) and one more per extra MSM lane; beside PyTorch's or the caller's own streams that
// exceeds four, and two streams that share a hardware queue serialise -- measured: jj_multi_* with a second context in the process
// 268 -> 523 M fixed-base scalar-muls/s, a fourth pipeline stream 316 -> 520 M/s (profiles/r4_pcie_inclusive.txt).  The runtime reads
// the variable when it initialises (first HIP call), so setting it here works whenever this library is loaded before that; a value
// the user has set is left alone.
__attribute__((constructor)) static void jj_default_hw_queues() { (void)setenv("GPU_MAX_HW_QUEUES", "8", 0); }

// ---------------------------------------------------------------------------------------------------- host buffers
// Page-locked host memory for callers that do not link HIP themselves (include/jubjub_hip.h).  The entry points recognise such
// memory (is_pinned_host) and move it with asynchronous copies on the copy streams without registering anything per call.
JJ_API int jj_host_alloc(size_t bytes, void** out) {
  if (!out) return JJ_ERR_INVALID;
  *out = nullptr;
  if (bytes == 0) return JJ_OK;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return JJ_ERR_NODEVICE; }
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return JJ_ERR_NOMEM; }
  *out = p;
  return JJ_OK;
}
JJ_API int jj_host_free(void* p) {
  if (!p) return JJ_OK;
  if (hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); return JJ_ERR_INVALID; }
  return JJ_OK;
}
JJ_API int jj_host_register(void* p, size_t bytes) {
  if (!p || !bytes) return JJ_ERR_INVALID;
  // a page-aligned start: the buffer owns the pages it is on (see REGISTER_MIN_BYTES: page-locking C-heap arrays in place hands the
  // neighbouring objects' pages to the GPU and ended in GPU memory faults)
  if ((uintptr_t)p & ((uintptr_t)sysconf(_SC_PAGESIZE) - 1)) return JJ_ERR_INVALID;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return JJ_ERR_NODEVICE; }
  const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
  if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorOutOfMemory ? JJ_ERR_NOMEM : JJ_ERR_INVALID; }
  return JJ_OK;
}
JJ_API int jj_host_unregister(void* p) {
  if (!p) return JJ_ERR_INVALID;
  if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return JJ_ERR_INVALID; }
  return JJ_OK;
}

// ---------------------------------------------------------------------------------------------------- context
JJ_API int jj_version(void) { return JJ_VERSION; }
// WnafGroup::recommended_wnaf_for_num_scalars (reference src/lib.rs:1320-1335): same thresholds, same result.
JJ_API int jj_recommended_wnaf_for_num_scalars(size_t num_scalars) {
  static const size_t rec[12] = {1, 3, 7, 20, 43, 120, 273, 563, 1630, 3128, 7933, 62569};
  int ret = 4;
  for (size_t r : rec) { if (num_scalars > r) ret++; else break; }
  return ret;
}

JJ_API int jj_ctx_create(int device, jj_ctx** out) {
  if (!out) return JJ_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) { (void)hipGetLastError(); return JJ_ERR_NODEVICE; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return JJ_ERR_NODEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return JJ_ERR_NODEVICE;   // this library ships gfx950 code only
  if (hipSetDevice(device) != hipSuccess) return JJ_ERR_HIP;
  jj_ctx* c = new jj_ctx();
  c->device = device;
  c->cus = prop.multiProcessorCount;
  c->clock_khz = prop.clockRate;
  c->wave = prop.warpSize;
  // every failure below releases what was created so far
  auto fail = [&](int code) {
    (void)hipGetLastError();
    if (c->sqrt_tabs.p) (void)hipFree(c->sqrt_tabs.p);
    if (c->order_ev) (void)hipEventDestroy(c->order_ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return code;
  };
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(JJ_ERR_HIP);
  if (hipEventCreateWithFlags(&c->order_ev, hipEventDisableTiming) != hipSuccess) return fail(JJ_ERR_HIP);
  c->stream = c->own_stream;
  if (const char* e = getenv("JJ_DEC_C_MID")) { int v = atoi(e); if (v == 8 || v == 16) c->dec_c_mid = v; }
  if (const char* e = getenv("JJ_PIPE_PAGEABLE")) c->pipe_bounce = strcmp(e, "register") != 0;
  if (const char* e = getenv("JJ_PIPE_COPY_THREADS")) { int v = atoi(e); if (v >= 0 && v <= 64) c->pipe_copy_threads = v; }
  if (const char* e = getenv("JJ_PIPE_RAMP")) c->pipe_ramp = atoi(e) != 0;
  if (const char* e = getenv("JJ_PIPE_PREFAULT")) c->pipe_prefault = atoi(e) != 0;
  if (const char* e = getenv("JJ_PIPE_STREAMS")) { int v = atoi(e); if (v >= 1 && v <= 3) c->pipe_mode = v; }
  if (const char* e = getenv("JJ_PIPE_CHUNK_LOG2")) { int v = atoi(e); if (v >= 8 && v <= 24) c->pipe_chunk = (size_t)1 << v; }   // overrides the per-entry-point chunk
  if (const char* e = getenv("JJ_MSM_WINDOWS")) { int v = atoi(e); if (v >= MSM_WINDOWS_MIN && v <= MSM_WINDOWS_MAX) c->msm_windows = v; else fprintf(stderr, "libjubjub_hip: JJ_MSM_WINDOWS=%s ignored (valid: %d..%d)\n", e, MSM_WINDOWS_MIN, MSM_WINDOWS_MAX); }
  if (const char* e = getenv("JJ_MSM_HOST_SPLIT")) c->msm_host_split = atoi(e) != 0;
  if (const char* e = getenv("JJ_MSM_LANES")) { int v = atoi(e); if (v >= 1 && v <= MSM_LANES_MAX) c->msm_lanes = v; }
  if (const char* e = getenv("JJ_MSM_SMALL_BLK")) { int v = atoi(e); if (v >= 1 && v <= MSM_TREE_QUADS) c->msm_small_blk = v; }
  if (const char* e = getenv("JJ_MSM_SMALL_MAX")) { int v = atoi(e); if (v >= 0 && v <= (1 << 20)) c->msm_small_max = v; }
  if (const char* e = getenv("JJ_MSM_ACCUM")) c->msm_segments = strcmp(e, "chunks") == 0 ? 0 : strcmp(e, "segments") == 0 ? 1 : -1;
  if (const char* e = getenv("JJ_MSM_SEG_LEN")) c->msm_seg_len = atoi(e);
  if (const char* e = getenv("JJ_MSM_CHUNK")) { int v = atoi(e); if (v >= 8 && v <= 1024) c->msm_chunk = v; }
  if (const char* e = getenv("JJ_MSM_REDUCE_CHUNK")) { int v = atoi(e); if (v >= 2 && v <= 256 && (v & (v - 1)) == 0) c->msm_reduce_chunk = v; }
  if (const char* e = getenv("JJ_MSM_REDUCE_L1")) { int v = atoi(e); if (v == 0 || (v >= 2 && v <= 64 && (v & (v - 1)) == 0)) c->msm_l1_rows = v; }
  if (const char* e = getenv("JJ_MSM_REDUCE_L2_CHUNK")) { int v = atoi(e); if (v >= 2 && v <= 64 && (v & (v - 1)) == 0) c->msm_l2_chunk = v; }
  if (const char* e = getenv("JJ_MSM_SORT")) c->msm_two_pass = !strcmp(e, "2pass") ? 1 : (!strcmp(e, "1pass") ? 0 : -1);
  if (const char* e = getenv("JJ_MSM_PASS_LOG2")) { int v = atoi(e); if (v >= 10 && v <= 24) c->msm_pass_log2 = v; }
  if (const char* e = getenv("JJ_VB_QUAD_MAX")) { int v = atoi(e); if (v >= 0 && v <= (1 << 20)) c->vb_quad_max = v; }
  if (const char* e = getenv("JJ_FB_GATHER_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 8) c->fb_gather_blocks_per_cu = v; }
  if (const char* e = getenv("JJ_VB_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 8) c->vb_blocks_per_cu = v; }
  // the fixed-base kernel needs the full 160 KiB LDS carve-out
  {
    const struct { const void* fn; int bytes; } lds_needs[] = {
      {reinterpret_cast<const void*>(k_fixedbase<true>), FB_LDS_BYTES}, {reinterpret_cast<const void*>(k_fixedbase<false>), FB_LDS_BYTES},
      {reinterpret_cast<const void*>(k_fixedbase_comb<true>), FBC_LDS_BYTES}, {reinterpret_cast<const void*>(k_fixedbase_comb<false>), FBC_LDS_BYTES}};
    for (const auto& a : lds_needs)
      if (hipFuncSetAttribute(a.fn, hipFuncAttributeMaxDynamicSharedMemorySize, a.bytes) != hipSuccess) return fail(JJ_ERR_HIP);   // the kernels could not launch later
  }
  if (const char* e = getenv("JJ_TORSION_CHECK")) c->torsion_ladder = strcmp(e, "ladder") == 0;
  if (const char* e = getenv("JJ_FIXEDBASE_SELECT")) c->fb_const_time = strcmp(e, "gather") != 0;
  if (const char* e = getenv("JJ_FIXEDBASE_DEFAULT")) { int v = atoi(e); if (v == 6 || v == 7) c->fb_default_kind = v; }
  // square-root tables (64 KiB dlog + 36 KiB powers), built on the device
  if (hipMalloc(&c->sqrt_tabs.p, 65536 + 4 * 256 * NL * 4) != hipSuccess) { c->sqrt_tabs.p = nullptr; return fail(JJ_ERR_NOMEM); }
  c->sqrt_tabs.cap = 65536 + 4 * 256 * NL * 4;
  if (hipMemsetAsync(c->sqrt_tabs.p, 0, c->sqrt_tabs.cap, c->stream) != hipSuccess) return fail(JJ_ERR_HIP);
  c->sqrt_tables.dlog = (const uint8_t*)c->sqrt_tabs.p;
  c->sqrt_tables.npow = (const u32*)((uint8_t*)c->sqrt_tabs.p + 65536);
  hipLaunchKernelGGL(k_sqrt_tables_init, dim3(5), dim3(256), 0, c->stream, (uint8_t*)c->sqrt_tabs.p, (u32*)((uint8_t*)c->sqrt_tabs.p + 65536));
  if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(JJ_ERR_HIP);
  *out = c;
  return JJ_OK;
}
JJ_API int jj_ctx_destroy(jj_ctx* c) {
  if (!c) return JJ_ERR_INVALID;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  DevBuf* all[] = {&c->in[0], &c->in[1], &c->in[2], &c->in[3], &c->out[0], &c->out[1], &c->okb, &c->ws0.ext, &c->ws0.scratch, &c->ws0.tables,
                   &c->ws_tmp[0], &c->ws_tmp[1], &c->ws_tmp[2], &c->ws_tmp[3], &c->sqrt_tabs, &c->ws0.cursor,
                   &c->pipe.wset.ext, &c->pipe.wset.scratch, &c->pipe.wset.tables, &c->pipe.wset.cursor};
  for (jj_msm_job* j : c->job_pool) { if (j->host) (void)hipHostFree(j->host); (void)hipEventDestroy(j->ev); delete j; }
  for (MsmLane& L : c->lanes) {
    if (L.owned) (void)hipStreamSynchronize(L.stream);
    DevBuf* lb[] = {&L.buf[0], &L.buf[1], &L.buf[2], &L.buf[3], &L.buf[4], &L.buf[5], &L.buf[6], &L.buf[7], &L.ctl, &L.bigpart, &L.seg, &L.rec};
    for (DevBuf* b : lb) if (b->p) (void)hipFree(b->p);
    if (L.owned) {
      (void)hipEventDestroy(L.ready_ev);
      (void)hipStreamDestroy(L.stream);
    }
  }
  for (DevBuf* b : all) if (b->p) (void)hipFree(b->p);
  delete c->copy_pool;
  for (int i = 0; i < 3; i++) { if (c->stage_in[i]) (void)hipHostFree(c->stage_in[i]); if (c->stage_out[i]) (void)hipHostFree(c->stage_out[i]); if (c->ev_stage[i]) (void)hipEventDestroy(c->ev_stage[i]); }
  if (c->gather_dev.p) (void)hipFree(c->gather_dev.p);
  if (c->gather_host) (void)hipHostFree(c->gather_host);
  if (c->pipe.ready) {
    for (int i = 0; i < 2; i++) {
      (void)hipEventDestroy(c->pipe.ev_in[i]); (void)hipEventDestroy(c->pipe.ev_done[i]); (void)hipEventDestroy(c->pipe.ev_out[i]);
      if (c->pipe.din[i].p) (void)hipFree(c->pipe.din[i].p);
      if (c->pipe.dout[i].p) (void)hipFree(c->pipe.dout[i].p);
    }
    (void)hipEventDestroy(c->pipe.ev_start); (void)hipEventDestroy(c->pipe.ev_tail);
    (void)hipStreamDestroy(c->pipe.h2d); (void)hipStreamDestroy(c->pipe.d2h); if (c->pipe.cs[0]) (void)hipStreamDestroy(c->pipe.cs[0]); if (c->pipe.cs[1]) (void)hipStreamDestroy(c->pipe.cs[1]);
  }
  for (auto& r : c->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); (void)hipEventDestroy(r.e2); }
  if (c->order_ev) (void)hipEventDestroy(c->order_ev);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return JJ_OK;
}
// Every call reuses the context's workspaces (window tables, extended SoA, staging buffers), so work queued on the
// previous launch stream must finish before work on a new one may touch them: the new stream waits on an event recorded
// on the old one (device-side ordering, no host synchronisation).
// A caller-owned stream must outlive its selection (include/jubjub_hip.h).  If it has been destroyed all the same, the record
// on it fails: the error is cleared, the device is drained instead (nothing of the old stream can still be in flight after
// that), and the context still moves to the new stream -- it must never stay stuck on a dead one.
static int switch_stream(jj_ctx* c, hipStream_t s) {
  JJ_ENTER(c);
  if (s == c->stream) return JJ_OK;
  bool ordered = hipEventRecord(c->order_ev, c->stream) == hipSuccess && hipStreamWaitEvent(s, c->order_ev, 0) == hipSuccess;
  if (!ordered) {
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
  }
  c->stream = s;
  return JJ_OK;
}
JJ_API int jj_ctx_set_stream(jj_ctx* c, void* s) {
  if (!c) return JJ_ERR_INVALID;
  return switch_stream(c, (hipStream_t)s);            // NULL is HIP's default (null) stream — e.g. torch's default stream
}
JJ_API int jj_ctx_use_own_stream(jj_ctx* c) {
  if (!c) return JJ_ERR_INVALID;
  return switch_stream(c, c->own_stream);
}
JJ_API int jj_ctx_sync(jj_ctx* c) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return JJ_OK;
}
JJ_API const char* jj_last_error(jj_ctx* c) { return c ? c->err.c_str() : "null context"; }
JJ_API int jj_device_info(jj_ctx* c, int64_t out[4]) {
  if (!c || !out) return JJ_ERR_INVALID;
  out[0] = c->cus; out[1] = c->clock_khz; out[2] = c->wave; out[3] = 0;
  return JJ_OK;
}

JJ_API int jj_ctx_profile(jj_ctx* c, int enable) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  c->profile = enable != 0;
  c->rec_used = 0;
  return JJ_OK;
}
// Returns up to `max` (main_ms, tail_ms) pairs recorded since jj_ctx_profile(ctx, 1) and resets the log.
JJ_API int jj_ctx_profile_read(jj_ctx* c, int max, float* main_ms, float* tail_ms, int* count) {
  if (!c || !count) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int k = 0;
  for (size_t i = 0; i < c->rec_used && k < max; i++, k++) {
    float a = 0, b = 0;
    HIPCHK(c, hipEventElapsedTime(&a, c->recs[i].e0, c->recs[i].e1));
    HIPCHK(c, hipEventElapsedTime(&b, c->recs[i].e1, c->recs[i].e2));
    if (main_ms) main_ms[k] = a;
    if (tail_ms) tail_ms[k] = b;
  }
  *count = k;
  c->rec_used = 0;
  return JJ_OK;
}
// Measured integer-VALU roofline denominator: sustained v_mad_u64_u32 lane-operations per second on this device.
// `count` timed launches after one warm-up launch, each ~1.5 ms of 8 independent multiply-add chains per lane on every SIMD; the
// clock the part sustains moves by a few percent with temperature and with what ran just before, so callers report the median with
// its spread (bench.py: before and after the workload) instead of one best value.
JJ_API int jj_peak_imad32_samples(jj_ctx* c, int count, double* out_per_sec) {
  if (!c || !out_per_sec || count < 1 || count > 64) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  int rc = ensure(c, c->ws_tmp[0], (size_t)c->cus * 8 * 256 * 4); if (rc) return rc;
  const int iters = 4000, blocks = c->cus * 8;
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
  hipLaunchKernelGGL(k_peak_mad, dim3(blocks), dim3(256), 0, c->stream, (u32*)c->ws_tmp[0].p, iters, 12345u);     // warm-up: clocks ramp
  for (int rep = 0; rep < count; rep++) {
    HIPCHK(c, hipEventRecord(e0, c->stream));
    hipLaunchKernelGGL(k_peak_mad, dim3(blocks), dim3(256), 0, c->stream, (u32*)c->ws_tmp[0].p, iters, 12345u);
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)iters * 64.0 /* mads per iteration */ * 256.0 * blocks;
    out_per_sec[rep] = ops / (ms * 1e-3);
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return JJ_OK;
}
// the median of five samples
JJ_API int jj_peak_imad32(jj_ctx* c, double* out_per_sec) {
  if (!c || !out_per_sec) return JJ_ERR_INVALID;
  double v[5];
  const int rc = jj_peak_imad32_samples(c, 5, v); if (rc) return rc;
  std::sort(v, v + 5);
  *out_per_sec = v[2];
  return JJ_OK;
}

// ---------------------------------------------------------------------------------------------------- fields
template <class P, int OP>
static int field_op(jj_ctx* c, size_t n, const void* a, const void* b, void* out, uint8_t* ok, bool want_ok) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  const size_t in_bytes = (OP == OP_FROM_WIDE ? 64 : 32) * n;
  const void *da = nullptr, *db = nullptr;
  int rc;
  if ((rc = stage_in(c, 0, a, in_bytes, &da))) return rc;
  const bool binary = (OP == OP_ADD || OP == OP_SUB || OP == OP_MUL);
  if (binary && (rc = stage_in(c, 1, b, 32 * n, &db))) return rc;
  OutRef o, ok_o; ok_o.host = false; ok_o.dev = nullptr;
  if ((rc = stage_out(c, c->out[0], out, 32 * n, &o))) return rc;
  if (want_ok && (rc = stage_out(c, c->okb, ok, n, &ok_o))) return rc;
  if (n) hipLaunchKernelGGL((k_field_op<P, OP>), dim3(blocks_for(n)), dim3(256), 0, c->stream, n, da, db, o.dev, (uint8_t*)ok_o.dev, c->sqrt_tables);
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  if (want_ok && (rc = finish_out(c, ok_o, &sync))) return rc;
  return finish(c, sync);
}
#define FIELD_BIN(name, P, OP) JJ_API int name(jj_ctx* c, size_t n, const void* a, const void* b, void* out) { return field_op<P, OP>(c, n, a, b, out, nullptr, false); }
#define FIELD_UN(name, P, OP) JJ_API int name(jj_ctx* c, size_t n, const void* a, void* out) { return field_op<P, OP>(c, n, a, nullptr, out, nullptr, false); }
#define FIELD_UN_OK(name, P, OP) JJ_API int name(jj_ctx* c, size_t n, const void* a, void* out, uint8_t* ok) { if (!ok && n) return JJ_ERR_INVALID; return field_op<P, OP>(c, n, a, nullptr, out, ok, true); }
FIELD_BIN(jj_fq_add, FqP, OP_ADD) FIELD_BIN(jj_fq_sub, FqP, OP_SUB) FIELD_BIN(jj_fq_mul, FqP, OP_MUL)
FIELD_UN(jj_fq_neg, FqP, OP_NEG) FIELD_UN(jj_fq_square, FqP, OP_SQUARE) FIELD_UN(jj_fq_double, FqP, OP_DOUBLE)
FIELD_UN_OK(jj_fq_invert, FqP, OP_INVERT) FIELD_UN_OK(jj_fq_sqrt, FqP, OP_SQRT) FIELD_UN_OK(jj_fq_from_bytes, FqP, OP_FROM_BYTES)
FIELD_UN(jj_fq_from_bytes_wide, FqP, OP_FROM_WIDE)
FIELD_BIN(jj_fr_add, FrP, OP_ADD) FIELD_BIN(jj_fr_sub, FrP, OP_SUB) FIELD_BIN(jj_fr_mul, FrP, OP_MUL)
FIELD_UN(jj_fr_neg, FrP, OP_NEG) FIELD_UN(jj_fr_square, FrP, OP_SQUARE) FIELD_UN(jj_fr_double, FrP, OP_DOUBLE)
FIELD_UN_OK(jj_fr_invert, FrP, OP_INVERT) FIELD_UN_OK(jj_fr_sqrt, FrP, OP_SQRT) FIELD_UN_OK(jj_fr_from_bytes, FrP, OP_FROM_BYTES)
FIELD_UN(jj_fr_from_bytes_wide, FrP, OP_FROM_WIDE)

template <class P>
static int field_pow(jj_ctx* c, size_t n, const void* a, const void* e, void* out) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  const void *da, *de; int rc; OutRef o;
  if ((rc = stage_in(c, 0, a, 32 * n, &da))) return rc;
  if ((rc = stage_in(c, 1, e, 32 * n, &de))) return rc;
  if ((rc = stage_out(c, c->out[0], out, 32 * n, &o))) return rc;
  if (n) hipLaunchKernelGGL((k_field_pow<P>), dim3(blocks_for(n)), dim3(256), 0, c->stream, n, da, de, o.dev);
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
JJ_API int jj_fq_pow(jj_ctx* c, size_t n, const void* a, const void* exp32, void* out) { return field_pow<FqP>(c, n, a, exp32, out); }
JJ_API int jj_fr_pow(jj_ctx* c, size_t n, const void* a, const void* exp32, void* out) { return field_pow<FrP>(c, n, a, exp32, out); }

template <class P>
static int field_to_bits(jj_ctx* c, size_t n, const void* a, void* out256) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  const void* da; int rc; OutRef o;
  if ((rc = stage_in(c, 0, a, 32 * n, &da))) return rc;
  if ((rc = stage_out(c, c->out[0], out256, 256 * n, &o))) return rc;
  if (n) hipLaunchKernelGGL((k_field_to_bits<P>), dim3(blocks_for(n)), dim3(256), 0, c->stream, n, da, o.dev);
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
JJ_API int jj_fq_to_le_bits(jj_ctx* c, size_t n, const void* a, void* out256) { return field_to_bits<FqP>(c, n, a, out256); }
JJ_API int jj_fr_to_le_bits(jj_ctx* c, size_t n, const void* a, void* out256) { return field_to_bits<FrP>(c, n, a, out256); }
// PrimeFieldBits::char_le_bits (reference src/fr.rs:775-785): the modulus, same layout; host-only
JJ_API int jj_fr_char_le_bits(uint8_t out256[256]) {
  if (!out256) return JJ_ERR_INVALID;
  for (int b = 0; b < 256; b++) out256[b] = (FR_MODULUS_BYTES[b >> 3] >> (b & 7)) & 1;
  return JJ_OK;
}

// ---------------------------------------------------------------------------------------------------- normalisation
static SoA soa_of(DevBuf& b, size_t n) { SoA s; s.base = (u32*)b.p; s.n = n; return s; }
static int ensure_ext(jj_ctx* c, size_t n, int coords) { return ensure(c, c->ws->ext, (size_t)coords * NL * 4 * std::max(n, (size_t)1)); }

// ext SoA (coords 0..2) -> affine 64 B (mode 0) or compressed 32 B (mode 1) at device pointer dout
static int normalize_launch(jj_ctx* c, size_t n, SoA ext, void* dout, int mode) {
  if (!n) return JJ_OK;
  int rc = ensure(c, c->ws->scratch, (size_t)NL * 4 * n); if (rc) return rc;
  SoA scratch = soa_of(c->ws->scratch, n);
  // chunk length: amortise the ~330-multiplication inversion, but keep >= ~8 waves per CU in flight (and two rounds of them: a 64-point
  // chunk at 2^23 units loses more to the single-round tail than the shared inversion returns, measured on the decoder)
  const size_t lanes_wanted = (size_t)c->cus * 64 * 8;
  if (n >= lanes_wanted * 128) { size_t T = (n + 63) / 64; hipLaunchKernelGGL((k_normalize<64>), dim3(blocks_for(T)), dim3(256), 0, c->stream, n, T, ext, scratch, dout, mode); }   // 2^24 units: -17 % (1.59 -> 1.32 ms)
  else if (n >= lanes_wanted * 32) { size_t T = (n + 31) / 32; hipLaunchKernelGGL((k_normalize<32>), dim3(blocks_for(T)), dim3(256), 0, c->stream, n, T, ext, scratch, dout, mode); }
  else if (n >= lanes_wanted * 4) { size_t T = (n + 15) / 16; hipLaunchKernelGGL((k_normalize<16>), dim3(blocks_for(T)), dim3(256), 0, c->stream, n, T, ext, scratch, dout, mode); }
  else { size_t T = (n + 3) / 4; hipLaunchKernelGGL((k_normalize<4>), dim3(blocks_for(T)), dim3(256), 0, c->stream, n, T, ext, scratch, dout, mode); }
  return JJ_OK;
}

// ---------------------------------------------------------------------------------------------------- point ops
template <int OP>
static int point_op(jj_ctx* c, size_t n, const void* p, const void* q, void* out, size_t out_elem) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  const void *dp = nullptr, *dq = nullptr;
  int rc;
  if ((rc = stage_in(c, 0, p, 64 * n, &dp))) return rc;
  if ((OP == PT_ADD || OP == PT_SUB) && (rc = stage_in(c, 1, q, 64 * n, &dq))) return rc;
  OutRef o;
  if ((rc = stage_out(c, c->out[0], out, out_elem * n, &o))) return rc;
  if ((rc = ensure_ext(c, n, 3))) return rc;
  SoA ext = soa_of(c->ws->ext, n);
  if (n) {
    hipLaunchKernelGGL((k_point_op<OP>), dim3(blocks_for(n)), dim3(256), 0, c->stream, n, dp, dq, ext, o.dev);
    if (OP <= PT_COFACTOR && (rc = normalize_launch(c, n, ext, o.dev, 0))) return rc;
  }
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
JJ_API int jj_point_double(jj_ctx* c, size_t n, const void* p, void* out) { return point_op<PT_DOUBLE>(c, n, p, nullptr, out, 64); }
JJ_API int jj_point_add(jj_ctx* c, size_t n, const void* p, const void* q, void* out) { return point_op<PT_ADD>(c, n, p, q, out, 64); }
JJ_API int jj_point_sub(jj_ctx* c, size_t n, const void* p, const void* q, void* out) { return point_op<PT_SUB>(c, n, p, q, out, 64); }
JJ_API int jj_point_neg(jj_ctx* c, size_t n, const void* p, void* out) { return point_op<PT_NEG>(c, n, p, nullptr, out, 64); }
JJ_API int jj_point_mul_by_cofactor(jj_ctx* c, size_t n, const void* p, void* out) { return point_op<PT_COFACTOR>(c, n, p, nullptr, out, 64); }
JJ_API int jj_point_to_niels(jj_ctx* c, size_t n, const void* p, void* out96) { return point_op<PT_TO_NIELS>(c, n, p, nullptr, out96, 96); }
JJ_API int jj_is_identity(jj_ctx* c, size_t n, const void* p, uint8_t* out) { return point_op<PT_IS_IDENTITY>(c, n, p, nullptr, out, 1); }
JJ_API int jj_is_small_order(jj_ctx* c, size_t n, const void* p, uint8_t* out) { return point_op<PT_IS_SMALL_ORDER>(c, n, p, nullptr, out, 1); }
JJ_API int jj_is_on_curve(jj_ctx* c, size_t n, const void* p, uint8_t* out) { return point_op<PT_IS_ON_CURVE>(c, n, p, nullptr, out, 1); }

// ---------------------------------------------------------------------------------------------------- var-base
// launch geometry of the windowed ladder: persistent grid, one 2448-byte table slot (17 entries x 144 B) per lane
static void varbase_geometry(jj_ctx* c, size_t n, unsigned* blocks, size_t* threads) {
  const size_t max_threads = (size_t)c->cus * 256 * c->vb_blocks_per_cu;   // k blocks of 256 per CU = k waves / SIMD
  size_t t = std::min(max_threads, ((n + 255) / 256) * 256);
  if (t == 0) t = 256;
  *blocks = (unsigned)(t / 256); *threads = t;
}
static int varbase_to_ext(jj_ctx* c, size_t n, const void* ds, const void* dp, SoA ext, bool five, bool shared_scalar = false) {
  if (n <= (size_t)c->vb_quad_max && !shared_scalar) {      // small batch: one scalar multiplication per quad of lanes (3x lower latency)
    int rc = ensure(c, c->ws->tables, n * (size_t)(VB_SLOTS * ENIELS_WORDS) * 4); if (rc) return rc;
    if (five) hipLaunchKernelGGL(k_varbase_quad<true>, dim3(blocks_for(4 * n)), dim3(256), 0, c->stream, n, ds, dp, (u32*)c->ws->tables.p, ext);
    else hipLaunchKernelGGL(k_varbase_quad<false>, dim3(blocks_for(4 * n)), dim3(256), 0, c->stream, n, ds, dp, (u32*)c->ws->tables.p, ext);
    return JJ_OK;
  }
  unsigned blocks; size_t threads;
  varbase_geometry(c, n, &blocks, &threads);
  int rc = ensure(c, c->ws->tables, threads * (size_t)(VB_SLOTS * ENIELS_WORDS) * 4); if (rc) return rc;
  if ((rc = ensure(c, c->ws->cursor, 64))) return rc;
  HIPCHK(c, hipMemsetAsync(c->ws->cursor.p, 0, 8, c->stream));          // the waves' work cursor
  if (shared_scalar) hipLaunchKernelGGL((k_varbase<false, true>), dim3(blocks), dim3(256), 0, c->stream, n, ds, dp, (u32*)c->ws->tables.p, ext, (unsigned long long*)c->ws->cursor.p);
  else if (five) hipLaunchKernelGGL((k_varbase<true, false>), dim3(blocks), dim3(256), 0, c->stream, n, ds, dp, (u32*)c->ws->tables.p, ext, (unsigned long long*)c->ws->cursor.p);
  else hipLaunchKernelGGL((k_varbase<false, false>), dim3(blocks), dim3(256), 0, c->stream, n, ds, dp, (u32*)c->ws->tables.p, ext, (unsigned long long*)c->ws->cursor.p);
  return JJ_OK;
}
static int varbase_api(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out, int mode) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  if (const size_t ch = pipe_chunk_for(c, n, 18); ch && all_host({scalars, points, out})) {
    const HostIn in[2] = {{scalars, 32}, {points, 64}};
    const HostOut ho[1] = {{out, (size_t)(mode ? 32 : 64)}};
    const int prc = run_pipelined(c, n, ch, in, ho, [&](size_t cn, const void* const* di, void* const* dout) -> int {
      int rc2;
      if ((rc2 = ensure_ext(c, cn, 3))) return rc2;
      SoA ext = soa_of(c->ws->ext, cn);
      if ((rc2 = varbase_to_ext(c, cn, di[0], di[1], ext, false))) return rc2;
      if ((rc2 = pipe_to_tail(c))) return rc2;
      return normalize_launch(c, cn, ext, dout[0], mode);
    });
    if (prc <= 0) return prc;      // +1: buffers could not be page-locked -> plain staging below
  }
  const void *ds, *dp; int rc; OutRef o;
  if ((rc = stage_in(c, 0, scalars, 32 * n, &ds))) return rc;
  if ((rc = stage_in(c, 1, points, 64 * n, &dp))) return rc;
  if ((rc = stage_out(c, c->out[0], out, (mode ? 32 : 64) * n, &o))) return rc;
  if ((rc = ensure_ext(c, n, 3))) return rc;
  SoA ext = soa_of(c->ws->ext, n);
  if (n) {
    prof_mark(c, 0);
    if ((rc = varbase_to_ext(c, n, ds, dp, ext, false))) return rc;
    prof_mark(c, 1);
    if ((rc = normalize_launch(c, n, ext, o.dev, mode))) return rc;
    prof_mark(c, 2);
  }
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
JJ_API int jj_varbase_mul(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out) { return varbase_api(c, n, scalars, points, out, 0); }
JJ_API int jj_varbase_mul_compressed(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out32) { return varbase_api(c, n, scalars, points, out32, 1); }
// one scalar, many bases (group::Wnaf's `scalar(..).base(..)` reuse pattern): the ladder reads the one scalar through a
// wave-uniform address (k_varbase<.., SHARED>): recoding and window digits are scalar-unit work, nothing is broadcast.
// Small batches use the quad kernel on a broadcast copy (latency path).
JJ_API int jj_varbase_mul_scalar(jj_ctx* c, size_t n, const void* scalar32, const void* points, void* out) {
  if (!c || !scalar32) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  const void* dp; int rc; OutRef o;
  if ((rc = stage_in(c, 1, points, 64 * n, &dp))) return rc;
  if ((rc = stage_out(c, c->out[0], out, 64 * n, &o))) return rc;
  if ((rc = ensure(c, c->ws_tmp[1], 32))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->ws_tmp[1].p, scalar32, 32, is_device_ptr(scalar32) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
  if ((rc = ensure_ext(c, n, 3))) return rc;
  SoA ext = soa_of(c->ws->ext, n);
  if (n) {
    if (n <= (size_t)c->vb_quad_max) {
      if ((rc = ensure(c, c->ws_tmp[0], 32 * n))) return rc;
      hipLaunchKernelGGL(k_fill_scalar, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, c->ws_tmp[0].p, (const uint8_t*)c->ws_tmp[1].p);
      if ((rc = varbase_to_ext(c, n, c->ws_tmp[0].p, dp, ext, false))) return rc;
    } else if ((rc = varbase_to_ext(c, n, c->ws_tmp[1].p, dp, ext, false, true))) return rc;
    if ((rc = normalize_launch(c, n, ext, o.dev, 0))) return rc;
  }
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
// constant-time ladder: table {P, 2P} in registers, signed 2-bit windows, mask selects (k_varbase_ct)
JJ_API int jj_varbase_mul_ct(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  const void *ds, *dp; int rc; OutRef o;
  if ((rc = stage_in(c, 0, scalars, 32 * n, &ds))) return rc;
  if ((rc = stage_in(c, 1, points, 64 * n, &dp))) return rc;
  if ((rc = stage_out(c, c->out[0], out, 64 * n, &o))) return rc;
  if ((rc = ensure_ext(c, n, 3))) return rc;
  SoA ext = soa_of(c->ws->ext, n);
  if (n) {
    prof_mark(c, 0);
    hipLaunchKernelGGL(k_varbase_ct, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, ds, dp, ext);
    prof_mark(c, 1);
    if ((rc = normalize_launch(c, n, ext, o.dev, 0))) return rc;
    prof_mark(c, 2);
  }
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
JJ_API int jj_varbase_mul_exact(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out160) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  const void *ds, *dp; int rc; OutRef o;
  if ((rc = stage_in(c, 0, scalars, 32 * n, &ds))) return rc;
  if ((rc = stage_in(c, 1, points, 64 * n, &dp))) return rc;
  if ((rc = stage_out(c, c->out[0], out160, 160 * n, &o))) return rc;
  if (n) hipLaunchKernelGGL(k_varbase_exact, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, ds, dp, o.dev);
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}

// [r]P == O for affine device points -> ok bytes (combine: 0 set, 1 and).  Default: order-8 Tate pairing
// (k_torsion_free); JJ_TORSION_CHECK=ladder runs the reference's definition, a var-base multiplication by r.
static int torsion_free_dev(jj_ctx* c, size_t n, const void* dpts, uint8_t* dok, int combine) {
  int rc;
  if (!c->torsion_ladder) {
    hipLaunchKernelGGL(k_torsion_free, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, dpts, dok, combine);
    return JJ_OK;
  }
  if ((rc = ensure(c, c->ws_tmp[0], 32 * std::max(n, (size_t)1)))) return rc;
  if ((rc = ensure(c, c->ws_tmp[1], 32))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->ws_tmp[1].p, FR_MODULUS_BYTES, 32, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_fill_scalar, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, c->ws_tmp[0].p, (const uint8_t*)c->ws_tmp[1].p);
  if ((rc = ensure_ext(c, n, 3))) return rc;
  SoA ext = soa_of(c->ws->ext, n);
  if ((rc = varbase_to_ext(c, n, c->ws_tmp[0].p, dpts, ext, false))) return rc;
  hipLaunchKernelGGL(k_is_identity_ext, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, ext, dok, combine);
  return JJ_OK;
}
static int torsion_pred(jj_ctx* c, size_t n, const void* p, uint8_t* out, bool prime_order) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  const void* dp; int rc; OutRef o;
  if ((rc = stage_in(c, 0, p, 64 * n, &dp))) return rc;
  if ((rc = stage_out(c, c->okb, out, n, &o))) return rc;
  if (n) {
    if ((rc = torsion_free_dev(c, n, dp, (uint8_t*)o.dev, 0))) return rc;
    if (prime_order) {   // & !is_identity  (reference src/lib.rs:717-719)
      if ((rc = ensure(c, c->ws_tmp[2], n))) return rc;
      if ((rc = ensure_ext(c, n, 3))) return rc;
      hipLaunchKernelGGL((k_point_op<PT_IS_IDENTITY>), dim3(blocks_for(n)), dim3(256), 0, c->stream, n, dp, (const void*)nullptr, soa_of(c->ws->ext, n), c->ws_tmp[2].p);
      hipLaunchKernelGGL(k_and_bytes, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, (uint8_t*)o.dev, (const uint8_t*)c->ws_tmp[2].p, 1);
    }
  }
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
JJ_API int jj_is_torsion_free(jj_ctx* c, size_t n, const void* p, uint8_t* out) { return torsion_pred(c, n, p, out, false); }
JJ_API int jj_is_prime_order(jj_ctx* c, size_t n, const void* p, uint8_t* out) { return torsion_pred(c, n, p, out, true); }

// ---------------------------------------------------------------------------------------------------- fixed-base
// entries (i, j) = j * 2^(w i) * B for i < W, j < E (j = 0: the identity), built on the GPU in two var-base passes
// (Q_i = 2^(w i) B, then (j+1) Q_i) so that no scalar ever reaches bit 252, which the ladder ignores.
static int build_window_table(jj_ctx* c, const uint8_t base[64], int w, int W, u32 E, size_t extra_top_entry, u32** out_dev, size_t* out_entries) {
  std::vector<uint8_t> s1((size_t)W * 32, 0), p1((size_t)W * 64), q((size_t)W * 64);
  for (int i = 0; i < W; i++) {
    const int bit = w * i;
    if (bit < 252) s1[(size_t)i * 32 + (bit >> 3)] = (uint8_t)(1u << (bit & 7));
    else { const int b2 = bit - 1; s1[(size_t)i * 32 + (b2 >> 3)] = (uint8_t)(1u << (b2 & 7)); }   // 2^(bit-1), doubled below
    memcpy(&p1[(size_t)i * 64], base, 64);
  }
  int rc = jj_varbase_mul(c, W, s1.data(), p1.data(), q.data()); if (rc) return rc;
  for (int i = 0; i < W; i++) if (w * i >= 252) { rc = jj_point_double(c, 1, &q[(size_t)i * 64], &q[(size_t)i * 64]); if (rc) return rc; }
  const size_t ne = (size_t)W * E + extra_top_entry;
  std::vector<uint8_t> s2(ne * 32, 0), p2(ne * 64), aff(ne * 64);
  for (size_t e = 0; e < (size_t)W * E; e++) {
    const size_t i = e / E; const u32 mult = (u32)(e % E);
    s2[e * 32] = (uint8_t)mult; s2[e * 32 + 1] = (uint8_t)(mult >> 8); s2[e * 32 + 2] = (uint8_t)(mult >> 16);
    memcpy(&p2[e * 64], &q[i * 64], 64);
  }
  rc = jj_varbase_mul(c, (size_t)W * E, s2.data(), p2.data(), aff.data()); if (rc) return rc;
  if (extra_top_entry) {                       // LDS layout: one extra entry 2^(w W) B = 2^w * Q_{W-1}
    uint8_t sc[32] = {0}; sc[w >> 3] = (uint8_t)(1u << (w & 7));
    rc = jj_varbase_mul(c, 1, sc, &q[(size_t)(W - 1) * 64], &aff[(size_t)W * E * 64]); if (rc) return rc;
  }
  u32* dev = nullptr;
  const int stride = extra_top_entry ? ANIELS_WORDS : GNIELS_WORDS;      // LDS-staged table: packed; gathered table: one line per entry
  if (hipMalloc((void**)&dev, ne * (size_t)stride * 4) != hipSuccess) { c->err = "hipMalloc(table) failed"; return JJ_ERR_NOMEM; }
  const void* dpts;
  if ((rc = stage_in(c, 0, aff.data(), ne * 64, &dpts))) { (void)hipFree(dev); return rc; }
  hipLaunchKernelGGL(k_affine_to_table, dim3(blocks_for(ne)), dim3(256), 0, c->stream, ne, dpts, dev, stride);
  rc = finish(c, true);
  if (rc) { (void)hipFree(dev); return rc; }
  *out_dev = dev; *out_entries = ne;
  return JJ_OK;
}
// Signed-comb table (layout of k_fixedbase_comb): 8 tables T_{j1}[idx] = 2^(4 j1) (2^224 + sum_{i<7} (2 idx_i - 1) 2^(32 i)) B of 128
// entries, then T_0 - B and T_0 + B.  Built on the GPU through the library's own entry points: Q_i = 2^(32 i) B and
// R_{j1,i} = 2^(4 j1) Q_i by the var-base ladder (no scalar reaches bit 252), then seven rounds of batched point additions.
static int build_comb_table(jj_ctx* c, const uint8_t base[64], u32** out_dev) {
  int rc;
  std::vector<uint8_t> s1((size_t)FBC_TEETH * 32, 0), p1((size_t)FBC_TEETH * 64), q((size_t)FBC_TEETH * 64);
  for (int i = 0; i < FBC_TEETH; i++) { const int bit = FBC_SPACING * i; s1[(size_t)i * 32 + (bit >> 3)] = (uint8_t)(1u << (bit & 7)); memcpy(&p1[(size_t)i * 64], base, 64); }
  if ((rc = jj_varbase_mul(c, FBC_TEETH, s1.data(), p1.data(), q.data()))) return rc;
  const size_t nr = (size_t)FBC_BLOCKS * FBC_TEETH;
  std::vector<uint8_t> s2(nr * 32, 0), p2(nr * 64), r(nr * 64), nrg(nr * 64);
  for (int j1 = 0; j1 < FBC_BLOCKS; j1++)
    for (int i = 0; i < FBC_TEETH; i++) {
      const size_t e = (size_t)j1 * FBC_TEETH + i; const int bit = FBC_COLS * j1;
      s2[e * 32 + (bit >> 3)] = (uint8_t)(1u << (bit & 7));
      memcpy(&p2[e * 64], &q[(size_t)i * 64], 64);
    }
  if ((rc = jj_varbase_mul(c, nr, s2.data(), p2.data(), r.data()))) return rc;
  if ((rc = jj_point_neg(c, nr, r.data(), nrg.data()))) return rc;
  const size_t ne = (size_t)FBC_BLOCKS * FBC_TENT;
  std::vector<uint8_t> acc(ne * 64), opnd(ne * 64), all((size_t)FBC_ENTRIES * 64);
  for (size_t e = 0; e < ne; e++) memcpy(&acc[e * 64], &r[((e / FBC_TENT) * FBC_TEETH + (FBC_TEETH - 1)) * 64], 64);     // the top tooth, always +
  for (int i = 0; i < FBC_TEETH - 1; i++) {
    for (size_t e = 0; e < ne; e++) {
      const size_t src = ((e / FBC_TENT) * FBC_TEETH + i) * 64;
      memcpy(&opnd[e * 64], ((e >> i) & 1) ? &r[src] : &nrg[src], 64);
    }
    if ((rc = jj_point_add(c, ne, acc.data(), opnd.data(), acc.data()))) return rc;
  }
  memcpy(all.data(), acc.data(), ne * 64);
  std::vector<uint8_t> b64((size_t)FBC_TENT * 64);
  for (int e = 0; e < FBC_TENT; e++) memcpy(&b64[(size_t)e * 64], base, 64);
  if ((rc = jj_point_sub(c, FBC_TENT, acc.data(), b64.data(), &all[ne * 64]))) return rc;                    // T_0 - B
  if ((rc = jj_point_add(c, FBC_TENT, acc.data(), b64.data(), &all[(ne + FBC_TENT) * 64]))) return rc;       // T_0 + B
  u32* dev = nullptr;
  if (hipMalloc((void**)&dev, (size_t)FBC_LDS_BYTES) != hipSuccess) { c->err = "hipMalloc(table) failed"; return JJ_ERR_NOMEM; }
  const void* dpts;
  if ((rc = stage_in(c, 0, all.data(), (size_t)FBC_ENTRIES * 64, &dpts))) { (void)hipFree(dev); return rc; }
  hipLaunchKernelGGL(k_affine_to_table, dim3(blocks_for(FBC_ENTRIES)), dim3(256), 0, c->stream, (size_t)FBC_ENTRIES, dpts, dev, ANIELS_WORDS);
  rc = finish(c, true);
  if (rc) { (void)hipFree(dev); return rc; }
  *out_dev = dev;
  return JJ_OK;
}
JJ_API int jj_fixedbase_table_create(jj_ctx* c, const void* base64, int window_bits, jj_table** out) {
  if (!c || !out || !base64) return JJ_ERR_INVALID;
  if (window_bits == 0) window_bits = c->fb_default_kind;
  if (window_bits != FB_W && window_bits != 7 && (window_bits < 8 || window_bits > 16)) { c->err = "window_bits must be 0 (default), 7 (signed comb in LDS), 6 (window table in LDS) or 8..16 (table gathered from L2 / Infinity Cache)"; return JJ_ERR_INVALID; }
  JJ_ENTER(c);
  uint8_t base[64];
  if (is_device_ptr(base64)) { HIPCHK(c, hipMemcpy(base, base64, 64, hipMemcpyDeviceToHost)); } else memcpy(base, base64, 64);
  jj_table* t = new jj_table();
  t->window_bits = window_bits;
  t->device = c->device;
  size_t ne = 0; int rc;
  if (window_bits == 7) {
    rc = build_comb_table(c, base, &t->dev);
  } else if (window_bits == FB_W) {
    // 42 windows x 32 entries + the carry entry 2^252 B  (layout of k_fixedbase)
    rc = build_window_table(c, base, FB_W, FB_NWIN, FB_ENT, 1, &t->dev, &ne);
  } else {
    FbParams& fp = t->fp;
    fp.w = window_bits; fp.W = (253 + window_bits - 1) / window_bits; fp.E = 1u << (window_bits - 1);
    memset(fp.recode, 0, sizeof fp.recode);
    for (int i = 0; i < fp.W - 1; i++) { const int bit = fp.w * i + fp.w - 1; fp.recode[bit >> 5] |= 1u << (bit & 31); }
    rc = build_window_table(c, base, fp.w, fp.W, fp.E + 1, 0, &t->dev, &ne);
  }
  if (rc) { delete t; return rc; }
  *out = t;
  return JJ_OK;
}
JJ_API int jj_fixedbase_table_destroy(jj_ctx* c, jj_table* t) {
  if (!c || !t) return JJ_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (t->dev) (void)hipFree(t->dev);
  delete t;
  return JJ_OK;
}
static int fixedbase_launch(jj_ctx* c, const jj_table* t, size_t n, const void* ds, SoA ext, int chain = 0) {
  // one workgroup per CU (the table fills the LDS)
  if (t->window_bits == 7) {
    const unsigned cblocks = (unsigned)std::min((size_t)c->cus, (n + FBC_THREADS - 1) / FBC_THREADS);
    if (c->fb_const_time) hipLaunchKernelGGL(k_fixedbase_comb<true>, dim3(cblocks), dim3(FBC_THREADS), FBC_LDS_BYTES, c->stream, n, ds, (const u32*)t->dev, ext, chain);
    else hipLaunchKernelGGL(k_fixedbase_comb<false>, dim3(cblocks), dim3(FBC_THREADS), FBC_LDS_BYTES, c->stream, n, ds, (const u32*)t->dev, ext, chain);
  } else if (t->window_bits != FB_W) {
    const unsigned gblocks = (unsigned)std::min((size_t)c->cus * c->fb_gather_blocks_per_cu, (n + 255) / 256);
    hipLaunchKernelGGL(k_fixedbase_gather, dim3(gblocks), dim3(256), 0, c->stream, n, ds, (const u32*)t->dev, t->fp, ext, chain);
  } else {
    const unsigned wblocks = (unsigned)std::min((size_t)c->cus, (n + FB_THREADS - 1) / FB_THREADS);
    if (c->fb_const_time) hipLaunchKernelGGL(k_fixedbase<true>, dim3(wblocks), dim3(FB_THREADS), FB_LDS_BYTES, c->stream, n, ds, (const u32*)t->dev, ext, chain);
    else hipLaunchKernelGGL(k_fixedbase<false>, dim3(wblocks), dim3(FB_THREADS), FB_LDS_BYTES, c->stream, n, ds, (const u32*)t->dev, ext, chain);
  }
  return JJ_OK;
}
static int fixedbase_api(jj_ctx* c, const jj_table* t, size_t n, const void* scalars, void* out, int mode) {
  if (!c || !t) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  if (t->device != c->device) { c->err = "fixed-base table belongs to another device"; return JJ_ERR_INVALID; }
  // lanes of the table's kernel: one workgroup per CU for the LDS tables, fb_gather_blocks_per_cu blocks of 256 for the gathered ones
  const size_t fb_lanes = t->window_bits == 7 ? (size_t)c->cus * FBC_THREADS : t->window_bits == FB_W ? (size_t)c->cus * FB_THREADS : (size_t)c->cus * c->fb_gather_blocks_per_cu * 256;
  if (const size_t ch = pipe_chunk_for(c, n, 20, fb_lanes); ch && all_host({scalars, out})) {
    const HostIn in[1] = {{scalars, 32}};
    const HostOut ho[1] = {{out, (size_t)(mode ? 32 : 64)}};
    const int prc = run_pipelined(c, n, ch, in, ho, [&](size_t cn, const void* const* di, void* const* dout) -> int {
      int rc2;
      if ((rc2 = ensure_ext(c, cn, 3))) return rc2;
      SoA ext = soa_of(c->ws->ext, cn);
      if ((rc2 = fixedbase_launch(c, t, cn, di[0], ext))) return rc2;
      if ((rc2 = pipe_to_tail(c))) return rc2;
      return normalize_launch(c, cn, ext, dout[0], mode);
    }, fb_lanes);
    if (prc <= 0) return prc;
  }
  const void* ds; int rc; OutRef o;
  if ((rc = stage_in(c, 0, scalars, 32 * n, &ds))) return rc;
  if ((rc = stage_out(c, c->out[0], out, (mode ? 32 : 64) * n, &o))) return rc;
  if ((rc = ensure_ext(c, n, 3))) return rc;
  SoA ext = soa_of(c->ws->ext, n);
  if (n) {
    prof_mark(c, 0);
    if ((rc = fixedbase_launch(c, t, n, ds, ext))) return rc;
    prof_mark(c, 1);
    if ((rc = normalize_launch(c, n, ext, o.dev, mode))) return rc;
    prof_mark(c, 2);
  }
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}

JJ_API int jj_fixedbase_mul(jj_ctx* c, const jj_table* t, size_t n, const void* scalars, void* out) { return fixedbase_api(c, t, n, scalars, out, 0); }
// out[i] = sum_j tables[j] * scalars[j * n + i]: the accumulator stays extended between the bases (one normalisation in all)
JJ_API int jj_fixedbase_multi_mul(jj_ctx* c, const jj_table* const* tables, int nbases, size_t n, const void* scalars, void* out64) {
  if (!c || !tables || nbases < 1) return JJ_ERR_INVALID;
  for (int j = 0; j < nbases; j++) if (!tables[j]) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  for (int j = 0; j < nbases; j++) if (tables[j]->device != c->device) { c->err = "fixed-base table belongs to another device"; return JJ_ERR_INVALID; }
  const void* ds; int rc; OutRef o;
  if ((rc = stage_in(c, 0, scalars, 32 * n * (size_t)nbases, &ds))) return rc;
  if ((rc = stage_out(c, c->out[0], out64, 64 * n, &o))) return rc;
  if ((rc = ensure_ext(c, n, 5))) return rc;
  SoA ext = soa_of(c->ws->ext, n);
  if (n) {
    prof_mark(c, 0);
    for (int j = 0; j < nbases; j++) {
      const int chain = (j > 0 ? 1 : 0) | (j + 1 < nbases ? 2 : 0);
      if ((rc = fixedbase_launch(c, tables[j], n, (const uint8_t*)ds + (size_t)j * n * 32, ext, chain))) return rc;
    }
    prof_mark(c, 1);
    if ((rc = normalize_launch(c, n, ext, o.dev, 0))) return rc;
    prof_mark(c, 2);
  }
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
// ---- several bases, short scalars, one pass (k_pack_composite + k_fixedbase on a composite table)
JJ_API int jj_fixedbase_composite_create(jj_ctx* c, int nbases, const void* bases64, const int* scalar_bits, jj_table** out) {
  if (!c || !out || !bases64 || !scalar_bits || nbases < 1 || nbases > FBX_MAX_BASES) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  jj_table* t = new jj_table();
  t->window_bits = FB_W;
  t->device = c->device;
  int slots = 0;
  for (int b = 0; b < nbases; b++) {
    if (scalar_bits[b] < 1 || scalar_bits[b] > 250) { c->err = "composite table: scalar_bits must be 1..250"; delete t; return JJ_ERR_INVALID; }
    t->fx.off[b] = slots; t->fx.bits[b] = scalar_bits[b];
    slots += (scalar_bits[b] + 2 + FB_W - 1) / FB_W;                 // 6 W >= bits + 2: the field's recoding never carries out of it
  }
  if (slots > FB_NWIN) { c->err = "composite table: the bases need more than 42 six-bit windows (sum of ceil((bits + 2) / 6))"; delete t; return JJ_ERR_INVALID; }
  t->fx.nb = nbases;
  std::vector<uint8_t> bases((size_t)nbases * 64);
  if (is_device_ptr(bases64)) {
    const hipError_t e = hipMemcpy(bases.data(), bases64, bases.size(), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { c->err = std::string("hipMemcpy(bases) failed: ") + hipGetErrorString(e); delete t; return JJ_ERR_HIP; }
  } else memcpy(bases.data(), bases64, bases.size());
  // Q_s = 64^(local window) B_b for every slot in use, then j Q_s for j = 0 .. 32; unused slots and the carry entry hold the identity
  int rc;
  std::vector<uint8_t> s1((size_t)slots * 32, 0), p1((size_t)slots * 64), q((size_t)slots * 64);
  for (int b = 0, sl = 0; b < nbases; b++) {
    const int W = (b + 1 < nbases ? t->fx.off[b + 1] : slots) - t->fx.off[b];
    for (int i = 0; i < W; i++, sl++) { const int bit = FB_W * i; s1[(size_t)sl * 32 + (bit >> 3)] = (uint8_t)(1u << (bit & 7)); memcpy(&p1[(size_t)sl * 64], &bases[(size_t)b * 64], 64); }
  }
  if ((rc = jj_varbase_mul(c, slots, s1.data(), p1.data(), q.data()))) { delete t; return rc; }
  const size_t ne = (size_t)slots * FB_ENT;
  std::vector<uint8_t> s2(ne * 32, 0), p2(ne * 64), aff((size_t)FB_ENTRIES * 64, 0);
  for (size_t e = 0; e < ne; e++) { s2[e * 32] = (uint8_t)(e % FB_ENT); memcpy(&p2[e * 64], &q[(e / FB_ENT) * 64], 64); }
  for (size_t e = 0; e < (size_t)FB_ENTRIES; e++) aff[e * 64 + 32] = 1;                    // affine identity (0, 1)
  if ((rc = jj_varbase_mul(c, ne, s2.data(), p2.data(), aff.data()))) { delete t; return rc; }
  for (size_t e = ne; e < (size_t)FB_ENTRIES; e++) { memset(&aff[e * 64], 0, 64); aff[e * 64 + 32] = 1; }
  if (hipMalloc((void**)&t->dev, (size_t)FB_LDS_BYTES) != hipSuccess) { c->err = "hipMalloc(table) failed"; delete t; return JJ_ERR_NOMEM; }
  const void* dpts;
  if ((rc = stage_in(c, 0, aff.data(), (size_t)FB_ENTRIES * 64, &dpts))) { (void)hipFree(t->dev); delete t; return rc; }
  hipLaunchKernelGGL(k_affine_to_table, dim3(blocks_for(FB_ENTRIES)), dim3(256), 0, c->stream, (size_t)FB_ENTRIES, dpts, t->dev, ANIELS_WORDS);
  rc = finish(c, true);
  if (rc) { (void)hipFree(t->dev); delete t; return rc; }
  *out = t;
  return JJ_OK;
}
JJ_API int jj_fixedbase_composite_mul(jj_ctx* c, const jj_table* t, size_t n, const void* scalars, void* out64) {
  if (!c || !t || t->fx.nb < 1) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  if (t->device != c->device) { c->err = "fixed-base table belongs to another device"; return JJ_ERR_INVALID; }
  const void* ds; int rc; OutRef o;
  if ((rc = stage_in(c, 0, scalars, 32 * n * (size_t)t->fx.nb, &ds))) return rc;
  if ((rc = stage_out(c, c->out[0], out64, 64 * n, &o))) return rc;
  if ((rc = ensure_ext(c, n, 3))) return rc;
  if ((rc = ensure(c, c->ws_tmp[2], 32 * std::max<size_t>(n, 1)))) return rc;
  SoA ext = soa_of(c->ws->ext, n);
  if (n) {
    prof_mark(c, 0);
    hipLaunchKernelGGL(k_pack_composite, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, ds, t->fx, c->ws_tmp[2].p);
    if ((rc = fixedbase_launch(c, t, n, c->ws_tmp[2].p, ext))) return rc;
    prof_mark(c, 1);
    if ((rc = normalize_launch(c, n, ext, o.dev, 0))) return rc;
    prof_mark(c, 2);
  }
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
JJ_API int jj_fixedbase_mul_compressed(jj_ctx* c, const jj_table* t, size_t n, const void* scalars, void* out32) { return fixedbase_api(c, t, n, scalars, out32, 1); }

// ---------------------------------------------------------------------------------------------------- sums / MSM
// folds a 5-coordinate SoA of n extended points down to one, result left in (U,V,Z) coords of the returned SoA
static int sum_reduce(jj_ctx* c, size_t n, DevBuf* a, DevBuf* b, SoA* result) {
  constexpr int FOLD = 32;
  DevBuf* cur = a; DevBuf* nxt = b;
  size_t m = n;
  while (m > 1) {
    const size_t T = (m + FOLD - 1) / FOLD;
    int rc = ensure(c, *nxt, (size_t)5 * NL * 4 * T); if (rc) return rc;
    hipLaunchKernelGGL((k_sum_pass<FOLD>), dim3(blocks_for(T)), dim3(256), 0, c->stream, m, T, soa_of(*cur, m), soa_of(*nxt, T));
    std::swap(cur, nxt);
    m = T;
  }
  *result = soa_of(*cur, 1);
  return JJ_OK;
}
static const uint8_t AFFINE_IDENTITY_BYTES[64] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1};
static int write_identity(jj_ctx* c, const OutRef& o) {
  HIPCHK(c, hipMemcpyAsync(o.dev, AFFINE_IDENTITY_BYTES, 64, hipMemcpyHostToDevice, c->stream));
  return JJ_OK;
}
JJ_API int jj_point_sum(jj_ctx* c, size_t n, const void* p, void* out64) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  int rc; OutRef o;
  if ((rc = stage_out(c, c->out[0], out64, 64, &o))) return rc;
  if (n == 0) { if ((rc = write_identity(c, o))) return rc; }
  else {
    const void* dp;
    if ((rc = stage_in(c, 0, p, 64 * n, &dp))) return rc;
    if ((rc = ensure(c, c->ws_tmp[2], (size_t)5 * NL * 4 * n))) return rc;
    hipLaunchKernelGGL(k_affine_to_soa5, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, dp, soa_of(c->ws_tmp[2], n));
    SoA res;
    if ((rc = sum_reduce(c, n, &c->ws_tmp[2], &c->ws_tmp[3], &res))) return rc;
    if ((rc = normalize_launch(c, 1, res, o.dev, 0))) return rc;
  }
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
// MSM on the device up to the record of partial window sums (jj_msm_kernels.h); everything here is launches only (the
// workspaces are grown first), so a pass can be queued behind another one without any host synchronisation in between.
//   part_w0 / part_stride: the pass owns windows part_w0, part_w0 + part_stride, ... (0 / 1: all of them)
//   rec_dev: MSM_REC bytes of device memory for the record
struct MsmGeometry { bool small; MsmParams mp; u32 nblk; };
static void msm_layout(MsmParams& mp, int W, int w0, int wstride) {
  mp.W = W; mp.c = 253 / W; mp.r = 253 % W;
  mp.w0 = w0; mp.wstride = wstride < 1 ? 1 : wstride;
  mp.Ws = w0 < W ? (W - w0 + mp.wstride - 1) / mp.wstride : 0;
  mp.B = 1u << (mp.c + (mp.r ? 1 : 0) - 1);
  memset(mp.recode, 0, sizeof mp.recode);
  int bit = 0;
  for (int w = 0; w < W - 1; w++) { const int width = mp.c + (w < mp.r ? 1 : 0); const int b = bit + width - 1; mp.recode[b >> 5] |= 1u << (b & 31); bit += width; }
}
// number of windows for n terms (measured on MI355X; JJ_MSM_WINDOWS overrides): the windows tile the 253 scalar bits exactly, so
// any W is as good as its entry count n W and its bucket count ~ W 2^(253/W - 1) make it
// from this many terms the large-input configuration (17 / 16 windows, length-sorted segments) is faster than 23 windows + chunks + fix-up
// (experiments/misc/msm_crossover.py: 131 072 terms 0.438 against 0.448 ms, 150 000 terms 0.485 against 0.459 ms, 235 000 terms 0.727 against 0.533 ms)
constexpr size_t MSM_LARGE_MIN = (size_t)9 << 14;
static int msm_windows_for(jj_ctx* c, size_t n) {
  if (c->msm_windows >= MSM_WINDOWS_MIN && c->msm_windows <= MSM_WINDOWS_MAX) return c->msm_windows;
  // measured (experiments/misc/msm_sweep*.sh, profiles/r3_msm_window_sweep.txt, r3_msm_reduce_grid.txt): 16 windows (13 of 16 bits, 3 of
  // 15) from 2^20 terms; 17 windows (15 of 15 bits, 2 of 14: half the buckets, so the bucket reduce is 145 us instead of 210, for
  // 6 % more additions) from MSM_LARGE_MIN terms; below, 23 windows of 11 bits: wider windows cut the additions but their buckets (4096+ per
  // window) make the fix-up and reduce chains longer than the additions they save
  return n >= ((size_t)1 << 20) ? 16 : n >= MSM_LARGE_MIN ? 17 : 23;
}
// counters (MSM_COUNTER_WORDS words, cleared by the first kernel of a pass) | big-bucket work list | workgroup partial sums
constexpr size_t MSM_BIG_OFF = 512, MSM_PART_OFF = MSM_BIG_OFF + sizeof(BigBucket) * FIXUP_BIG_MAX;
static int msm_ensure_ctl(jj_ctx* c, MsmLane& L) { return ensure(c, L.ctl, MSM_PART_OFF + (size_t)64 * MSM_TREE_QUADS * MSM_PART_WORDS * 4); }
static int msm_enqueue_small(jj_ctx* c, MsmLane& L, size_t n, const void* ds, const void* dp, int part_w0, int part_stride, void* rec_dev) {
  MsmParams mp;
  msm_layout(mp, SM_W, part_w0, part_stride);
  int rc;
  if ((rc = ensure(c, L.buf[0], n * 32))) return rc;
  if ((rc = ensure(c, L.buf[1], n * (size_t)(SM_SLOTS * ENIELS_WORDS) * 4))) return rc;
  if ((rc = msm_ensure_ctl(c, L))) return rc;
  u32* counters = (u32*)L.ctl.p; u32* part = (u32*)((uint8_t*)L.ctl.p + MSM_PART_OFF);
  // workgroups of 64 quads per window: about 4 terms per quad, at most msm_small_blk (4 x 64 windows = one workgroup per CU)
  const u32 nblk = (u32)std::min<size_t>(c->msm_small_blk, std::max<size_t>(1, (n + 255) / 256));
  hipLaunchKernelGGL(k_msm_small_tables, dim3(blocks_for(4 * n)), dim3(256), 0, L.stream, n, ds, dp, mp, (u32*)L.buf[1].p, (u32*)L.buf[0].p, counters);
  hipLaunchKernelGGL(k_msm_small_sum, dim3(nblk, mp.Ws), dim3(4 * MSM_TREE_QUADS), 0, L.stream, n, mp, nblk, (const u32*)L.buf[1].p, (const u32*)L.buf[0].p, part, counters, (u32*)rec_dev);
  return JJ_OK;
}

static int msm_enqueue_pippenger(jj_ctx* c, MsmLane& ln, size_t n, const void* ds, const void* dp, int part_w0, int part_stride, void* rec_dev) {
  MsmParams mp;
  msm_layout(mp, msm_windows_for(c, n), part_w0, part_stride);
  const u32 B = mp.B, Ws = (u32)mp.Ws;
  const size_t nb = (size_t)Ws * B;
  // buckets per reduce chunk: a quad walks L buckets (2 L additions), then ~2 c operations multiply by the chunk's first index, and
  // every workgroup of 64 quads is one wave per SIMD of a CU.  The chains are bound by the instructions a wave issues, and a
  // second workgroup on a CU slows both by ~1.6x, so: the smallest L (at least 4) for which the workgroups that have chunks fit
  // one per CU.  JJ_MSM_REDUCE_CHUNK overrides; never more than one window.
  auto reduce_blocks = [&](u32 l) { const u32 nk = std::min<u32>(MSM_TREE_QUADS, (B / l + MSM_TREE_QUADS - 1) / MSM_TREE_QUADS); u32 t = 0; for (u32 s = 0; s < Ws; s++) t += msm_reduce_blocks(mp, (int)s, l, nk); return t; };
  u32 L_auto = 4;
  while (L_auto < B && reduce_blocks(L_auto) > (u32)c->cus) L_auto <<= 1;
  const u32 L = std::min<u32>(c->msm_reduce_chunk ? (u32)c->msm_reduce_chunk : L_auto, B);
  if (B % L || (L & (L - 1))) { c->err = "inconsistent MSM tuning override (JJ_MSM_REDUCE_CHUNK must be a power of two dividing the bucket count)"; return JJ_ERR_INVALID; }
  const u32 K = B / L, nblk = std::min<u32>(MSM_TREE_QUADS, (K + MSM_TREE_QUADS - 1) / MSM_TREE_QUADS);      // workgroups of 64 quads per window, at most
  const u32 reduce_grid = reduce_blocks(L);
  int jbits = 0; while ((1u << jbits) < B) jbits++;
  // Two-level reduce (k_msm_reduce_l1 + k_msm_reduce_l2) for wide windows: level 1 sums R rows of the bucket matrix per lane (whole-lane
  // additions at full throughput), level 2 is the quad chain over M = B / R columns per window.  Narrow windows (2^17-term passes: 1024
  // buckets per window) keep the one-level kernel: level 1 would be an extra launch and ~20 us of chain for a level 2 that is as deep.
  u32 l1_rows = 0;
  if (c->msm_l1_rows > 0) l1_rows = (u32)c->msm_l1_rows;
  else if (c->msm_l1_rows < 0 && B >= 16384) l1_rows = B >= 32768 ? 8 : 4;
  const u32 Bmin = mp.r ? B / 2 : B;                                     // buckets of the narrowest window of the layout
  while (l1_rows > 1 && (l1_rows > Bmin || B / l1_rows < 64)) l1_rows >>= 1;     // every window has at least one row; whole waves per window
  if (l1_rows < 2) l1_rows = 0;
  int mbits = 0; u32 L2 = 0, nblk2 = 0;
  if (l1_rows) {
    const u32 M = B / l1_rows;
    while ((1u << mbits) < M) mbits++;
    L2 = 4;
    while (L2 < M && ((u64)Ws * ((M / L2 + MSM_TREE_QUADS - 1) / MSM_TREE_QUADS) > (u64)c->cus || (M / L2 + MSM_TREE_QUADS - 1) / MSM_TREE_QUADS > (u32)MSM_TREE_QUADS)) L2 <<= 1;
    if (c->msm_l2_chunk && (u32)c->msm_l2_chunk <= M && (M / (u32)c->msm_l2_chunk + MSM_TREE_QUADS - 1) / MSM_TREE_QUADS <= (u32)MSM_TREE_QUADS) L2 = (u32)c->msm_l2_chunk;
    nblk2 = (M / L2 + MSM_TREE_QUADS - 1) / MSM_TREE_QUADS;
  }
  const size_t l1_bytes = l1_rows ? (size_t)Ws * (B / l1_rows) * ENIELS_WORDS * 4 : 0;      // one array of extended-Niels records (S, then T)
  int rc;
  DevBuf &kprime = ln.buf[0], &niels = ln.buf[1], &offb = ln.buf[2], &idx = ln.buf[3], &buckets = ln.buf[4], &ra = ln.buf[5], &tcnt = ln.buf[7];
  u32 chunk = MSM_CHUNK_MIN;                           // 16 entries per lane up to 2^19 terms, 32 at 2^20, then proportional to n (measured)
  while (chunk < 256 && ((size_t)chunk << 15) < n) chunk <<= 1;
  if (n < ((size_t)1 << 15)) chunk = 8;               // small inputs (only reached with the small-batch path switched off): more lanes, shorter chains
  if (c->msm_chunk) chunk = (u32)c->msm_chunk;
  const u32 nchunk = (u32)((n + chunk - 1) / chunk);
  const bool two_pass = B > 4096 || (B == 4096 && c->msm_two_pass != 0);      // the one-pass plan kernel covers 4096 buckets per window
  const u32 HB = B >> MSM_LO_BITS;
  if (two_pass && HB > MSM_HB_MAX) { c->err = "MSM window layout has more coarse bins per window than the two-pass sort holds (JJ_MSM_WINDOWS must be 16..36)"; return JJ_ERR_INVALID; }
  const u32 ptiles = (u32)((n + MSM_P1_TILE - 1) / MSM_P1_TILE);        // the first pass of the two-pass sort orders a whole tile in LDS
  const size_t pm = (size_t)HB * ptiles;                               // runs per slot
  // one-pass sort tiles: enough (tile, window) blocks to fill the GPU, each at least 4096 terms
  const u32 ntiles = (u32)std::max<size_t>(1, std::min<size_t>((size_t)(c->cus + Ws - 1) / Ws, (n + 4095) / 4096));
  const size_t tile = (n + ntiles - 1) / ntiles;
  const bool use_segments = c->msm_segments == 1 || (c->msm_segments < 0 && n >= MSM_LARGE_MIN);
  // segments of at most P entries, sorted by length; P bounds the serial depth of one lane: twice the mean bucket of the widest windows
  u32 P = (u32)std::min<size_t>(SEG_PMAX, std::max<size_t>(32, 2 * n / B));
  if (c->msm_seg_len >= 8 && c->msm_seg_len <= SEG_PMAX) P = (u32)c->msm_seg_len;
  const u32 stiles = (u32)std::max<size_t>(1, std::min<size_t>(4096, (nb + 255) / 256));          // tiles of the two segment passes: 256 buckets each, more above 2^20 buckets
  const u32 per_tile = (u32)((nb + stiles - 1) / stiles);
  const size_t max_segs = nb + (n * (size_t)Ws) / P + 1;
  const size_t bh_words = (size_t)stiles * (P + 1), hdr_words = bh_words + 2 * (P + 2) + 16;
  if ((rc = ensure(c, kprime, n * 32))) return rc;
  if ((rc = ensure(c, niels, n * (size_t)GNIELS_WORDS * 4))) return rc;
  if ((rc = ensure(c, offb, (size_t)Ws * (B + 1) * 4))) return rc;
  if ((rc = ensure(c, idx, n * (size_t)Ws * 4))) return rc;
  if ((rc = ensure(c, tcnt, two_pass ? ((size_t)Ws * (2 * pm + 1)) * 4 : (size_t)Ws * ntiles * B * 4))) return rc;
  if ((rc = ensure(c, buckets, (size_t)EXT_AOS_WORDS * 4 * nb))) return rc;
  // ra: first the two-pass sort's records (4 + 1 bytes per entry), then the chunk heads / segment heads
  // (and, once the heads are folded in, the two arrays level 1 of the reduce hands to level 2)
  if ((rc = ensure(c, ra, std::max<size_t>(std::max<size_t>((size_t)EXT_AOS_WORDS * 4 * std::max<size_t>((size_t)Ws * nchunk, (n * (size_t)Ws) / 8 + 1), n * (size_t)Ws * 5 + 64), 2 * l1_bytes)))) return rc;
  if ((rc = msm_ensure_ctl(c, ln))) return rc;                                                           // counters, big-bucket work list, workgroup partial sums
  if ((rc = ensure(c, ln.bigpart, (size_t)5 * NL * 4 * FIXUP_BIG_MAX * FIXUP_BIG_QUADS))) return rc;      // the big buckets' partial sums
  if (use_segments && (rc = ensure(c, ln.seg, hdr_words * 4 + 16 + nb * sizeof(MergeItem) + max_segs * sizeof(Seg)))) return rc;   // bh [stiles][P+1] | count [P+1] | offset [P+2] | merge list | segments
  hipStream_t st = ln.stream;
  u32* off = (u32*)offb.p;
  u32* counters = (u32*)ln.ctl.p;                          // cleared by the sort's plan kernel
  BigBucket* big = (BigBucket*)((uint8_t*)ln.ctl.p + MSM_BIG_OFF);
  u32* part = (u32*)((uint8_t*)ln.ctl.p + MSM_PART_OFF);
  // One conversion launch for scalars and points.  (Rounds 2-3 ran the point half on a second stream beside the sort from 2^18
  // terms; with the entries staged through LDS the conversion is short enough that the fork, its two events and the contention
  // with the sort's first kernel cost more than the overlap returns: 2^18 terms 0.565 -> 0.556 ms, 2^20 1.262 -> 1.254 ms.)
  hipLaunchKernelGGL(k_msm_convert, dim3(blocks_for(n)), dim3(256), 0, st, n, ds, dp, mp, (u32*)kprime.p, (u32*)niels.p, 3);
  if (two_pass) {
    u32* tc = (u32*)tcnt.p; u32* tcs = tc + (size_t)Ws * pm;
    u32* rec = (u32*)ra.p; uint8_t* lo8 = (uint8_t*)ra.p + n * (size_t)Ws * 4;     // the head buffer is free until the accumulation
    hipLaunchKernelGGL(k_msm_part_hist, dim3(ptiles, Ws), dim3(MSM_SORT_THREADS), 0, st, n, (size_t)MSM_P1_TILE, mp, (const u32*)kprime.p, tc);
    hipLaunchKernelGGL(k_msm_part_plan, dim3(Ws), dim3(1024), 0, st, n, (u32)pm, (const u32*)tc, tcs, counters);
    hipLaunchKernelGGL(k_msm_part_scatter, dim3(ptiles, Ws), dim3(MSM_SORT_THREADS), 0, st, n, (size_t)MSM_P1_TILE, mp, (const u32*)kprime.p, (const u32*)tcs, rec, lo8);
    hipLaunchKernelGGL(k_msm_part_sort, dim3(HB, Ws), dim3(MSM_P2_THREADS), 0, st, mp, ptiles, (const u32*)tcs, (const u32*)rec, (const uint8_t*)lo8, (u32*)idx.p, off);
  } else {
    hipLaunchKernelGGL(k_msm_hist, dim3(ntiles, Ws), dim3(MSM_SORT_THREADS), B * 4, st, n, tile, mp, (const u32*)kprime.p, (u32*)tcnt.p);
    hipLaunchKernelGGL(k_msm_plan, dim3(Ws), dim3(1024), 0, st, n, B, ntiles, (u32*)tcnt.p, off, counters);
    hipLaunchKernelGGL(k_msm_scatter, dim3(ntiles * 8 * ((Ws + 7) / 8)), dim3(MSM_SORT_THREADS), B * 4, st, n, tile, ntiles, mp, (const u32*)kprime.p, (const u32*)tcnt.p, (u32*)idx.p);
  }
  const ExtAoS head{(u32*)ra.p}, bk{(u32*)buckets.p};
  SoA partial = soa_of(ln.bigpart, (size_t)FIXUP_BIG_MAX * FIXUP_BIG_QUADS);
  const MergeItem* merge_list = nullptr;
  if (use_segments) {
    u32* bh = (u32*)ln.seg.p; u32* soff = bh + bh_words + (P + 1);
    MergeItem* merge = (MergeItem*)(((uintptr_t)(bh + hdr_words) + 15) & ~(uintptr_t)15);
    Seg* seg = (Seg*)(merge + nb);
    hipLaunchKernelGGL(k_seg_hist, dim3(stiles), dim3(256), 0, st, nb, B, per_tile, P, (const u32*)off, bk, bh);
    hipLaunchKernelGGL(k_seg_plan, dim3(P + 1), dim3(256), 0, st, stiles, bh, soff);
    hipLaunchKernelGGL(k_seg_scatter, dim3(stiles), dim3(256), 0, st, nb, B, per_tile, P, (const u32*)off, (const u32*)bh, (const u32*)soff, soff + (P + 1), seg, counters, merge, big);
    hipLaunchKernelGGL(k_msm_accumulate_seg, dim3(blocks_for(max_segs)), dim3(256), 0, st, (const u32*)(soff + (P + 1)), (const Seg*)seg, (const u32*)idx.p, (const u32*)niels.p, bk, head);
    merge_list = merge;
  } else {
    hipLaunchKernelGGL(k_msm_accumulate, dim3(blocks_for(nchunk), Ws), dim3(256), 0, st, n, B, chunk, nchunk, (const u32*)off, (const u32*)idx.p, (const u32*)niels.p, bk, head);
    hipLaunchKernelGGL(k_msm_fixup, dim3(blocks_for(2 * nb)), dim3(256), 0, st, n, B, Ws, chunk, nchunk, (const u32*)off, bk, head, counters, big);
  }
  // (segment path: 512 workgroups, the merge list of repeated scalars is walked by the same launch)
  hipLaunchKernelGGL(k_msm_fixup_big, dim3(merge_list ? 2u * (unsigned)c->cus : 256u), dim3(256), 0, st, counters, (const BigBucket*)big, bk, head, partial, merge_list);
  if (l1_rows) {
    u32* SN = (u32*)ra.p; u32* TN = (u32*)((uint8_t*)ra.p + l1_bytes);        // the heads are dead: k_msm_fixup_big was their last reader
    hipLaunchKernelGGL(k_msm_reduce_l1, dim3(blocks_for((size_t)Ws << mbits)), dim3(256), 0, st, mp, mbits, bk, SN, TN);
    hipLaunchKernelGGL(k_msm_reduce_l2, dim3(Ws * nblk2), dim3(4 * MSM_TREE_QUADS), 0, st, n, mp, mbits, L2, nblk2, (const u32*)SN, (const u32*)TN, part, counters, (u32*)rec_dev);
  } else if (K > MSM_TREE_QUADS * nblk) hipLaunchKernelGGL(k_msm_reduce_fold<true>, dim3(reduce_grid), dim3(4 * MSM_TREE_QUADS), 0, st, n, mp, L, nblk, jbits, bk, part, counters, (u32*)rec_dev);
  else hipLaunchKernelGGL(k_msm_reduce_fold<false>, dim3(reduce_grid), dim3(4 * MSM_TREE_QUADS), 0, st, n, mp, L, nblk, jbits, bk, part, counters, (u32*)rec_dev);
  return JJ_OK;
}
// one pass (at most 2^24 terms: 32-bit sort indices), record left at rec_dev
static int msm_enqueue(jj_ctx* c, MsmLane& L, size_t n, const void* ds, const void* dp, int part_w0, int part_stride, void* rec_dev, size_t* rec_bytes) {
  const bool small = n <= (size_t)c->msm_small_max;
  *rec_bytes = jjhost::rec_bytes(small ? SM_W : msm_windows_for(c, n));
  return small ? msm_enqueue_small(c, L, n, ds, dp, part_w0, part_stride, rec_dev) : msm_enqueue_pippenger(c, L, n, ds, dp, part_w0, part_stride, rec_dev);
}
// lane k of the context, ready for use: lane 0 follows the context's launch stream; the others own two streams, created on first
// use, and start their work after everything already queued on the launch stream (the inputs may have been produced there)
static int msm_lane(jj_ctx* c, int k, MsmLane** out) {
  MsmLane& L = c->lanes[k];
  if (k == 0) { L.stream = c->stream; *out = &L; return JJ_OK; }
  if (!L.owned) {
    HIPCHK(c, hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&L.ready_ev, hipEventDisableTiming));
    L.owned = true;
  }
  HIPCHK(c, hipEventRecord(L.ready_ev, c->stream));
  HIPCHK(c, hipStreamWaitEvent(L.stream, L.ready_ev, 0));
  *out = &L;
  return JJ_OK;
}

// ---- asynchronous jobs: jj_msm_begin queues every pass of one MSM and the copy of its records into the job's own page-locked
// buffer, then returns; jj_msm_finish waits for that copy only and runs the host tail (window sums, Horner, one inversion),
// while the kernels of jobs begun meanwhile keep the device busy.  The context's workspaces are shared by all jobs: the stream
// orders them.
static int msm_job_get(jj_ctx* c, size_t nrec, jj_msm_job** out) {
  jj_msm_job* j = nullptr;
  if (!c->job_pool.empty()) { j = c->job_pool.back(); c->job_pool.pop_back(); }
  else {
    j = new jj_msm_job();
    j->c = c;
    if (hipEventCreateWithFlags(&j->ev, hipEventDisableTiming) != hipSuccess) { delete j; c->err = "hipEventCreate failed"; return JJ_ERR_HIP; }
  }
  const size_t want = std::max<size_t>(nrec, 1) * jjhost::REC_MAX_BYTES;
  if (j->cap < want) {
    if (j->host) (void)hipHostFree(j->host);
    j->host = nullptr; j->cap = 0;
    if (hipHostMalloc((void**)&j->host, want, hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) { (void)hipEventDestroy(j->ev); delete j; c->err = "hipHostMalloc failed"; return JJ_ERR_NOMEM; }
    j->cap = want;
  }
  j->nrec = 0;
  for (size_t r = 0; r < std::max<size_t>(nrec, 1); r++) memset(j->host + r * jjhost::REC_MAX_BYTES, 0, jjhost::REC_HDR_BYTES);   // a stale header of a pooled buffer must never validate
  *out = j;
  return JJ_OK;
}
static void msm_job_put(jj_ctx* c, jj_msm_job* j) {
  if (c->job_pool.size() < 8) { c->job_pool.push_back(j); return; }
  if (j->host) (void)hipHostFree(j->host);
  (void)hipEventDestroy(j->ev);
  delete j;
}
// spread: device-pointer jobs alternate over the context's lanes (jj_msm_begin); otherwise lane 0 (jj_msm; host arrays are staged
// through buffers the launch stream owns)
static int msm_begin_locked(jj_ctx* c, size_t n, const void* scalars, const void* points, int part_w0, int part_stride, bool spread, jj_msm_job** out) {
  size_t PASS = (size_t)1 << c->msm_pass_log2;
  // Host arrays of 2^19 terms and more are reduced in SEVERAL passes (one record each, one host tail): the copy of a pass's slice runs on the
  // copy stream beside the kernels of the pass before it -- 96 bytes per term over the link cost more than the whole reduction (2^20 terms:
  // 1.7 ms of copy, 1.3 ms of kernels).  Two to eight passes of at least 2^19 terms (smaller passes reduce less efficiently than the copy
  // they hide): page-locked arrays 2^20 terms 3.15 -> 2.67 ms, 2^22 terms 12.3 -> 9.7 ms with two passes.  JJ_MSM_HOST_SPLIT=0: one pass
  // after the whole copy (round 3).
  const bool host_in = n && !is_device_ptr(scalars) && !is_device_ptr(points);
  const bool split = host_in && c->msm_host_split && n >= ((size_t)1 << 19);
  PASS = msm_host_pass_terms(n, c->msm_pass_log2, split);
  const size_t npass = n ? (n + PASS - 1) / PASS : 0;
  jj_msm_job* j;
  int rc = msm_job_get(c, npass, &j); if (rc) return rc;
  int k = 0;
  if (spread && n && c->msm_lanes > 1 && is_device_ptr(scalars) && is_device_ptr(points)) k = (int)(c->next_lane++ % (unsigned)c->msm_lanes);
  MsmLane* L = nullptr;
  if ((rc = msm_lane(c, k, &L))) { msm_job_put(c, j); return rc; }
  auto fail = [&](int code) { if (split && c->pipe.h2d) (void)hipStreamSynchronize(c->pipe.h2d); (void)hipStreamSynchronize(L->stream); (void)hipGetLastError(); msm_job_put(c, j); return code; };   // kernels may still be writing into the job's buffer
  if (n) {
    const void *ds, *dp;
    if (split) {
      // the passes' slices are copied on the copy stream (ordered after what the launch stream has queued: the staging buffers may still
      // be read by an earlier call's kernels), each pass's kernels wait for their slice only
      if ((rc = ensure(c, c->in[0], 32 * n)) || (rc = ensure(c, c->in[1], 64 * n)) || (rc = pipe_prepare(c, 0, 0))) return fail(rc);
      ds = c->in[0].p; dp = c->in[1].p;
      hipError_t e0 = hipEventRecord(c->order_ev, c->stream);
      if (e0 == hipSuccess) e0 = hipStreamWaitEvent(c->pipe.h2d, c->order_ev, 0);
      if (e0 != hipSuccess) { c->err = std::string("MSM staging failed: ") + hipGetErrorString(e0); return fail(JJ_ERR_HIP); }
    } else if ((rc = stage_in(c, 0, scalars, 32 * n, &ds)) || (rc = stage_in(c, 1, points, 64 * n, &dp))) return fail(rc);
    size_t stage_seq = 0;             // staging slots of the bounce path, counted over all arrays of all passes
    for (size_t lo = 0; lo < n; lo += PASS) {
      const size_t cnt = std::min(PASS, n - lo);
      size_t used = 0;
      if (split) {
        const struct { const void* host; void* dev; size_t elem; } arr[2] = {{scalars, c->in[0].p, 32}, {points, c->in[1].p, 64}};
        for (const auto& a : arr) {
          const uint8_t* src = (const uint8_t*)a.host + lo * a.elem; uint8_t* dst = (uint8_t*)a.dev + lo * a.elem;
          if (cnt * a.elem >= BOUNCE_THRESHOLD && !is_pinned_host(src, cnt * a.elem)) { if ((rc = host_to_dev_bounced(c, dst, src, cnt * a.elem, c->pipe.h2d, &stage_seq))) return fail(rc); }
          else if (hipMemcpyAsync(dst, src, cnt * a.elem, hipMemcpyHostToDevice, c->pipe.h2d) != hipSuccess) { c->err = "MSM staging copy failed"; return fail(JJ_ERR_HIP); }
        }
        const int ei = (int)((lo / PASS) & 1);
        if (hipEventRecord(c->pipe.ev_in[ei], c->pipe.h2d) != hipSuccess || hipStreamWaitEvent(L->stream, c->pipe.ev_in[ei], 0) != hipSuccess) { c->err = "MSM staging event failed"; return fail(JJ_ERR_HIP); }
      }
      // the kernels that finish a window write its point straight into the job's page-locked buffer (device-visible host memory):
      // no copy operation between the last kernel and the host tail
      if ((rc = msm_enqueue(c, *L, cnt, (const uint8_t*)ds + lo * 32, (const uint8_t*)dp + lo * 64, part_w0, part_stride, j->host + j->nrec * jjhost::REC_MAX_BYTES, &used))) return fail(rc);
      j->nrec++;
    }
    if (stage_seq && (rc = stage_in_drain(c, stage_seq))) return fail(rc);     // the staging slots are free for the next call
  }
  hipError_t e = hipEventRecord(j->ev, L->stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) { c->err = std::string("MSM launch failed: ") + hipGetErrorString(e); return fail(JJ_ERR_HIP); }
  *out = j;
  return JJ_OK;
}
JJ_API int jj_msm_begin(jj_ctx* c, size_t n, const void* scalars, const void* points, jj_msm_job** job) {
  if (!c || !job) return JJ_ERR_INVALID;
  *job = nullptr;
  JJ_ENTER(c);
  return msm_begin_locked(c, n, scalars, points, 0, 1, true, job);
}
// waits for the job's records, host tail, result to out64 (host pointer: written before the call returns; device pointer: a
// 64-byte copy queued on the context's stream).  The job is released in every case.
JJ_API int jj_msm_finish(jj_msm_job* j, void* out64) {
  if (!j || !j->c) return JJ_ERR_INVALID;
  jj_ctx* c = j->c;
  if (!out64) { (void)hipEventSynchronize(j->ev); std::lock_guard<std::recursive_mutex> lk(c->mu); msm_job_put(c, j); return JJ_ERR_INVALID; }   // the job's kernels may still be writing into its buffer
  hipError_t e = hipEventSynchronize(j->ev);                       // no context lock while waiting: other threads may queue work
  jjhost::Ext total = jjhost::identity();
  const bool ok = e == hipSuccess && jjhost::combine_records(j->host, j->nrec, jjhost::REC_MAX_BYTES, &total);
  JJ_ENTER(c);
  int rc = JJ_OK;
  if (e != hipSuccess) { c->err = std::string("hipEventSynchronize failed: ") + hipGetErrorString(e); rc = JJ_ERR_HIP; }
  else if (!ok) { c->err = "MSM record is damaged (bad header)"; rc = JJ_ERR_HIP; }
  else if (is_device_ptr(out64)) {
    jjhost::to_affine64(c->host_out[c->host_out_next], total);
    e = hipMemcpyAsync(out64, c->host_out[c->host_out_next], 64, hipMemcpyHostToDevice, c->stream);
    c->host_out_next = (c->host_out_next + 1) % 8;
    if (e != hipSuccess) { c->err = std::string("hipMemcpyAsync failed: ") + hipGetErrorString(e); rc = JJ_ERR_HIP; }
  } else jjhost::to_affine64((uint8_t*)out64, total);
  msm_job_put(c, j);
  return rc;
}
JJ_API int jj_msm(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out64) {
  if (!c || !out64) return JJ_ERR_INVALID;
  jj_msm_job* j = nullptr;
  {
    JJ_ENTER(c);
    prof_mark(c, 0);
    const int rc = msm_begin_locked(c, n, scalars, points, 0, 1, false, &j);
    if (rc) return rc;
  }
  const int rc = jj_msm_finish(j, out64);
  { JJ_ENTER(c); prof_mark(c, 1); prof_mark(c, 2); }
  return rc;
}
// Opt-in device-side finish: the record stays on the device, one quad runs the Horner chain and the inversion there, the affine sum
// is written to DEVICE memory; nothing is copied to the host and no host thread waits (fully asynchronous on the context's stream).
JJ_API int jj_msm_dev(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out64_dev) {
  if (!c || !out64_dev) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  if (!is_device_ptr(out64_dev) || ((uintptr_t)out64_dev & 15u)) { c->err = "jj_msm_dev writes its result to (16-byte aligned) device memory; use jj_msm for a host result"; return JJ_ERR_INVALID; }
  if (n > ((size_t)1 << c->msm_pass_log2)) { c->err = "jj_msm_dev takes at most one pass of terms (2^24); use jj_msm, or add the sums of the parts with jj_point_add"; return JJ_ERR_INVALID; }
  if (n == 0) { HIPCHK(c, hipMemcpyAsync(out64_dev, AFFINE_IDENTITY_BYTES, 64, hipMemcpyHostToDevice, c->stream)); return JJ_OK; }
  int rc;
  const void *ds, *dp;
  if ((rc = stage_in(c, 0, scalars, 32 * n, &ds))) return rc;
  if ((rc = stage_in(c, 1, points, 64 * n, &dp))) return rc;
  MsmLane* L = nullptr;
  if ((rc = msm_lane(c, 0, &L))) return rc;
  if ((rc = ensure(c, L->rec, JJ_MSM_PARTIAL_BYTES))) return rc;
  size_t used = 0;
  if ((rc = msm_enqueue(c, *L, n, ds, dp, 0, 1, L->rec.p, &used))) return rc;
  hipLaunchKernelGGL(k_msm_finish_dev, dim3(1), dim3(64), 0, c->stream, (const u32*)L->rec.p, out64_dev);
  return finish(c, false);
}
// First half of an MSM that is cut across devices or ranks: the record of partial window sums, left where the
// caller wants it (device memory: ready for an all_gather over RCCL; host memory: the call waits for the copy).
//   part_index / part_count = 0 / 1   all windows of the n terms given (term partition: every rank passes its own terms)
//   part_index = g, part_count = G    windows g, g + G, ... of the n terms given (window partition: every rank passes ALL terms)
JJ_API int jj_msm_partial(jj_ctx* c, size_t n, const void* scalars, const void* points, int part_index, int part_count, void* record) {
  if (!c || !record || part_count < 1 || part_index < 0 || part_index >= part_count) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  if (n > ((size_t)1 << c->msm_pass_log2)) { c->err = "jj_msm_partial takes at most one pass of terms (2^24); cut larger inputs"; return JJ_ERR_INVALID; }
  int rc; OutRef o;
  if ((rc = stage_out(c, c->out[0], record, JJ_MSM_PARTIAL_BYTES, &o))) return rc;
  static_assert(JJ_MSM_PARTIAL_BYTES == jjhost::REC_MAX_BYTES, "record size");
  HIPCHK(c, hipMemsetAsync(o.dev, 0, JJ_MSM_PARTIAL_BYTES, c->stream));
  // the window count this call's layout has (as msm_enqueue picks it); a window partition over more parts than windows leaves
  // the parts beyond the last window nothing to do (zero-sized grids would fail the launches and no header would be written)
  const int layout_W = n <= (size_t)c->msm_small_max ? SM_W : msm_windows_for(c, n);
  if (n == 0 || part_index >= layout_W) {
    // an empty shard: a valid record without windows
    uint32_t hdr[MSM_REC_HDR_WORDS] = {MSM_REC_MAGIC, 2u, (uint32_t)(n == 0 ? SM_W : layout_W), 1u};
    hdr[6] = (uint32_t)n; hdr[7] = (uint32_t)((uint64_t)n >> 32);
    memcpy(c->host_out[c->host_out_next], hdr, 64);
    HIPCHK(c, hipMemcpyAsync(o.dev, c->host_out[c->host_out_next], 64, hipMemcpyHostToDevice, c->stream));
    c->host_out_next = (c->host_out_next + 1) % 8;
  } else {
    const void *ds, *dp;
    if ((rc = stage_in(c, 0, scalars, 32 * n, &ds))) return rc;
    if ((rc = stage_in(c, 1, points, 64 * n, &dp))) return rc;
    size_t used = 0;
    MsmLane* L = nullptr;
    if ((rc = msm_lane(c, 0, &L))) return rc;
    if ((rc = msm_enqueue(c, *L, n, ds, dp, part_index, part_count, o.dev, &used))) return rc;
  }
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
// Second half: `count` records (HOST memory, JJ_MSM_PARTIAL_BYTES apart: what the ranks' all_gather delivered, copied back once)
// -> one affine point.  Host only, no context: window sums of all records, one Horner chain per window layout, one inversion.
JJ_API int jj_msm_combine(size_t count, const void* records_host, void* out64_host) {
  if (!out64_host || (count && !records_host) || is_device_ptr(out64_host) || (count && is_device_ptr(records_host))) return JJ_ERR_INVALID;
  jjhost::Ext total = jjhost::identity();
  if (!jjhost::combine_records((const uint8_t*)records_host, count, JJ_MSM_PARTIAL_BYTES, &total)) return JJ_ERR_INVALID;
  jjhost::to_affine64((uint8_t*)out64_host, total);
  return JJ_OK;
}

// ---- multi-rank MSM behind the C ABI ("context: streams, tables, RCCL comm"; 8(e)).  The communicator is the caller's
// (its rendezvous -- who carries the ncclUniqueId to whom -- belongs to the application: examples/msm_rccl.cpp does it with a file,
// bench.py over torch.distributed); the context borrows it.  libjubjub_hip.so does not link RCCL: ncclAllGather is taken from the
// caller, or looked up in the process, or in librccl.so.1 -- it must be the ncclAllGather of the library that made the communicator.
JJ_API int jj_ctx_set_comm(jj_ctx* c, void* nccl_comm, int rank, int nranks, void* all_gather_fn) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  if (!nccl_comm) { c->comm = nullptr; c->comm_rank = 0; c->comm_nranks = 1; c->all_gather = nullptr; return JJ_OK; }   // detach
  if (nranks < 1 || nranks > 4096 || rank < 0 || rank >= nranks) { c->err = "jj_ctx_set_comm: bad rank / nranks"; return JJ_ERR_INVALID; }
  void* fn = all_gather_fn;
  if (!fn) fn = dlsym(RTLD_DEFAULT, "ncclAllGather");
  if (!fn) { void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL); if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL); if (h) fn = dlsym(h, "ncclAllGather"); }
  if (!fn) { c->err = "jj_ctx_set_comm: ncclAllGather not found (pass its address, or load librccl first)"; return JJ_ERR_INVALID; }
  c->comm = nccl_comm; c->comm_rank = rank; c->comm_nranks = nranks; c->all_gather = (jj_ctx::AllGatherFn)fn;
  return JJ_OK;
}
// One MSM over the terms (partition 0: each rank passes ITS terms) or the windows (partition 1: each rank passes ALL terms) of every
// rank of the communicator: record of window sums on this device -> ncclAllGather of JJ_MSM_PARTIAL_BYTES per rank over xGMI -> ONE
// copy of the gathered records to the host -> ONE host tail (jj_msm_combine) on every rank.  Every rank gets the same point.
JJ_API int jj_msm_allgather(jj_ctx* c, size_t n, const void* scalars, const void* points, int partition, void* out64) {
  if (!c || !out64 || (partition != 0 && partition != 1)) return JJ_ERR_INVALID;
  {
    JJ_ENTER(c);
    if (!c->comm || !c->all_gather) { c->err = "jj_msm_allgather: no communicator (jj_ctx_set_comm)"; return JJ_ERR_INVALID; }
  }
  const int G = c->comm_nranks;
  int rc;
  {
    JJ_ENTER(c);
    if ((rc = ensure(c, c->gather_dev, (size_t)(G + 1) * JJ_MSM_PARTIAL_BYTES))) return rc;
    if (c->gather_host_cap < (size_t)G * JJ_MSM_PARTIAL_BYTES) {
      if (c->gather_host) (void)hipHostFree(c->gather_host);
      c->gather_host = nullptr; c->gather_host_cap = 0;
      if (hipHostMalloc((void**)&c->gather_host, (size_t)G * JJ_MSM_PARTIAL_BYTES, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->err = "hipHostMalloc failed"; return JJ_ERR_NOMEM; }
      c->gather_host_cap = (size_t)G * JJ_MSM_PARTIAL_BYTES;
    }
  }
  uint8_t* mine = (uint8_t*)c->gather_dev.p;                       // this rank's record, then the G gathered ones
  uint8_t* all = mine + JJ_MSM_PARTIAL_BYTES;
  if ((rc = jj_msm_partial(c, n, scalars, points, partition ? c->comm_rank : 0, partition ? G : 1, mine))) return rc;
  JJ_ENTER(c);
  const int nrc = c->all_gather(mine, all, JJ_MSM_PARTIAL_BYTES, /* ncclUint8 */ 1, c->comm, c->stream);
  if (nrc != 0) { c->err = "ncclAllGather failed with ncclResult_t " + std::to_string(nrc); return JJ_ERR_HIP; }
  HIPCHK(c, hipMemcpyAsync(c->gather_host, all, (size_t)G * JJ_MSM_PARTIAL_BYTES, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  jjhost::Ext total = jjhost::identity();
  if (!jjhost::combine_records(c->gather_host, (size_t)G, JJ_MSM_PARTIAL_BYTES, &total)) { c->err = "a gathered MSM record is damaged (bad header)"; return JJ_ERR_HIP; }
  if (is_device_ptr(out64)) {
    jjhost::to_affine64(c->host_out[c->host_out_next], total);
    HIPCHK(c, hipMemcpyAsync(out64, c->host_out[c->host_out_next], 64, hipMemcpyHostToDevice, c->stream));
    c->host_out_next = (c->host_out_next + 1) % 8;
  } else jjhost::to_affine64((uint8_t*)out64, total);
  return JJ_OK;
}

// ---------------------------------------------------------------------------------------------------- synthetic inputs
static int synth32(jj_ctx* c, size_t n, uint64_t seed, uint64_t first_index, int raw, void* out32);
JJ_API int jj_synth_scalars(jj_ctx* c, size_t n, uint64_t seed, uint64_t first_index, void* out32) { return synth32(c, n, seed, first_index, 0, out32); }
JJ_API int jj_synth_bytes32(jj_ctx* c, size_t n, uint64_t seed, uint64_t first_index, void* out32) { return synth32(c, n, seed, first_index, 1, out32); }
static int synth32(jj_ctx* c, size_t n, uint64_t seed, uint64_t first_index, int raw, void* out32) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  int rc; OutRef o;
  if ((rc = stage_out(c, c->out[0], out32, 32 * n, &o))) return rc;
  if (n) hipLaunchKernelGGL(k_synth_scalars, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, (u64)seed, (u64)first_index, raw, o.dev);
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
JJ_API int jj_random_points(jj_ctx* c, size_t n, uint64_t seed, uint64_t first_index, int subgroup, void* out64, uint32_t* attempts) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  int rc; OutRef o, ao; ao.host = false; ao.dev = nullptr;
  if ((rc = stage_out(c, c->out[0], out64, 64 * n, &o))) return rc;
  if (attempts && (rc = stage_out(c, c->out[1], attempts, 4 * n, &ao))) return rc;
  if (n) hipLaunchKernelGGL(k_random_points, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, (u64)seed, (u64)first_index, subgroup ? 1 : 0, c->sqrt_tables, o.dev, (u32*)ao.dev);
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  if (attempts && (rc = finish_out(c, ao, &sync))) return rc;
  return finish(c, sync);
}

// ---------------------------------------------------------------------------------------------------- encodings
JJ_API int jj_compress(jj_ctx* c, size_t n, const void* points, void* out32) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  const void* dp; int rc; OutRef o;
  if ((rc = stage_in(c, 0, points, 64 * n, &dp))) return rc;
  if ((rc = stage_out(c, c->out[0], out32, 32 * n, &o))) return rc;
  if (n) hipLaunchKernelGGL(k_compress, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, dp, o.dev);
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
// the decoder and the flag kernels that follow it, on device pointers (n > 0), all on c->stream
static int decompress_dev(jj_ctx* c, size_t n, const void* di, unsigned flags, void* dout, uint8_t* dok, bool prof) {
  int rc;
  if ((rc = ensure(c, c->ws->scratch, (size_t)NL * 4 * n))) return rc;
  SoA scratch = soa_of(c->ws->scratch, n);
  if (prof) prof_mark(c, 0);
  const size_t lanes_wanted = (size_t)c->cus * 64 * 8;
  if (n >= lanes_wanted * 32) { size_t T = (n + 31) / 32; hipLaunchKernelGGL((k_decompress<32>), dim3(blocks_for(T)), dim3(256), 0, c->stream, n, T, di, flags, scratch, c->sqrt_tables, dout, dok); }
  else if (n >= lanes_wanted * 8 && n < lanes_wanted * 16 && c->dec_c_mid == 8) { size_t T = (n + 7) / 8; hipLaunchKernelGGL((k_decompress<8>), dim3(blocks_for(T)), dim3(256), 0, c->stream, n, T, di, flags, scratch, c->sqrt_tables, dout, dok); }
  else if (n >= lanes_wanted * 8) { size_t T = (n + 15) / 16; hipLaunchKernelGGL((k_decompress<16>), dim3(blocks_for(T)), dim3(256), 0, c->stream, n, T, di, flags, scratch, c->sqrt_tables, dout, dok); }
  else if (n <= 16384) { hipLaunchKernelGGL((k_decompress<1>), dim3(blocks_for(n)), dim3(256), 0, c->stream, n, n, di, flags, scratch, c->sqrt_tables, dout, dok); }   // latency: no shared inversion
  else { size_t T = (n + 3) / 4; hipLaunchKernelGGL((k_decompress<4>), dim3(blocks_for(T)), dim3(256), 0, c->stream, n, T, di, flags, scratch, c->sqrt_tables, dout, dok); }
  if (prof) { prof_mark(c, 1); prof_mark(c, 2); }
  if ((rc = pipe_to_tail(c))) return rc;                      // (host-buffer pipeline, stream mode 3: the flag kernels run beside the next chunk's decoder)
  // Invalid encodings were written as (0,0); the subgroup kernels below may compute garbage for them, the ok byte masks it.
  if (flags & JJ_DECOMPRESS_TORSION_FREE) { if ((rc = torsion_free_dev(c, n, dout, dok, 1))) return rc; }
  if (flags & (JJ_DECOMPRESS_NOT_SMALL_ORDER | JJ_DECOMPRESS_CLEAR_COFACTOR)) {
    if ((rc = ensure_ext(c, n, 3))) return rc;
    SoA ext = soa_of(c->ws->ext, n);
    hipLaunchKernelGGL(k_small_order_cofactor, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, (const void*)dout, flags, ext, dok);
    if (flags & JJ_DECOMPRESS_CLEAR_COFACTOR) { if ((rc = normalize_launch(c, n, ext, dout, 0))) return rc; }
  }
  if (flags & (JJ_DECOMPRESS_TORSION_FREE | JJ_DECOMPRESS_NOT_SMALL_ORDER | JJ_DECOMPRESS_CLEAR_COFACTOR))
    hipLaunchKernelGGL(k_mask_outputs, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, dout, (const uint8_t*)dok);
  return JJ_OK;
}
JJ_API int jj_decompress(jj_ctx* c, size_t n, const void* in32, unsigned flags, void* out64, uint8_t* ok) {
  if (!c || (!ok && n)) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  if (const size_t ch = pipe_chunk_for(c, n, 21); ch && all_host({in32, out64, ok})) {
    const HostIn in[1] = {{in32, 32}};
    const HostOut ho[2] = {{out64, 64}, {ok, 1}};
    const int prc = run_pipelined(c, n, ch, in, ho, [&](size_t cn, const void* const* di, void* const* dout) -> int {
      return decompress_dev(c, cn, di[0], flags, dout[0], (uint8_t*)dout[1], false);
    });
    if (prc <= 0) return prc;      // +1: buffers could not be page-locked -> plain staging below
  }
  const void* di; int rc; OutRef o, ko;
  if ((rc = stage_in(c, 0, in32, 32 * n, &di))) return rc;
  if ((rc = stage_out(c, c->out[0], out64, 64 * n, &o))) return rc;
  if ((rc = stage_out(c, c->okb, ok, n, &ko))) return rc;
  if (n && (rc = decompress_dev(c, n, di, flags, o.dev, (uint8_t*)ko.dev, true))) return rc;
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  if ((rc = finish_out(c, ko, &sync))) return rc;
  return finish(c, sync);
}
JJ_API int jj_batch_normalize(jj_ctx* c, size_t n, const void* ext160, void* out64) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  const void* de; int rc; OutRef o;
  if ((rc = stage_in(c, 0, ext160, 160 * n, &de))) return rc;
  if ((rc = stage_out(c, c->out[0], out64, 64 * n, &o))) return rc;
  if ((rc = ensure_ext(c, n, 3))) return rc;
  SoA ext = soa_of(c->ws->ext, n);
  if (n) {
    hipLaunchKernelGGL(k_ext160_to_soa, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, de, ext);
    if ((rc = normalize_launch(c, n, ext, o.dev, 0))) return rc;
  }
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}

// ---------------------------------------------------------------------------------------------------- several devices
// one context per device, one host thread + stream per device, contiguous shards [g*n/G, (g+1)*n/G), no
// data-path collective for the independent-batch workloads; the MSM's partial points (64 bytes per device) are folded on
// the calling host thread, where the Horner tail of every device's Pippenger already ran.  Array arguments are HOST
// pointers here (the batch lives in host memory and is cut across the devices; each shard goes through the single-device
// entry point, i.e. page-locked in place and pipelined over copy streams when it is large).  The same device may be listed
// more than once (several contexts on one GPU: used by the tests, and a way to overlap copies and kernels).
// Processes that keep their batches in HBM scale as one process per GPU instead (jubjub_amd/dist.py, bench.py).
struct jj_multi {
  std::vector<jj_ctx*> ctx;
  std::mutex mu;
  std::string err;
};
struct jj_mtable { std::vector<jj_table*> t; };

JJ_API int jj_multi_create(const int* devices, int ndev, jj_multi** out) {
  if (!out) return JJ_ERR_INVALID;
  *out = nullptr;
  if (!devices || ndev < 1 || ndev > 64) return JJ_ERR_INVALID;
  jj_multi* m = new jj_multi();
  for (int g = 0; g < ndev; g++) {
    jj_ctx* c = nullptr;
    const int rc = jj_ctx_create(devices[g], &c);
    if (rc) { for (jj_ctx* x : m->ctx) (void)jj_ctx_destroy(x); delete m; return rc; }
    m->ctx.push_back(c);
  }
  *out = m;
  return JJ_OK;
}
JJ_API int jj_multi_destroy(jj_multi* m) {
  if (!m) return JJ_ERR_INVALID;
  for (jj_ctx* c : m->ctx) (void)jj_ctx_destroy(c);
  delete m;
  return JJ_OK;
}
JJ_API int jj_multi_device_count(jj_multi* m) { return m ? (int)m->ctx.size() : JJ_ERR_INVALID; }
JJ_API jj_ctx* jj_multi_ctx(jj_multi* m, int g) { return (m && g >= 0 && g < (int)m->ctx.size()) ? m->ctx[g] : nullptr; }
JJ_API const char* jj_multi_last_error(jj_multi* m) { return m ? m->err.c_str() : "null context"; }

static inline void shard_of(size_t n, int g, int G, size_t* lo, size_t* hi) {
  const size_t base = n / G, rem = n % G;
  *lo = (size_t)g * base + std::min<size_t>((size_t)g, rem);
  *hi = *lo + base + ((size_t)g < rem ? 1 : 0);
}
// body(ctx, g, lo, hi) on one host thread per device; the first failing status (lowest device index) is returned
template <class Body>
static int multi_run(jj_multi* m, size_t n, Body body) {
  const int G = (int)m->ctx.size();
  std::vector<int> rc(G, JJ_OK);
  std::vector<std::thread> th;
  for (int g = 0; g < G; g++) th.emplace_back([&, g]() { size_t lo, hi; shard_of(n, g, G, &lo, &hi); rc[g] = body(m->ctx[g], g, lo, hi); });
  for (auto& t : th) t.join();
  for (int g = 0; g < G; g++) if (rc[g]) { std::lock_guard<std::mutex> lk(m->mu); m->err = "device shard " + std::to_string(g) + ": " + jj_last_error(m->ctx[g]); return rc[g]; }
  return JJ_OK;
}
// Page-locks whole caller buffers for the lifetime of one jj_multi_* call.  The per-device shards are cut at element, not page,
// boundaries: if every device thread registered its own sub-range, neighbouring shards would register the same page twice and the
// loser (hipErrorHostMemoryAlreadyRegistered) would silently fall back to synchronous pageable staging.  Registered once here,
// every shard finds its range pinned (run_pipelined / is_pinned_host) and none registers anything.
struct MultiPin {
  std::vector<void*> locked;
  void add(const void* p, size_t bytes) {
    if (!p || bytes < REGISTER_MIN_BYTES || is_pinned_host(p, bytes)) return;         // smaller arrays: through each context's staging slots
    if (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) == hipSuccess) locked.push_back(const_cast<void*>(p));
    else (void)hipGetLastError();                                                    // not fatal: the shards fall back to staging
  }
  ~MultiPin() { for (void* p : locked) (void)hipHostUnregister(p); }
};
static bool host_args(jj_multi* m, std::initializer_list<const void*> ptrs, size_t n) {
  if (n == 0) return true;
  for (const void* p : ptrs) if (!p || is_device_ptr(p)) { std::lock_guard<std::mutex> lk(m->mu); m->err = "multi-device entry points take host pointers"; return false; }
  return true;
}
#define U8(p) ((const uint8_t*)(p))
#define U8W(p) ((uint8_t*)(p))
JJ_API int jj_multi_varbase_mul(jj_multi* m, size_t n, const void* scalars, const void* points, void* out64) {
  if (!m || !host_args(m, {scalars, points, out64}, n)) return JJ_ERR_INVALID;
  (void)hipSetDevice(m->ctx[0]->device);
  MultiPin pin; pin.add(scalars, 32 * n); pin.add(points, 64 * n); pin.add(out64, 64 * n);
  return multi_run(m, n, [&](jj_ctx* c, int, size_t lo, size_t hi) { return jj_varbase_mul(c, hi - lo, U8(scalars) + 32 * lo, U8(points) + 64 * lo, U8W(out64) + 64 * lo); });
}
JJ_API int jj_multi_fixedbase_table_create(jj_multi* m, const void* base64, int window_bits, jj_mtable** out) {
  if (!m || !out || !base64) return JJ_ERR_INVALID;
  *out = nullptr;
  jj_mtable* mt = new jj_mtable();
  mt->t.assign(m->ctx.size(), nullptr);
  const int rc = multi_run(m, m->ctx.size(), [&](jj_ctx* c, int g, size_t, size_t) { return jj_fixedbase_table_create(c, base64, window_bits, &mt->t[g]); });
  if (rc) { for (size_t g = 0; g < mt->t.size(); g++) if (mt->t[g]) (void)jj_fixedbase_table_destroy(m->ctx[g], mt->t[g]); delete mt; return rc; }
  *out = mt;
  return JJ_OK;
}
JJ_API int jj_multi_fixedbase_table_destroy(jj_multi* m, jj_mtable* mt) {
  if (!m || !mt || mt->t.size() != m->ctx.size()) return JJ_ERR_INVALID;
  for (size_t g = 0; g < mt->t.size(); g++) if (mt->t[g]) (void)jj_fixedbase_table_destroy(m->ctx[g], mt->t[g]);
  delete mt;
  return JJ_OK;
}
JJ_API int jj_multi_fixedbase_mul(jj_multi* m, const jj_mtable* mt, size_t n, const void* scalars, void* out64) {
  if (!m || !mt || mt->t.size() != m->ctx.size() || !host_args(m, {scalars, out64}, n)) return JJ_ERR_INVALID;
  (void)hipSetDevice(m->ctx[0]->device);
  MultiPin pin; pin.add(scalars, 32 * n); pin.add(out64, 64 * n);
  return multi_run(m, n, [&](jj_ctx* c, int g, size_t lo, size_t hi) { return jj_fixedbase_mul(c, mt->t[g], hi - lo, U8(scalars) + 32 * lo, U8W(out64) + 64 * lo); });
}
JJ_API int jj_multi_decompress(jj_multi* m, size_t n, const void* in32, unsigned flags, void* out64, uint8_t* ok) {
  if (!m || !host_args(m, {in32, out64, ok}, n)) return JJ_ERR_INVALID;
  (void)hipSetDevice(m->ctx[0]->device);
  MultiPin pin; pin.add(in32, 32 * n); pin.add(out64, 64 * n); pin.add(ok, n);
  return multi_run(m, n, [&](jj_ctx* c, int, size_t lo, size_t hi) { return jj_decompress(c, hi - lo, U8(in32) + 32 * lo, flags, U8W(out64) + 64 * lo, ok + lo); });
}
// Last step of an MSM that was cut across devices or processes: the sum of the `count` partial points (canonical
// affine, 64 bytes each, HOST memory: what jj_msm wrote on every device / what the ranks' all_gather delivered) -> one affine
// point.  Runs on the calling host thread with the arithmetic of the MSM's own host tail (jj_host_tail.h): a chain of `count`
// dependent additions and one inversion takes a few microseconds there and ~180 us as GPU launches (jj_point_sum).
JJ_API int jj_msm_fold_partials(size_t count, const void* parts64, void* out64) {
  if (!out64 || (count && !parts64) || is_device_ptr(out64) || (count && is_device_ptr(parts64))) return JJ_ERR_INVALID;
  jjhost::Ext total = jjhost::identity();
  for (size_t g = 0; g < count; g++) {
    const uint8_t* src = U8(parts64) + 64 * g;
    jjhost::Ext p;
    p.u = jjhost::from_canon(src); p.v = jjhost::from_canon(src + 32);
    p.z = jjhost::consts().one; p.t1 = p.u; p.t2 = p.v;
    total = jjhost::point_add(total, p);
  }
  jjhost::to_affine64((uint8_t*)out64, total);
  return JJ_OK;
}
// sum over ALL terms: every device reduces its shard to a record of partial window sums (one per pass of 2^24 terms), the
// records of all devices meet in ONE host tail: window sums, one Horner chain per window layout, one inversion
JJ_API int jj_multi_msm(jj_multi* m, size_t n, const void* scalars, const void* points, void* out64) {
  if (!m || !out64 || is_device_ptr(out64) || !host_args(m, {scalars, points}, n)) return JJ_ERR_INVALID;
  const int G = (int)m->ctx.size();
  // every device: the passes of its shard as ONE job (msm_begin_locked: shards of 2^19 terms and more are cut so that the copy of a pass
  // runs beside the kernels of the pass before), its records collected here; one host tail over the records of all devices
  std::vector<std::vector<uint8_t>> recs(G);
  (void)hipSetDevice(m->ctx[0]->device);
  MultiPin pin; pin.add(scalars, 32 * n); pin.add(points, 64 * n);
  const int rc = multi_run(m, n, [&](jj_ctx* c, int g, size_t lo, size_t hi) -> int {
    jj_msm_job* j = nullptr;
    {
      JJ_ENTER(c);
      const int r2 = msm_begin_locked(c, hi - lo, U8(scalars) + 32 * lo, U8(points) + 64 * lo, 0, 1, false, &j);
      if (r2) return r2;
    }
    const hipError_t e = hipEventSynchronize(j->ev);
    if (e == hipSuccess) recs[g].assign(j->host, j->host + j->nrec * jjhost::REC_MAX_BYTES);
    JJ_ENTER(c);
    msm_job_put(c, j);
    if (e != hipSuccess) { c->err = std::string("hipEventSynchronize failed: ") + hipGetErrorString(e); return (int)JJ_ERR_HIP; }
    return (int)JJ_OK;
  });
  if (rc) return rc;
  std::vector<uint8_t> all;
  for (int g = 0; g < G; g++) all.insert(all.end(), recs[g].begin(), recs[g].end());
  static_assert(JJ_MSM_PARTIAL_BYTES == jjhost::REC_MAX_BYTES, "record size");
  return jj_msm_combine(all.size() / JJ_MSM_PARTIAL_BYTES, all.data(), out64);
}
