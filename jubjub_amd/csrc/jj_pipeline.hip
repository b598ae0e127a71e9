// libjubjub_hip.so: context, streams, staging of host arguments, the host-buffer pipeline's machinery, page-locked host memory.
#define JJ_KERNELS_PROBE
#include "jj_engine.h"

void prof_mark(jj_ctx* c, int which) {
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

int ensure(jj_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return JJ_OK;
  if (b.p) { HIPCHK(c, hipDeviceSynchronize()); HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }   // growth only; the buffer may be in use on any of the context's streams
  size_t want = std::max(bytes, (size_t)4096);
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) { c->err = std::string("hipMalloc failed: ") + hipGetErrorString(e); b.p = nullptr; return JJ_ERR_NOMEM; }
  b.cap = want;
  return JJ_OK;
}

bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// Resolves an input pointer: device pointers pass through (must be 16-byte aligned), host data is copied into a
// staging buffer (large pageable arrays through the page-locked staging slots, see host_to_dev_bounced).
int stage_in(jj_ctx* c, int slot, const void* p, size_t bytes, const void** dev) {
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
int stage_out(jj_ctx* c, DevBuf& buf, void* p, size_t bytes, OutRef* o) {
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
int finish_out(jj_ctx* c, const OutRef& o, bool* need_sync) {
  if (o.host) {
    if (o.bytes >= BOUNCE_THRESHOLD && !is_pinned_host(o.user, o.bytes)) { const int rc = dev_to_host_bounced(c, o.user, o.dev, o.bytes); if (rc) return rc; }
    else if (o.bytes) HIPCHK(c, hipMemcpyAsync(o.user, o.dev, o.bytes, hipMemcpyDeviceToHost, c->stream));
    *need_sync = true;
  }
  return JJ_OK;
}
int finish(jj_ctx* c, bool need_sync) {
  HIPCHK(c, hipGetLastError());
  if (need_sync) HIPCHK(c, hipStreamSynchronize(c->stream));
  return JJ_OK;
}

// ---------------------------------------------------------------------------------------------------- host-buffer pipeline
// When every array argument is a host pointer and the batch is large, the caller's buffers are page-locked in place
// (hipHostRegister: ~1 ms per 100 MB, measured) and the batch is cut into chunks that flow over two copy streams
// while the kernels of the neighbouring chunk run:  H2D (h2d stream) -> kernels (compute stream) -> D2H (d2h stream),
// two device slots, all ordering by events (no host synchronisation inside the loop, no CPU bounce copies).
// If registration fails (e.g. overlapping or already registered buffers) the caller falls back to plain staging.
int pipe_prepare(jj_ctx* c, size_t in_bytes, size_t out_bytes) {
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
bool is_pinned_host(const void* p, size_t bytes) {
  if (!p || !bytes) return false;
  for (const uint8_t* q : {(const uint8_t*)p, (const uint8_t*)p + bytes - 1}) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a.type != hipMemoryTypeHost) return false;
  }
  return true;
}
bool all_host(std::initializer_list<const void*> ptrs) { for (const void* p : ptrs) if (!p || is_device_ptr(p)) return false; return true; }

// A result array the caller has just allocated (calloc / vec![0; n] / np.empty) has no pages yet: page-locking it makes the kernel fault
// every page in, one after the other, inside hipHostRegister -- 12 ms per 100 MB on the box measured (profiles/r4_pcie_probe.txt: 123 ms
// for the 1 GB result of a 2^24-unit fixed-base call, four times the call's own 31 ms).  Touching one byte per page from several
// threads first (read and write back the same value: the array's contents, if any, stay) spreads the faults over the cores.
void prefault_parallel(void* p, size_t bytes) {
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
size_t pipe_chunk_for(const jj_ctx* c, size_t n, int pref_log2, size_t quantum) {
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
std::vector<size_t> pipe_chunk_bounds(size_t n, size_t CH, size_t quantum, bool ramp) {
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
size_t msm_host_pass_terms(size_t n, int pass_log2, bool split) {
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
int pipe_to_tail(jj_ctx* c) {
  if (!c->pipe_tail || c->stream == c->pipe_tail) return JJ_OK;
  HIPCHK(c, hipEventRecord(c->pipe_tail_ev, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->pipe_tail, c->pipe_tail_ev, 0));
  c->stream = c->pipe_tail;
  return JJ_OK;
}
// page-locked staging of the bounce path: three slots each way, grown on demand
int stage_ensure(jj_ctx* c, size_t in_bytes, size_t out_bytes) {
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
int host_to_dev_bounced(jj_ctx* c, void* dev, const void* host, size_t bytes, hipStream_t stream, size_t* seq) {
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
int stage_in_drain(jj_ctx* c, size_t seq) {
  for (size_t j = (seq > 3 ? seq - 3 : 0); j < seq; j++) HIPCHK(c, hipEventSynchronize(c->ev_stage[j % 3]));
  return JJ_OK;
}
int dev_to_host_bounced(jj_ctx* c, void* host, const void* dev, size_t bytes) {
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
// HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A context uses three streams in its host-
// buffer pipeline (compute, copy in, copy out) and one more per extra MSM lane; beside PyTorch's or the caller's own streams that
// exceeds four, and two streams that share a hardware queue serialise -- measured: jj_multi_* with a second context in the process
// 268 -> 523 M fixed-base scalar-muls/s, a fourth pipeline stream 316 -> 520 M/s (profiles/r4_pcie_inclusive.txt).  The runtime reads
// the variable when it initialises (first HIP call).  It is the APPLICATION's to set (bench.py does): rounds 3-4 set it
// from a load-time constructor here, which changed the HIP configuration of the whole host process behind its back.

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
// [p, p + bytes) consists of whole pages: both ends page-aligned.  Only such a range can be page-locked for the GPU without handing it pages that
// belong to other objects of the process (the first and last page of an unaligned range do).
bool owns_its_pages(const void* p, size_t bytes) {
  const uintptr_t pg = (uintptr_t)sysconf(_SC_PAGESIZE);
  return p && bytes && ((uintptr_t)p & (pg - 1)) == 0 && (bytes & (pg - 1)) == 0;
}
JJ_API int jj_host_register(void* p, size_t bytes) {
  if (!p || !bytes) return JJ_ERR_INVALID;
  // whole pages only (both ends page-aligned): the buffer owns every page it touches.  An aligned start alone is not enough -- aligned_alloc(4096, 5000)
  // shares its LAST page with the next heap object -- and page-locking pages that belong to neighbouring objects, then releasing them,
  // is what the GPU memory faults of round 4 followed (experiments/hsa_stale_mapping/).
  if (!owns_its_pages(p, bytes)) return JJ_ERR_INVALID;
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

// ---- result pool: library-owned page-locked result buffers (include/jubjub_hip.h)
JJ_API int jj_result_acquire(jj_ctx* c, size_t bytes, void** out) {
  if (!c || !out) return JJ_ERR_INVALID;
  *out = nullptr;
  if (bytes == 0) return JJ_OK;
  JJ_ENTER(c);
  int best = -1;
  for (size_t i = 0; i < c->result_pool.size(); i++) {           // the smallest free buffer that is large enough (and not more than twice too large)
    const auto& b = c->result_pool[i];
    if (!b.in_use && b.cap >= bytes && b.cap <= 2 * bytes + ((size_t)2 << 20) && (best < 0 || b.cap < c->result_pool[best].cap)) best = (int)i;
  }
  if (best >= 0) { c->result_pool[best].in_use = true; *out = c->result_pool[best].p; return JJ_OK; }
  const size_t cap = (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1);          // whole 2 MB
  uint8_t* p = nullptr;
  if (hipHostMalloc((void**)&p, cap, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    // make room: drop the free buffers, try once more
    for (size_t i = 0; i < c->result_pool.size();) { if (!c->result_pool[i].in_use) { (void)hipHostFree(c->result_pool[i].p); c->result_pool.erase(c->result_pool.begin() + i); } else i++; }
    if (hipHostMalloc((void**)&p, cap, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); c->err = "jj_result_acquire: hipHostMalloc failed"; return JJ_ERR_NOMEM; }
  }
  c->result_pool.push_back({p, cap, true});
  *out = p;
  return JJ_OK;
}
JJ_API int jj_result_release(jj_ctx* c, void* p) {
  if (!c) return JJ_ERR_INVALID;
  if (!p) return JJ_OK;
  JJ_ENTER(c);
  size_t held = 0;
  int at = -1;
  for (size_t i = 0; i < c->result_pool.size(); i++) { held += c->result_pool[i].cap; if (c->result_pool[i].p == p && c->result_pool[i].in_use) at = (int)i; }
  if (at < 0) { c->err = "jj_result_release: not a buffer this context handed out (or released twice)"; return JJ_ERR_INVALID; }
  // copies into the buffer were waited for by the call that produced them; a buffer the caller passed as an INPUT to a call that is still queued must
  // not be released before jj_ctx_sync
  if (held > c->result_pool_keep) { (void)hipHostFree(c->result_pool[at].p); c->result_pool.erase(c->result_pool.begin() + at); }
  else c->result_pool[at].in_use = false;
  return JJ_OK;
}
JJ_API int jj_result_pool_stats(jj_ctx* c, size_t* buffers, size_t* bytes, size_t* in_use) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  size_t nb = 0, by = 0, iu = 0;
  for (const auto& b : c->result_pool) { nb++; by += b.cap; iu += b.in_use ? 1 : 0; }
  if (buffers) *buffers = nb;
  if (bytes) *bytes = by;
  if (in_use) *in_use = iu;
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

// ---- options (jj_ctx_set_option).  The library reads NO environment variable of its own: what rounds 2-5 took from JJ_* variables is set per context,
// by key, through the C ABI -- and nothing here can change the timing discipline of an entry point (the constant-time ladders and selects have no switch).
struct CtxOption { const char* key; long long lo, hi; void (*set)(jj_ctx*, long long); long long (*get)(const jj_ctx*); };
#define JJ_OPT(key, lo, hi, field, type) {key, lo, hi, [](jj_ctx* c, long long v) { c->field = (type)v; }, [](const jj_ctx* c) { return (long long)c->field; }}
static const CtxOption* ctx_options() {
  static const CtxOption table[] = {
    // what a caller may legitimately need
    JJ_OPT("msm_lanes", 1, MSM_LANES_MAX, msm_lanes, int),                   // streams the jobs in flight alternate over (1: every job on the context's stream)
    JJ_OPT("msm_fold_min", 2, 4096, msm_fold_min, int),                      // gathered records are folded on the device from this many
    JJ_OPT("msm_fold_dev", 0, 1, msm_fold_dev, bool),                        // 0: every gathered record is copied and the host adds them
    JJ_OPT("msm_host_split", 0, 1, msm_host_split, bool),                    // host arrays of 2^19+ terms in several passes, copies beside kernels
    JJ_OPT("msm_pass_log2", 10, 24, msm_pass_log2, int),                     // terms per Pippenger pass
    {"result_pool_mb", 0, 1 << 20, [](jj_ctx* c, long long v) { c->result_pool_keep = (size_t)v << 20; }, [](const jj_ctx* c) { return (long long)(c->result_pool_keep >> 20); }},
    JJ_OPT("torsion_check_ladder", 0, 1, torsion_ladder, bool),              // subgroup test by [r]P (the reference's definition) instead of the pairing
    {"pipe_pageable_register", 0, 1, [](jj_ctx* c, long long v) { c->pipe_bounce = v == 0; }, [](const jj_ctx* c) { return (long long)!c->pipe_bounce; }},
    JJ_OPT("pipe_copy_threads", 0, 64, pipe_copy_threads, int),
    JJ_OPT("pipe_ramp", 0, 1, pipe_ramp, bool),
    JJ_OPT("pipe_prefault", 0, 1, pipe_prefault, bool),
    {"pipe_chunk_log2", 0, 24, [](jj_ctx* c, long long v) { c->pipe_chunk = v >= 8 ? (size_t)1 << v : 0; }, [](const jj_ctx* c) { long long l = 0; while (((size_t)1 << l) < c->pipe_chunk) l++; return c->pipe_chunk ? l : 0LL; }},
    {"fixedbase_default", 6, 7, [](jj_ctx* c, long long v) { c->fb_default_kind = (int)v; }, [](const jj_ctx* c) { return (long long)c->fb_default_kind; }},
    // planner overrides (tests and measurements; every value gives the same results)
    JJ_OPT("msm_windows", 0, MSM_WINDOWS_MAX, msm_windows, int),             // 0 = from n; else 16..36
    JJ_OPT("msm_small_max", 0, 1 << 20, msm_small_max, int),
    JJ_OPT("msm_small_blk", 1, MSM_SMALL_BLK_MAX, msm_small_blk, int),
    JJ_OPT("msm_accum", -1, 1, msm_segments, int),                           // 1 = length-sorted segments, 0 = chunks + fix-up, -1 = by size
    JJ_OPT("msm_seg_len", 0, 1024, msm_seg_len, int),
    JJ_OPT("msm_chunk", 0, 1024, msm_chunk, int),
    JJ_OPT("msm_chunk_waves", 1, 8, msm_chunk_waves, int),
    JJ_OPT("msm_sort_blocks_per_cu", 1, 4, msm_sort_blocks_per_cu, int),
    JJ_OPT("msm_reduce_chunk", 0, 256, msm_reduce_chunk, int),
    JJ_OPT("msm_reduce_l1", -1, 64, msm_l1_rows, int),
    JJ_OPT("msm_reduce_l2_chunk", 0, 64, msm_l2_chunk, int),
    JJ_OPT("msm_sort_hist_fused", 0, 1, msm_fused_hist, bool),
    JJ_OPT("msm_sort_two_pass", -1, 1, msm_two_pass, int),
    JJ_OPT("msm_front1", 0, 1, msm_front1, bool),
    JJ_OPT("msm_acc_lds", 0, 1, msm_acc_lds, bool),
    JJ_OPT("vb_ct_window", 2, 3, vb_ct_window, int),
    JJ_OPT("vb_quad_max", 0, 1 << 20, vb_quad_max, int),
    JJ_OPT("dec_c_mid", 8, 16, dec_c_mid, int),
    {nullptr, 0, 0, nullptr, nullptr}};
  return table;
}
#undef JJ_OPT
static int ctx_option_apply(jj_ctx* c, const CtxOption* o, long long v) {
  const auto pow2 = [](long long x) { return x > 0 && (x & (x - 1)) == 0; };
  if (v < o->lo || v > o->hi) return JJ_ERR_INVALID;
  const std::string k = o->key;
  if (k == "msm_windows" && v != 0 && v < MSM_WINDOWS_MIN) return JJ_ERR_INVALID;
  if (k == "msm_seg_len" && v != 0 && v < 8) return JJ_ERR_INVALID;
  if (k == "msm_chunk" && v != 0 && v < 8) return JJ_ERR_INVALID;
  if ((k == "msm_reduce_chunk" || k == "msm_reduce_l2_chunk") && v != 0 && (v < 2 || !pow2(v))) return JJ_ERR_INVALID;
  if (k == "msm_reduce_l1" && v > 0 && (v < 2 || !pow2(v))) return JJ_ERR_INVALID;
  if (k == "dec_c_mid" && v != 8 && v != 16) return JJ_ERR_INVALID;
  if (k == "pipe_chunk_log2" && v != 0 && v < 8) return JJ_ERR_INVALID;
  o->set(c, v);
  return JJ_OK;
}
// key == "host_tail_scalar" with ctx == NULL is the one process-wide option (the host tail of jj_msm_combine has no context)
JJ_API int jj_ctx_set_option(jj_ctx* c, const char* key, long long value) {
  if (!key) return JJ_ERR_INVALID;
  if (!strcmp(key, "host_tail_scalar")) { if (value != 0 && value != 1) return JJ_ERR_INVALID; jjhost::ifma::force_scalar().store((int)value); return JJ_OK; }
  if (!c) return JJ_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  for (const CtxOption* o = ctx_options(); o->key; o++) if (!strcmp(o->key, key)) {
    const int rc = ctx_option_apply(c, o, value);
    if (rc) c->err = std::string("jj_ctx_set_option: value out of range for ") + key + " (" + std::to_string(o->lo) + " .. " + std::to_string(o->hi) + ")";
    return rc;
  }
  c->err = std::string("jj_ctx_set_option: unknown key ") + key;
  return JJ_ERR_INVALID;
}
JJ_API int jj_ctx_get_option(jj_ctx* c, const char* key, long long* value) {
  if (!key || !value) return JJ_ERR_INVALID;
  if (!strcmp(key, "host_tail_scalar")) { *value = jjhost::ifma::force_scalar().load(); return JJ_OK; }
  if (!c) return JJ_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  for (const CtxOption* o = ctx_options(); o->key; o++) if (!strcmp(o->key, key)) { *value = o->get(c); return JJ_OK; }
  c->err = std::string("jj_ctx_get_option: unknown key ") + key;
  return JJ_ERR_INVALID;
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
#ifdef JJ_EXPERIMENTS
  // Probe builds only (-DJJ_EXPERIMENTS, tools/ and experiments/): every option of jj_ctx_set_option can be preset as JJ_<KEY>, and the
  // switches that change the TIMING DISCIPLINE of an entry point exist here and nowhere else.  The shipped library reads no JJ_* variable.
  for (const CtxOption* o = ctx_options(); o->key; o++) {
    std::string name = "JJ_";
    for (const char* q = o->key; *q; q++) name += (char)toupper((unsigned char)*q);
    if (const char* e = getenv(name.c_str())) { if (ctx_option_apply(c, o, atoll(e)) != JJ_OK) fprintf(stderr, "libjubjub_hip: %s=%s ignored (valid: %lld..%lld)\n", name.c_str(), e, (long long)o->lo, (long long)o->hi); }
  }
  if (const char* e = getenv("JJ_PIPE_STREAMS")) { int v = atoi(e); if (v >= 1 && v <= 3) c->pipe_mode = v; }      // =2 is 1.7x slower, =3 equals the default: jj_engine.h pipe_mode
  if (const char* e = getenv("JJ_FB_GATHER_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 8) c->fb_gather_blocks_per_cu = v; }
  if (const char* e = getenv("JJ_VB_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 8) c->vb_blocks_per_cu = v; }
  if (const char* e = getenv("JJ_VARBASE_DEFAULT")) c->vb_default_ct = strcmp(e, "vartime") != 0;          // A/B only: jj_varbase_mul takes the table ladder
  if (const char* e = getenv("JJ_FIXEDBASE_SELECT")) c->fb_const_time = strcmp(e, "gather") != 0;         // A/B only: per-lane LDS gather instead of the shuffle select
#endif
  { const int rc = jj_batch_init(c); if (rc) return fail(rc); }       // LDS carve-outs of the fixed-base kernels, square-root tables (jj_abi.hip)
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
  for (jj_msm_job* j : c->job_pool) { if (j->host) (void)hipHostFree(j->host); if (j->gdev) (void)hipFree(j->gdev); (void)hipEventDestroy(j->ev); delete j; }
  for (MsmLane& L : c->lanes) {
    if (L.owned) (void)hipStreamSynchronize(L.stream);
    DevBuf* lb[] = {&L.buf[0], &L.buf[1], &L.buf[2], &L.buf[3], &L.buf[4], &L.buf[5], &L.buf[6], &L.buf[7], &L.ctl, &L.bigpart, &L.seg, &L.rec, &L.bins};
    for (DevBuf* b : lb) if (b->p) (void)hipFree(b->p);
    if (L.owned) {
      (void)hipEventDestroy(L.ready_ev);
      (void)hipStreamDestroy(L.stream);
    }
  }
  for (DevBuf* b : all) if (b->p) (void)hipFree(b->p);
  delete c->copy_pool;
  for (int i = 0; i < 3; i++) { if (c->stage_in[i]) (void)hipHostFree(c->stage_in[i]); if (c->stage_out[i]) (void)hipHostFree(c->stage_out[i]); if (c->ev_stage[i]) (void)hipEventDestroy(c->ev_stage[i]); }
  for (auto& b : c->result_pool) (void)hipHostFree(b.p);
  if (c->gather_dev.p) (void)hipFree(c->gather_dev.p);
  if (c->poison_dev.p) (void)hipFree(c->poison_dev.p);
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
int switch_stream(jj_ctx* c, hipStream_t s) {
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

