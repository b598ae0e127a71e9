// libjubjub_hip.so: jj_multi_* -- several devices of one node driven from one process.
#include "jj_engine.h"

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
    if (!p || bytes < REGISTER_MIN_BYTES || !owns_its_pages(p, bytes) || is_pinned_host(p, bytes)) return;         // arrays that share pages with other objects, and small ones: through each context's staging slots
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

