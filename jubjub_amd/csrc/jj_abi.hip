// libjubjub_hip.so: the batch entry points of the C ABI (include/jubjub_hip.h) and the launches of their kernels (jj_kernels.h).
#define JJ_KERNELS_BATCH
#include "jj_engine.h"

// Called once by jj_ctx_create (jj_pipeline.hip), with the context's device selected: what the batch kernels need before their first launch.
// On failure the caller releases the context (including sqrt_tabs).
int jj_batch_init(jj_ctx* c) {
  // the fixed-base kernels need the full 160 KiB LDS carve-out
  const struct { const void* fn; int bytes; } lds_needs[] = {
    {reinterpret_cast<const void*>(k_fixedbase<true>), FB_LDS_BYTES}, {reinterpret_cast<const void*>(k_fixedbase<false>), FB_LDS_BYTES},
    {reinterpret_cast<const void*>(k_fixedbase_comb<true>), FBC_LDS_BYTES}, {reinterpret_cast<const void*>(k_fixedbase_comb<false>), FBC_LDS_BYTES},
    {reinterpret_cast<const void*>(k_varbase_ct3), CT3_LDS_BYTES_PER_BLOCK}};
  for (const auto& a : lds_needs)
    if (hipFuncSetAttribute(a.fn, hipFuncAttributeMaxDynamicSharedMemorySize, a.bytes) != hipSuccess) return JJ_ERR_HIP;   // the kernels could not launch later
  // square-root tables (64 KiB dlog + 36 KiB powers), built on the device
  if (hipMalloc(&c->sqrt_tabs.p, 65536 + 4 * 256 * NL * 4) != hipSuccess) { c->sqrt_tabs.p = nullptr; return JJ_ERR_NOMEM; }
  c->sqrt_tabs.cap = 65536 + 4 * 256 * NL * 4;
  if (hipMemsetAsync(c->sqrt_tabs.p, 0, c->sqrt_tabs.cap, c->stream) != hipSuccess) return JJ_ERR_HIP;
  c->sqrt_tables.dlog = (const uint8_t*)c->sqrt_tabs.p;
  c->sqrt_tables.npow = (const u32*)((uint8_t*)c->sqrt_tabs.p + 65536);
  hipLaunchKernelGGL(k_sqrt_tables_init, dim3(5), dim3(256), 0, c->stream, (uint8_t*)c->sqrt_tabs.p, (u32*)((uint8_t*)c->sqrt_tabs.p + 65536));
  if (hipStreamSynchronize(c->stream) != hipSuccess) return JJ_ERR_HIP;
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
// ct: the ladder with the reference's timing discipline (no scalar-dependent address or branch): k_varbase_ct3 / k_varbase_ct_quad; otherwise the
// per-lane window table in memory (k_varbase / k_varbase_quad: digit-dependent addresses)
static int varbase_to_ext(jj_ctx* c, size_t n, const void* ds, const void* dp, SoA ext, bool five, bool shared_scalar = false, bool ct = false) {
  if (ct && !five && !shared_scalar) {
    if (n <= (size_t)c->vb_quad_max) hipLaunchKernelGGL(k_varbase_ct_quad, dim3(blocks_for(4 * n)), dim3(256), 0, c->stream, n, ds, dp, ext);   // small batch: one scalar multiplication per quad of lanes
    else if (c->vb_ct_window == 3) hipLaunchKernelGGL(k_varbase_ct3, dim3(blocks_for(n)), dim3(256), CT3_LDS_BYTES_PER_BLOCK, c->stream, n, ds, dp, ext);
    else hipLaunchKernelGGL(k_varbase_ct, dim3(blocks_for(n)), dim3(256), 0, c->stream, n, ds, dp, ext);
    return JJ_OK;
  }
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
static int varbase_api(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out, int mode, bool ct) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  if (const size_t ch = pipe_chunk_for(c, n, 18); ch && all_host({scalars, points, out})) {
    const HostIn in[2] = {{scalars, 32}, {points, 64}};
    const HostOut ho[1] = {{out, (size_t)(mode ? 32 : 64)}};
    const int prc = run_pipelined(c, n, ch, in, ho, [&](size_t cn, const void* const* di, void* const* dout) -> int {
      int rc2;
      if ((rc2 = ensure_ext(c, cn, 3))) return rc2;
      SoA ext = soa_of(c->ws->ext, cn);
      if ((rc2 = varbase_to_ext(c, cn, di[0], di[1], ext, false, false, ct))) return rc2;
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
    if ((rc = varbase_to_ext(c, n, ds, dp, ext, false, false, ct))) return rc;
    prof_mark(c, 1);
    if ((rc = normalize_launch(c, n, ext, o.dev, mode))) return rc;
    prof_mark(c, 2);
  }
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
// ExtendedPoint * Fr (reference src/lib.rs:873-879 -> 357-379): the reference's ladder is constant-time (conditional_select, 334-343), and so is
// the default here (round 5): signed 3-bit windows, mask selects, no table in memory (k_varbase_ct3; one scalar multiplication per quad of lanes up
// to JJ_VB_QUAD_MAX units: k_varbase_ct_quad).  The _vartime entry points keep the per-lane window table in memory and signed 5-bit windows
// (digit-dependent addresses): 1.6-4.5 % faster at 2^20 units depending on the box (profiles/r5_vb_ct_window.txt: ratio 0.984, r6_vb_ct_window.txt: 0.955), for public scalars.
JJ_API int jj_varbase_mul(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out) { return varbase_api(c, n, scalars, points, out, 0, c && c->vb_default_ct); }
JJ_API int jj_varbase_mul_compressed(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out32) { return varbase_api(c, n, scalars, points, out32, 1, c && c->vb_default_ct); }
JJ_API int jj_varbase_mul_ct(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out) { return varbase_api(c, n, scalars, points, out, 0, true); }      // (the name rounds 3-4 gave the opt-in; always constant-time)
JJ_API int jj_varbase_mul_vartime(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out) { return varbase_api(c, n, scalars, points, out, 0, false); }
JJ_API int jj_varbase_mul_vartime_compressed(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out32) { return varbase_api(c, n, scalars, points, out32, 1, false); }
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

