/*
 * jubjub_hip.h — C ABI of libjubjub_hip.so, the MI355X-native batched Jubjub engine.
 *
 * This is the drop-in boundary for the scalar-multiplication hot path of the zkcrypto/jubjub crate
 * (reference: /root/reference, crate jubjub 0.10.0).  The reference has no FFI; its boundary is its public
 * Rust API.  Rust struct layouts are unspecified, so the only stable representations — and the wire formats
 * here — are the ones the reference itself exposes publicly:
 *
 *   scalar            32 bytes, little-endian integer      Fr::to_bytes / from_bytes      src/fr.rs:268-308
 *                     (ladder entry points take the raw 32-byte bit pattern, top 4 bits ignored:
 *                      ExtendedPoint::multiply / multiply_bits  src/lib.rs:272-301, 357-385, 831-833)
 *   base field elem   32 bytes, canonical little-endian     Fq::to_bytes / from_bytes      src/lib.rs:456-457, 500
 *   affine point      64 bytes = u || v, each canonical LE  AffinePoint::get_u/get_v       src/lib.rs:630-637,
 *                                                           from_raw_unchecked             src/lib.rs:662-664
 *   compressed point  32 bytes, v with sign(u) in bit 255   AffinePoint::to_bytes          src/lib.rs:455-464
 *   validity          one uint8_t per element (1 = Some, 0 = None); output zeroed when 0   (CtOption / Choice)
 *
 * Pointers may be host or device pointers (detected per pointer with hipPointerGetAttributes); host data is
 * staged through the context's buffers.  With device pointers the call is asynchronous on the context's
 * stream; with any host pointer it returns after the results are in host memory.  A context belongs to one
 * device; its entry points may be called from several host threads (they serialise on a per-context lock), and a
 * change of launch stream is ordered after the work queued on the previous stream (the context's workspaces are shared).
 * jj_multi_* below drives several devices of one node from one process.
 *
 * Every function returns 0 on success or a negative jj_status.
 */
#ifndef JUBJUB_HIP_H
#define JUBJUB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jj_ctx jj_ctx;
typedef struct jj_table jj_table;

typedef enum {
  JJ_OK = 0,
  JJ_ERR_INVALID = -1,   /* bad argument (null pointer, unknown op, length mismatch: cf. the assert at src/lib.rs:841) */
  JJ_ERR_HIP = -2,       /* a HIP runtime call failed; see jj_last_error() */
  JJ_ERR_NOMEM = -3,
  JJ_ERR_NODEVICE = -4   /* no gfx950 device visible — there is no CPU fallback */
} jj_status;

/* ---- context ------------------------------------------------------------------------------------------- */
int jj_ctx_create(int device, jj_ctx** out);
int jj_ctx_destroy(jj_ctx* ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream; NULL = HIP's default stream).  A new context
 * starts on its own non-blocking stream; jj_ctx_use_own_stream() returns to it. */
int jj_ctx_set_stream(jj_ctx* ctx, void* hip_stream);
int jj_ctx_use_own_stream(jj_ctx* ctx);
int jj_ctx_sync(jj_ctx* ctx);
/* Options.  The library reads NO environment variable: what a caller may tune is set per context, by key, right after jj_ctx_create (before the
 * first batch call).  No option changes WHAT an entry point computes or its timing discipline: the constant-time ladders and selects have no
 * switch (for public scalars there are the explicit *_vartime entry points).  Unknown key or value out of range: JJ_ERR_INVALID.
 *   msm_lanes 1..4 (3)            streams the jobs of jj_msm_begin / jj_msm_allgather_begin alternate over; 1 = every job on the context's stream
 *                                 (jj_ctx_set_comm with more than one rank sets 1: all gathers of one communicator then go through ONE stream)
 *   msm_fold_min 2..4096 (8)      jj_msm_allgather / _combine_dev fold the gathered records on the device from this many records
 *   msm_fold_dev 0|1 (1)          0: the gathered records are copied to the host and added there
 *   msm_host_split 0|1 (1)        host arrays of 2^19 terms and more in several passes, each copy beside the kernels of the pass before
 *   msm_pass_log2 10..24 (24)     terms per Pippenger pass
 *   result_pool_mb 0..2^20 (4096) bytes of released result buffers kept for reuse (jj_result_acquire)
 *   torsion_check_ladder 0|1 (0)  jj_is_torsion_free / _prime_order by [r]P (the reference's definition) instead of the order-8 pairing
 *   pipe_pageable_register 0|1 (0), pipe_copy_threads 0..64 (0 = auto), pipe_ramp 0|1 (1), pipe_prefault 0|1 (1), pipe_chunk_log2 0|8..24 (0 = per
 *                                 entry point): the host-buffer pipeline
 *   fixedbase_default 6|7 (7)     what window_bits = 0 means for jj_fixedbase_table_create
 * Planner overrides, for tests and measurements (every value gives the same results): msm_windows, msm_small_max, msm_small_blk, msm_accum,
 * msm_seg_len, msm_chunk, msm_reduce_chunk, msm_reduce_l1, msm_reduce_l2_chunk, msm_sort_hist_fused, msm_sort_two_pass, msm_front1, msm_acc_lds, vb_ct_window,
 * vb_quad_max, dec_c_mid (ranges: jj_pipeline.hip ctx_options).
 * One process-wide option, set with ctx = NULL: host_tail_scalar 0|1 (0) -- the MSM host tail on the scalar 4 x 64-bit chain even where AVX-512 IFMA is there. */
int jj_ctx_set_option(jj_ctx* ctx, const char* key, long long value);
int jj_ctx_get_option(jj_ctx* ctx, const char* key, long long* value);
const char* jj_last_error(jj_ctx* ctx);
int jj_version(void);
/* WnafGroup::recommended_wnaf_for_num_scalars (src/lib.rs:1320-1335) */
int jj_recommended_wnaf_for_num_scalars(size_t num_scalars);
/* Device properties used for roofline accounting: out[0]=CU count, out[1]=clock kHz, out[2]=wavefront size. */
int jj_device_info(jj_ctx* ctx, int64_t out[4]);

/* Per-call kernel timing with HIP events on the launch stream (for roofline accounting).  While enabled, each
 * jj_varbase_mul / jj_fixedbase_mul call records (main kernel ms, normalise-tail ms); read drains the log. */
int jj_ctx_profile(jj_ctx* ctx, int enable);
int jj_ctx_profile_read(jj_ctx* ctx, int max, float* main_ms, float* tail_ms, int* count);
/* Integer-VALU roofline denominator, measured on this device: sustained v_mad_u64_u32 (32x32+64 -> 64 multiply-
 * accumulate) lane-operations per second. */
int jj_peak_imad32(jj_ctx* ctx, double* out_per_sec);                            /* median of five samples */
/* `count` (1..64) single timed launches after one warm-up launch, in launch order: the sustained clock moves by a few percent within a
 * session, so report median and spread (bench.py samples before and after its workload). */
int jj_peak_imad32_samples(jj_ctx* ctx, int count, double* out_per_sec);

/* ---- host buffers ---------------------------------------------------------------------------------------------------------
 * A drop-in caller (the Rust shim of INTEGRATION.md, examples/scalar_mul.c) hands HOST arrays to the entry points below.  Large
 * batches (>= 2^18 units: four chunks of 2^16 and up) of jj_varbase_mul(_compressed), jj_fixedbase_mul(_compressed) and jj_decompress are then cut into chunks
 * that flow over two copy streams while the kernels of the neighbouring chunk run.  That needs PAGE-LOCKED memory:
 *   - memory from jj_host_alloc, or memory registered once with jj_host_register, is used as it is: the copies run straight
 *     from and to it (2^24 fixed-base units: 0.90 of the device-resident rate, profiles/r4_pcie_inclusive.txt);
 *   - any other (pageable) array passes through page-locked staging buffers of the context, copied by a few host threads (default
 *     8, option pipe_copy_threads) beside the GPU's work: within 2-3 % of the page-locked rates, nothing of the caller's is registered,
 *     and a result array the caller has only just allocated costs no more than its page faults.  The GPU never touches the caller's
 *     pageable pages: arrays of 1 MB and more are not handed to the HIP runtime either (which would page-lock them itself).
 *     option pipe_pageable_register = 1 selects round 3's way instead for arrays that consist of whole pages (both ends page-aligned, 1 MB and
 *     more: page-locked in place for the call: no CPU copies, but a freshly allocated 1 GB result array then costs ~65 ms of serial page
 *     faults and pinning inside the call); all other arrays are staged in that mode too -- page-locking them in place would hand pages of
 *     neighbouring objects to the GPU as well (two GPU write faults in ~3000 randomised test rounds followed that).
 * Page-locked buffers that are reused across calls are the fastest arrangement and cost the host no copy threads.
 * These four functions need no context and no HIP headers on the caller's side.  jj_host_alloc: page-locked, visible to every
 * device of the node, *out = NULL for bytes = 0.  jj_host_register: p .. p + bytes must be mapped and stay mapped until
 * jj_host_unregister(p); registering overlapping ranges twice fails with JJ_ERR_INVALID.  The range must consist of WHOLE PAGES -- p
 * page-aligned AND bytes a multiple of the page size (JJ_ERR_INVALID otherwise): register mappings of your own (mmap; or a page-aligned
 * allocation whose length you rounded up to whole pages).  An array on the C heap shares its first and last page with other heap
 * objects -- so does aligned_alloc(4096, 5000) at its end -- and page-locking those pages for the GPU and releasing them again was
 * followed by GPU memory faults on later transfers.
 * HARDWARE QUEUES: HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A context's host pipeline uses
 * three streams (plus one per extra MSM lane); beside other streams of the process two of them may share a queue and then serialise (pipelines
 * 1.7-2x slower, profiles/r4_pcie_inclusive.txt).  Applications that drive host batches beside other HIP work should export
 * GPU_MAX_HW_QUEUES=8 before the first HIP call; the library does not touch the process environment. */
int jj_host_alloc(size_t bytes, void** out);
int jj_host_free(void* p);
int jj_host_register(void* p, size_t bytes);
int jj_host_unregister(void* p);
/* RESULT POOL: page-locked result buffers owned by the context, for callers whose API returns a NEW result per call -- the shape of
 * every batch function of the reference (`-> Vec<..>`: batch_from_bytes src/lib.rs:541-627, batch_normalize 1084-1107).  A fresh
 * pageable result array costs its page faults inside the call (2^24 fixed-base units: 195 M/s against 557 with reused page-locked
 * buffers, profiles/r4_bench_host_*_fresh.json); jj_result_acquire hands out a buffer of at least `bytes` bytes instead (allocated on
 * first use, recycled afterwards: the smallest free buffer that fits), the entry points recognise it as page-locked memory (copies run
 * straight into it), and jj_result_release gives it back when the caller has consumed the result -- several buffers may be out at a
 * time, so every call can return a DIFFERENT result object.  Released buffers are kept while the pool holds at most result_pool_mb (option)
 * (default 4096) megabytes, and freed with the context.  jj_result_release(NULL) is a no-op; a pointer the context did not hand out
 * is JJ_ERR_INVALID.  Thread-safe per context like every entry point. */
int jj_result_acquire(jj_ctx*, size_t bytes, void** out);
int jj_result_release(jj_ctx*, void* p);
int jj_result_pool_stats(jj_ctx*, size_t* buffers, size_t* bytes, size_t* in_use);
/* How a host batch is cut -- pure functions of their arguments (no context, no device: the CPU-side tests call them).
 * jj_plan_host_chunks: the chunk boundaries of a pipelined host batch of n >= 1 units with chunks of `chunk` units (what the library
 * picks per entry point or option pipe_chunk_log2 sets): chunk k = [bounds[k], bounds[k + 1]), *count = entries of bounds (chunks + 1).
 * ramp != 0: the first and the last chunk are chunk / 4 when the batch has at least four chunks of at least 2^18 units -- the first
 * copy in and the last copy out are the two transfers nothing overlaps; `quantum` (0 = none): the units one round of the kernel's
 * lanes takes, edges are whole multiples of it.  cap = 0 (bounds may be NULL) returns the count only.
 * jj_plan_msm_host_passes: terms per pass and number of passes of jj_msm over HOST arrays of n terms (2^pass_log2 terms per pass at
 * most, 24 by default; split != 0: arrays of 2^19 terms and more are cut into two to eight passes -- more when eight would exceed
 * 2^pass_log2 terms each -- whose copies overlap the kernels). */
int jj_plan_host_chunks(size_t n, size_t chunk, size_t quantum, int ramp, size_t* bounds, size_t cap, size_t* count);
int jj_plan_msm_host_passes(size_t n, int pass_log2, int split, size_t* pass_terms, size_t* passes);

/* ---- fields: Fq (base, = bls12_381::Scalar, src/lib.rs:62) and Fr (scalar, src/fr.rs) ------------------------ */
/* Elements are 32-byte little-endian integers; inputs are reduced mod p like from_raw (src/fr.rs:347-349),
 * outputs are canonical.  reference: add 638-647, sub 620-634, mul 592-616, neg 651-665, square 353-381,
 * double 261-263, invert 438-540 (ok=0 & out=0 for zero), sqrt 384-399 (Fr) / Tonelli-Shanks (Fq, bls12_381).
 * jj_fq_sqrt returns the root that ff 0.13's sqrt_tonelli_shanks is RECALLED to return (that crate is not part of the reference
 * tree and no reference test stores a raw Fq root), so WHICH of the two roots comes back is parity-UNVERIFIED; that it is a root
 * (or ok = 0 for a non-residue) is checked.  Point decompression does not depend on it: the encoding's sign bit picks the root
 * (src/lib.rs:518-520), and that path is pinned by the reference's vectors.
 * TIMING: jj_fq_sqrt / jj_fr_sqrt, jj_decompress and jj_random_points are VARIABLE-TIME in their input: the square root reads a
 * discrete-log table at an address derived from the element's bits (jj_kernels.h sqrt_pohlig).  The dependency's Fq::sqrt
 * (bls12_381, called at src/lib.rs:515) is constant-time; that does not matter for PUBLIC encodings (signature / note / proof
 * bytes), which is what these entry points are for -- do not feed them secret field elements. */
int jj_fq_add(jj_ctx*, size_t n, const void* a, const void* b, void* out);
int jj_fq_sub(jj_ctx*, size_t n, const void* a, const void* b, void* out);
int jj_fq_mul(jj_ctx*, size_t n, const void* a, const void* b, void* out);
int jj_fq_neg(jj_ctx*, size_t n, const void* a, void* out);
int jj_fq_square(jj_ctx*, size_t n, const void* a, void* out);
int jj_fq_double(jj_ctx*, size_t n, const void* a, void* out);
int jj_fq_invert(jj_ctx*, size_t n, const void* a, void* out, uint8_t* ok);
int jj_fq_sqrt(jj_ctx*, size_t n, const void* a, void* out, uint8_t* ok);
int jj_fr_add(jj_ctx*, size_t n, const void* a, const void* b, void* out);
int jj_fr_sub(jj_ctx*, size_t n, const void* a, const void* b, void* out);
int jj_fr_mul(jj_ctx*, size_t n, const void* a, const void* b, void* out);
int jj_fr_neg(jj_ctx*, size_t n, const void* a, void* out);
int jj_fr_square(jj_ctx*, size_t n, const void* a, void* out);
int jj_fr_double(jj_ctx*, size_t n, const void* a, void* out);
int jj_fr_invert(jj_ctx*, size_t n, const void* a, void* out, uint8_t* ok);
int jj_fr_sqrt(jj_ctx*, size_t n, const void* a, void* out, uint8_t* ok);
/* pow (src/fr.rs:403-414): out[i] = a[i] ^ exp[i], exponent = 32-byte little-endian integer, constant-time ladder */
int jj_fq_pow(jj_ctx*, size_t n, const void* a, const void* exp32, void* out);
int jj_fr_pow(jj_ctx*, size_t n, const void* a, const void* exp32, void* out);
/* from_bytes: ok=0 (and out=0) when the integer is >= p (src/fr.rs:268-292).  from_bytes_wide: 64-byte input
 * reduced mod p (src/fr.rs:312-343). */
int jj_fq_from_bytes(jj_ctx*, size_t n, const void* in32, void* out, uint8_t* ok);
int jj_fr_from_bytes(jj_ctx*, size_t n, const void* in32, void* out, uint8_t* ok);
int jj_fq_from_bytes_wide(jj_ctx*, size_t n, const void* in64, void* out);
int jj_fr_from_bytes_wide(jj_ctx*, size_t n, const void* in64, void* out);
/* PrimeFieldBits::to_le_bits (src/fr.rs:746-773): the canonical integer as 256 bytes of 0/1, bit 0 first;
 * char_le_bits (src/fr.rs:775-785): the modulus r in the same layout (host only, no context). */
int jj_fq_to_le_bits(jj_ctx*, size_t n, const void* a, void* out256);
int jj_fr_to_le_bits(jj_ctx*, size_t n, const void* a, void* out256);
int jj_fr_char_le_bits(uint8_t out256[256]);

/* ---- elementwise point operations (affine in, affine out; extended coordinates inside) ----------------- */
/* double src/lib.rs:739-828; add/sub = Ext +/- Affine src/lib.rs:1012-1028; neg 92-104; mul_by_cofactor 722-724 */
int jj_point_double(jj_ctx*, size_t n, const void* p, void* out);
int jj_point_add(jj_ctx*, size_t n, const void* p, const void* q, void* out);
int jj_point_sub(jj_ctx*, size_t n, const void* p, const void* q, void* out);
int jj_point_neg(jj_ctx*, size_t n, const void* p, void* out);
int jj_point_mul_by_cofactor(jj_ctx*, size_t n, const void* p, void* out);
/* AffinePoint::to_niels src/lib.rs:652-658: out = 96 bytes (v+u, v-u, 2d*u*v), each canonical LE */
int jj_point_to_niels(jj_ctx*, size_t n, const void* p, void* out96);
/* predicates -> uint8_t; src/lib.rs:691-719 and (is_on_curve) 670-675 */
int jj_is_identity(jj_ctx*, size_t n, const void* p, uint8_t* out);
int jj_is_small_order(jj_ctx*, size_t n, const void* p, uint8_t* out);
/* is_torsion_free: same predicate as [r]P == O (lib.rs:709-711) for points on the curve, computed with the order-8
 * Tate pairing (one exponentiation) instead of the 252-step ladder; option torsion_check_ladder = 1 selects the ladder. */
int jj_is_torsion_free(jj_ctx*, size_t n, const void* p, uint8_t* out);
int jj_is_prime_order(jj_ctx*, size_t n, const void* p, uint8_t* out);
int jj_is_on_curve(jj_ctx*, size_t n, const void* p, uint8_t* out);
/* Sum for ExtendedPoint (src/lib.rs:183-193): out = one affine point = sum of n affine points (identity if n=0) */
int jj_point_sum(jj_ctx*, size_t n, const void* p, void* out64);

/* ---- scalar multiplication ----------------------------------------------------------------------------- */
/* out[i] = to_affine(points[i] * scalars[i])   (`ExtendedPoint * Fr`, src/lib.rs:873-879 -> 831-833 -> 357-379).
 * scalars are raw 32-byte patterns; only the low 252 bits are used, as in the reference ladder.  Results are specified for on-curve points.
 * CONSTANT-TIME like the reference's ladder (conditional_select, src/lib.rs:334-343): neither the instruction stream nor any memory
 * address depends on the scalar.  Signed 3-bit windows (k' = k + sum 4 * 8^i: 84 windows tile the 252 bits, bit 252 is the recoding carry),
 * table {P, 2P, 3P, 4P}: {P, 2P} in registers, {3P, 4P} in a per-lane LDS slot that is read whole for every window; the entry is picked
 * with bit masks, the sign applied through the subtraction formulas: 85 additions + 252 doublings.  Batches up to vb_quad_max
 * (32 768) units run one scalar multiplication per quad of lanes (every lane keeps its own coordinate of the four entries in
 * registers): same discipline, a third of the latency. */
int jj_varbase_mul(jj_ctx*, size_t n, const void* scalars32, const void* points64, void* out64);
/* same, result written as 32-byte compressed encodings (to_bytes of the product, src/lib.rs:455-464, 1419-1421) */
int jj_varbase_mul_compressed(jj_ctx*, size_t n, const void* scalars32, const void* points64, void* out32);
/* the name rounds 3-4 gave the constant-time ladder when it was the opt-in: the same as jj_varbase_mul */
int jj_varbase_mul_ct(jj_ctx*, size_t n, const void* scalars32, const void* points64, void* out64);
/* VARIABLE-TIME variants for PUBLIC scalars (what rounds 1-4 shipped as jj_varbase_mul): signed 5-bit windows, the lane's table
 * {0 .. 16} P in device memory, read at a digit-dependent address: a scalar-independent instruction stream but scalar-dependent
 * memory addresses (cache timing).  1.6-4.5 % faster than jj_varbase_mul at 2^20 units depending on the box (ratios 0.984 and 0.955: profiles/r5_vb_ct_window.txt, r6_vb_ct_window.txt).
 * Nothing makes jj_varbase_mul / _compressed take this ladder: no option, no environment variable. */
int jj_varbase_mul_vartime(jj_ctx*, size_t n, const void* scalars32, const void* points64, void* out64);
int jj_varbase_mul_vartime_compressed(jj_ctx*, size_t n, const void* scalars32, const void* points64, void* out32);
/* One scalar, many bases: out[i] = points[i] * scalar (the `Wnaf::scalar(..).base(..)` reuse pattern of the group crate,
 * cf. WnafGroup src/lib.rs:1318-1336; the crate's Wnaf machinery is variable-time by design).  The kernel of jj_varbase_mul_vartime with the
 * one scalar read through a wave-uniform address. */
int jj_varbase_mul_scalar(jj_ctx*, size_t n, const void* scalar32, const void* points64, void* out64);
/* Same group element, but computed with the reference's exact 252-step double-and-add-always ladder and
 * returned in projective form: out = 160 bytes (U,V,Z,T1,T2 canonical LE) matching the Rust ExtendedPoint
 * fields bit for bit.  For parity testing, not for throughput. */
int jj_varbase_mul_exact(jj_ctx*, size_t n, const void* scalars32, const void* points64, void* out160);

/* Fixed-base: `AffineNielsPoint * Fr` / multiply_bits (src/lib.rs:272-310) for one base point.
 * The table holds multiples of the base as affine-Niels triples.
 *   window_bits 0 or 7 : signed comb, 8 teeth x 8 column blocks, 140 KiB table staged in LDS: 32 mixed additions + 3 doublings
 *                        per scalar; the entry (one of 128) is selected with two ds_bpermute shuffles and a mask (constant-time:
 *                        no secret-dependent address);
 *   window_bits 6      : signed 6-bit windows, 152 KiB table staged in LDS, one ds_bpermute shuffle per entry (constant-time) —
 *                        43 mixed additions per scalar;
 *   window_bits 8..16  : wider windows (0.6 MB .. 64 MB table, one 128-byte line per entry) kept in L2 / Infinity Cache and gathered per lane
 *                        (variable-time addressing) — ceil(253/w) additions per scalar.
 * A table lives in the memory of the device of the context that built it and serves every context of that device; a context of
 * another device gets JJ_ERR_INVALID (jj_multi_fixedbase_table_create replicates a table per device). */
int jj_fixedbase_table_create(jj_ctx*, const void* base64, int window_bits /* 0 = default */, jj_table** out);
int jj_fixedbase_table_destroy(jj_ctx*, jj_table* t);
int jj_fixedbase_mul(jj_ctx*, const jj_table* t, size_t n, const void* scalars32, void* out64);
int jj_fixedbase_mul_compressed(jj_ctx*, const jj_table* t, size_t n, const void* scalars32, void* out32);
/* Sums over several fixed bases (the primitive is AffineNielsPoint::multiply_bits, src/lib.rs:297-301):
 *   out[i] = sum_{j < nbases} tables[j] * scalars32[j * n + i]      (base-major scalar array, nbases * n * 32 bytes)
 * e.g. value commitments v*G_v + r*G_r or windowed Pedersen sums.  One pass per base; the accumulator stays in
 * extended coordinates between the passes, so there is a single normalisation. */
int jj_fixedbase_multi_mul(jj_ctx*, const jj_table* const* tables, int nbases, size_t n, const void* scalars32, void* out64);

/* The same sums when the scalars are SHORT, in one pass over one shared LDS table set (Pedersen-style windowed sums, value
 * commitments v*G_v with 64-bit v, ...; the primitive is AffineNielsPoint::multiply_bits, src/lib.rs:297-301, which takes a bit
 * slice):   out[i] = sum_{b < nbases} bases[b] * (scalars32[b * n + i] mod 2^scalar_bits[b])
 * The 42 six-bit window slots of the LDS-staged table are divided among the bases (base b takes ceil((scalar_bits[b] + 2) / 6)
 * of them; the sum must not exceed 42, e.g. 3 bases x 64 bits, 2 x 124, 6 x 40), one accumulator per lane walks all of them:
 * 43 additions per unit whatever nbases is (jj_fixedbase_multi_mul: 32-43 per base), constant-time shuffle select.  The table
 * is destroyed with jj_fixedbase_table_destroy. */
int jj_fixedbase_composite_create(jj_ctx*, int nbases, const void* bases64, const int* scalar_bits, jj_table** out);
int jj_fixedbase_composite_mul(jj_ctx*, const jj_table* t, size_t n, const void* scalars32, void* out64);

/* Multi-scalar multiplication: out = to_affine(sum_i points[i] * scalars[i])  (semantics: iterator Sum of
 * `p * k`, src/lib.rs:183-193 + 873-879; the reference has no MSM algorithm).  n = 0 gives the identity.
 * The device reduces the terms to a record of partial window sums (two launches for small batches, Pippenger above); the last
 * step -- adding the partial sums of each window, Horner over the windows (a chain of 252 dependent doublings) and one
 * inversion -- runs on the calling host thread, so for every n this call waits for the stream even when all pointers are
 * device pointers (the 64-byte result is then copied to out64 asynchronously).  HOST arrays of 2^19 terms and more are summed in
 * two to eight passes, the copy of each pass's slice beside the kernels of the pass before (option msm_host_split = 0: one pass). */
int jj_msm(jj_ctx*, size_t n, const void* scalars32, const void* points64, void* out64);
/* The same in two halves, so that the host tail of one MSM overlaps the kernels of the next: jj_msm_begin queues all device
 * work of one MSM plus the copy of its records into a page-locked buffer owned by the job and returns at once (device
 * pointers; host arrays are staged first); jj_msm_finish waits for THAT job only, runs the host tail and writes the 64-byte
 * result (host pointer: complete on return; device pointer: copy queued on the context's stream).  A context owns several MSM
 * lanes (streams of their own + workspaces; option msm_lanes, default 3 (two until round 6; with three or more jobs in flight the third lane returns 10 % at 2^17 terms); 1 = every job on the context's stream): jobs with
 * device-pointer inputs alternate over them, so that the latency-bound end of one MSM (fix-up, bucket reduce: a few hundred
 * wavefronts) overlaps the sort and accumulation of the next; every job starts after the work already queued on the
 * context's stream when it was begun.  Jobs may be finished in any order, each
 * exactly once (finish releases the job, also on error).  Device input arrays must stay valid until the job is finished; every job must be finished
 * before its context is destroyed. */
typedef struct jj_msm_job jj_msm_job;
int jj_msm_begin(jj_ctx*, size_t n, const void* scalars32, const void* points64, jj_msm_job** job);
int jj_msm_finish(jj_msm_job* job, void* out64);
/* Opt-in device-side finish: the same sum with NO host hop.  The record of window sums stays on the device, one quad of lanes runs
 * the Horner chain (252 dependent doublings) and the inversion there and writes the affine point to out64_dev (DEVICE memory, 16-byte
 * aligned); the call only queues work on the context's stream and returns.  The finish is a ~0.5 ms chain on four lanes against ~0.05 ms
 * on a host core (profiles/r4_msm_dev_finish.txt): use it when the sum feeds the next kernel and the host thread must not wait on the
 * stream (jj_msm does: D2H of the record + host tail + H2D of the point); use jj_msm / jj_msm_begin for latency and throughput.
 * At most 2^24 terms per call; inputs may be host arrays (staged) or device arrays. */
int jj_msm_dev(jj_ctx*, size_t n, const void* scalars32, const void* points64, void* out64_dev);
/* MSM cut across devices or ranks.  jj_msm_partial leaves the RECORD of partial window sums instead of the
 * point: JJ_MSM_PARTIAL_BYTES bytes (64-byte header: magic, version, number of windows W, 1, bit mask of the windows
 * present, n; then one 128-byte point per window: U, V, Z and T = T1 T2, each the 256-bit little-endian integer of
 * value x 2^256 mod q, the Montgomery form of the host tail; unused space zeroed), written to device memory (asynchronous: ready for an all_gather over RCCL) or host memory.  At most 2^24 terms per call.
 *   part_index = 0, part_count = 1   all windows of the n terms given       (term partition: each rank passes ITS terms)
 *   part_index = g, part_count = G   windows g, g + G, ... of the n terms    (window partition: each rank passes ALL terms)
 * jj_msm_combine (host only, no context) adds any number of records -- window by window where their layouts agree, so the
 * Horner chain runs once per layout -- and returns the affine sum: one copy to the host, one host tail and one inversion
 * for the whole distributed MSM.  Records of a window partition must come from calls with the same n. */
#define JJ_MSM_PARTIAL_BYTES 8256u   /* 64 + 64 windows x 128 */
int jj_msm_partial(jj_ctx*, size_t n, const void* scalars32, const void* points64, int part_index, int part_count, void* record);
int jj_msm_combine(size_t count, const void* records_host, void* out64_host);
/* The same sum (Sum for ExtendedPoint, src/lib.rs:183-193, over the parts) for records that are in DEVICE memory -- what an all_gather
 * delivered, JJ_MSM_PARTIAL_BYTES apart, 16-byte aligned: the records are added window by window on the device into ONE record
 * (a quad of lanes per record and window, an LDS tree over the records), which takes one 8 KB copy and the single-record host tail
 * whatever `count` is.  Records of different window layouts fall back to one copy of all of them and the host's additions.
 * out64 may be a host or a device pointer. */
int jj_msm_combine_dev(jj_ctx*, size_t count, const void* records_dev, void* out64);

/* The same exchange behind ONE call, for callers that run one process per GPU (the context holds "streams, tables,
 * RCCL comm"; 8(e)).  jj_ctx_set_comm lends the context an RCCL communicator that the CALLER created (ncclCommInitRank; the
 * rendezvous that carries the ncclUniqueId between the processes belongs to the application -- examples/msm_rccl.cpp does it with a
 * file, bench.py over torch.distributed) together with this process's rank.  all_gather_fn: address of the ncclAllGather of the
 * RCCL library that made the communicator, or NULL (then looked up among the symbols of the process, then in librccl.so.1);
 * libjubjub_hip.so itself does not link RCCL.  nccl_comm = NULL detaches.  The communicator must outlive its use here; the caller
 * destroys it.
 * jj_msm_allgather: every rank calls it with ITS terms (partition 0) or with ALL terms (partition 1: rank g reduces windows g, g + G,
 * ... of the whole batch; same n on every rank): record of window sums (jj_msm_partial) -> ncclAllGather of JJ_MSM_PARTIAL_BYTES per
 * rank on the context's stream -> the G records folded into one on the device (jj_msm_combine_dev) -> one 8 KB copy -> one host tail.  Every rank returns the
 * same point; out64 may be a host or a device pointer.  A collective: all ranks must call it, in the same order -- a rank whose call
 * fails BEFORE the gather (bad arguments, out of memory) never enters it and the other ranks wait for it: treat an error of
 * jj_msm_allgather as fatal for the communicator (as with any RCCL collective). */
int jj_ctx_set_comm(jj_ctx*, void* nccl_comm, int rank, int nranks, void* all_gather_fn);
int jj_msm_allgather(jj_ctx*, size_t n, const void* scalars32, const void* points64, int partition, void* out64);
/* jj_msm_allgather in two halves, for a caller with many MSMs to sum (a prover's commitments): jj_msm_allgather_begin queues the
 * rank's window sums, the ncclAllGather and the fold of the gathered records on one of the context's MSM lanes (all stream-ordered;
 * the folded record lands in a page-locked buffer the job owns) and returns; jj_msm_finish (above) waits for THAT job and runs the
 * single-record host tail.  With two to four jobs in flight a rank's gather, fold, wait and host tail run beside the kernels of its
 * next MSM, so the sustained rate of the distributed sum is set by the kernels of one share (2^17 terms of a 2^20-term MSM cut
 * eight ways: 0.245-0.255 ms per MSM against 0.385 ms + the gather's latency for one synchronous call).  A collective like jj_msm_allgather:
 * every rank begins the same jobs in the same order (the gathers of one communicator run in the order they were queued); jobs
 * may be finished in any order, each exactly once; at most 2^24 terms per rank and job; device input arrays stay valid until the
 * job is finished. */
int jj_msm_allgather_begin(jj_ctx*, size_t n, const void* scalars32, const void* points64, int partition, jj_msm_job** job);

/* ---- encodings ----------------------------------------------------------------------------------------- */
#define JJ_DECOMPRESS_ZIP216          1u  /* reject the two non-canonical encodings (src/lib.rs:469-471, 522-531) */
#define JJ_DECOMPRESS_TORSION_FREE    2u  /* additionally require [r]P = O  (SubgroupPoint::from_bytes, lib.rs:1427-1429) */
#define JJ_DECOMPRESS_NOT_SMALL_ORDER 4u  /* additionally reject small-order points (lib.rs:699-705) */
#define JJ_DECOMPRESS_CLEAR_COFACTOR  8u  /* return [8]P (lib.rs:722-724, 1343-1345) */
/* AffinePoint::from_bytes / from_bytes_pre_zip216_compatibility / batch_from_bytes (src/lib.rs:469-627) */
int jj_decompress(jj_ctx*, size_t n, const void* in32, unsigned flags, void* out64, uint8_t* ok);
/* AffinePoint::to_bytes src/lib.rs:455-464 */
int jj_compress(jj_ctx*, size_t n, const void* points64, void* out32);
/* batch_normalize src/lib.rs:1084-1107: n x 160 bytes (U,V,Z,T1,T2 canonical LE) -> n x 64 bytes affine */
int jj_batch_normalize(jj_ctx*, size_t n, const void* ext160, void* out64);


/* ---- synthetic inputs (Group::random semantics; counter-based, reproducible on any device or on the host) ---- */
/* scalars[i] = four splitmix64 words of the stream seed + (first_index + i) * 4 + j, top 4 bits cleared, minus r if >= r:
 * canonical Fr elements (the input side of Field::random, src/fr.rs:684-688, with a counter-based generator). */
int jj_synth_scalars(jj_ctx*, size_t n, uint64_t seed, uint64_t first_index, void* out32);
/* the same four words per unit as they come (arbitrary 256-bit patterns: values >= q, sign-bit noise for the decoder) */
int jj_synth_bytes32(jj_ctx*, size_t n, uint64_t seed, uint64_t first_index, void* out32);
/* ExtendedPoint::random (src/lib.rs:1244-1267): v = Fq::random (64 PRNG bytes through from_bytes_wide), flip = next_u32 % 2,
 * u = sqrt((v^2 - 1) / (1 + d v^2)) or draw again, (flip ? -u : u, v), draw again if it is the identity.  subgroup != 0:
 * SubgroupPoint::random (src/lib.rs:1290-1298), i.e. [8]P, drawn again if that is the identity.  Unit i reads the
 * splitmix64 stream seed + ((first_index + i) << 16) + 16 * attempt + {0..7: v, 8: flip}.  attempts (optional): draws used. */
int jj_random_points(jj_ctx*, size_t n, uint64_t seed, uint64_t first_index, int subgroup, void* out64, uint32_t* attempts);


/* ---- several devices of one node ---------------------------------------------------------- */
/* One context per listed device, one host thread + stream per device, contiguous shards [g*n/G, (g+1)*n/G); no data-path
 * collective for the independent-batch workloads (north_star: "shard embarrassingly across the 8 GPUs of one node").
 * jj_multi_msm: every device reduces its own terms to a record of partial window sums (jj_msm_partial); the records of all
 * devices meet in one host tail on the calling thread (jj_msm_combine).  Array arguments are HOST pointers (a device pointer is JJ_ERR_INVALID).  A device may be listed
 * more than once.  Processes that keep their batches resident in HBM run one process per GPU instead and exchange the MSM
 * records with an RCCL all_gather: jj_ctx_set_comm + jj_msm_allgather above (C / C++ / Rust: examples/msm_rccl.cpp), or
 * jubjub_amd/dist.py over torch.distributed.  A jj_ctx may be used from several host threads: its entry
 * points serialise on a per-context lock. */
typedef struct jj_multi jj_multi;
typedef struct jj_mtable jj_mtable;
int jj_multi_create(const int* devices, int ndev, jj_multi** out);
int jj_multi_destroy(jj_multi* m);
int jj_multi_device_count(jj_multi* m);
jj_ctx* jj_multi_ctx(jj_multi* m, int index);            /* the per-device context, for the single-device entry points */
const char* jj_multi_last_error(jj_multi* m);
int jj_multi_varbase_mul(jj_multi* m, size_t n, const void* scalars32, const void* points64, void* out64);
int jj_multi_fixedbase_table_create(jj_multi* m, const void* base64, int window_bits, jj_mtable** out);   /* replicated per device */
int jj_multi_fixedbase_table_destroy(jj_multi* m, jj_mtable* t);
int jj_multi_fixedbase_mul(jj_multi* m, const jj_mtable* t, size_t n, const void* scalars32, void* out64);
int jj_multi_decompress(jj_multi* m, size_t n, const void* in32, unsigned flags, void* out64, uint8_t* ok);
int jj_multi_msm(jj_multi* m, size_t n, const void* scalars32, const void* points64, void* out64);
/* Last step of an MSM cut across devices or processes (the reference's `Sum`, src/lib.rs:183-193, over the partial sums): adds
 * `count` partial points (canonical affine, 64 bytes each) -> one affine point.  HOST pointers, no context: a short chain of
 * dependent additions and one inversion, run on the calling thread with the arithmetic of the MSM's own host tail.
 * jj_multi_msm ends with it; processes that all_gather their partial points (one process per GPU) call it on the gathered bytes. */
int jj_msm_fold_partials(size_t count, const void* parts64_host, void* out64_host);

#ifdef __cplusplus
}
#endif
#endif /* JUBJUB_HIP_H */
