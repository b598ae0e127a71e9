// HIP kernels of the Jubjub engine (gfx950).  One field element / one curve point per lane, limbs in VGPRs.
// Included through jj_engine.h by the library's translation units.  The device helpers and argument types are always visible; the
// __global__ kernels are compiled into the ONE translation unit that launches them: JJ_KERNELS_BATCH (jj_abi.hip: fields, points,
// ladders, fixed-base, codec, generators), JJ_KERNELS_MSM (jj_msm.hip: jj_msm_kernels.h), JJ_KERNELS_PROBE (jj_pipeline.hip: k_peak_mad).
#pragma once
#include "jj_curve.h"

namespace jj {

// ------------------------------------------------------------------------------------------------ I/O helpers
// 32-byte wire elements are read/written as two 16-byte vectors per lane: 64 lanes x 32 B = 2 KiB contiguous per
// wave instruction pair (coalesced AoS).  Pointers must be 16-byte aligned.
static JJ_DEV void load8(u32 (&w)[8], const void* base, size_t idx) {
  const uint4* p = reinterpret_cast<const uint4*>(static_cast<const uint8_t*>(base) + idx * 32);
  const uint4 a = p[0], b = p[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
static JJ_DEV void store8(void* base, size_t idx, const u32 (&w)[8]) {
  uint4* p = reinterpret_cast<uint4*>(static_cast<uint8_t*>(base) + idx * 32);
  p[0] = make_uint4(w[0], w[1], w[2], w[3]);
  p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
static JJ_DEV void zero8(u32 (&w)[8]) { _Pragma("unroll") for (int i = 0; i < 8; i++) w[i] = 0; }

// Internal device format for points between kernels: structure-of-arrays, limb-major: X[limb][i] (u32), so every
// limb load/store of a wave is one contiguous 256-byte segment.
struct SoA {
  u32* base;      // [ncoord][NL][n]
  size_t n;
  JJ_DEV void put(int coord, size_t i, const Fe& a) const {
    _Pragma("unroll") for (int l = 0; l < NL; l++) base[((size_t)coord * NL + l) * n + i] = a.l[l];
  }
  JJ_DEV Fe get(int coord, size_t i) const {
    Fe a; _Pragma("unroll") for (int l = 0; l < NL; l++) a.l[l] = base[((size_t)coord * NL + l) * n + i]; return a;
  }
};

static JJ_DEV Affine load_affine(const void* pts, size_t i) {
  u32 wu[8], wv[8];
  load8(wu, pts, 2 * i); load8(wv, pts, 2 * i + 1);
  Affine a; a.u = Fq::from_words(wu); a.v = Fq::from_words(wv);   // from_raw semantics (reduces mod q)
  return a;
}
static JJ_DEV void store_affine(void* out, size_t i, const Fe& u, const Fe& v) {
  u32 w[8];
  Fq::to_words(w, u); store8(out, 2 * i, w);
  Fq::to_words(w, v); store8(out, 2 * i + 1, w);
}

// ------------------------------------------------------------------------------------------------ K1: field ops
enum FieldOp { OP_ADD = 0, OP_SUB, OP_MUL, OP_NEG, OP_SQUARE, OP_DOUBLE, OP_INVERT, OP_SQRT, OP_FROM_BYTES, OP_FROM_WIDE };

// Fr::sqrt: a^((r+1)/4), Some iff it squares back (reference src/fr.rs:384-399)
static JJ_DEV Fe fr_sqrt(const Fe& a, bool& ok) {
  const Fe s = Fr::pow_const<8, FrP::SQRT_EXP>(a);
  ok = Fr::eq(Fr::sqr(s), a);
  return s;
}
// ---- Fq square root: the value ff::helpers::sqrt_tonelli_shanks returns (bls12_381 0.8.0 Scalar::sqrt, S = 32,
// ROOT_OF_UNITY = 7^t; call sites reference src/lib.rs:515,610,1253), computed by a 4 x 8-bit Pohlig-Hellman discrete
// log in the 2^32-torsion instead of Tonelli-Shanks' ~500 data-dependent squarings.  With g = ROOT_OF_UNITY, b = a^t = g^e; the digits e_i come from a 64 KiB direct table
// keyed by 16 bits (limb SQRT_KEY_LIMB) of the canonical integer b_i^(2^(24-8i)); Tonelli-Shanks' answer is x = a^((t+1)/2) * g^s with
// s = ((2^32 - e) mod 2^32) / 2, i.e. x for e = 0 and -(x * g^(-e/2)) otherwise.
// the 256 values (g^(2^24))^k are told apart by 16 bits of their canonical integer: bits [0, 16) of limb 3 (bits 87..102) are
// collision-free (limb 0 is not: 254 distinct keys); tests/test_bounds.py::test_sqrt_dlog_key_is_collision_free
constexpr int SQRT_KEY_LIMB = 3;
struct SqrtTables {
  const uint8_t* dlog;   // [65536]  key16 -> k   with key16 = 16 bits (limb SQRT_KEY_LIMB) of the canonical integer (g^(2^24))^k
  const u32* npow;       // [4][256][NL]  g^(-k * 2^(8i))
};
static JJ_DEV Fe sqrt_tab(const SqrtTables& T, int i, u32 k) {
  const u32* p = T.npow + ((size_t)i * 256 + k) * NL;
  Fe r; _Pragma("unroll") for (int l = 0; l < NL; l++) r.l[l] = p[l]; return r;
}
static JJ_DEV Fe fq_sqrt_fast(const Fe& a, bool& ok, const SqrtTables& T) {
  const Fe w = Fq::pow_const<8, FqP::TM1D2>(a);
  const Fe x = Fq::mul(a, w);
  const Fe b = Fq::mul(x, w);                          // a^t = g^e
  // digits of e, least significant first: b^(2^24) = gamma^e0; then strip the known digits from b^(2^16), b^(2^8), b
  // with table entries g^(-k 2^(8i)) instead of squaring a reduced b again (24 squarings in all)
  Fe c1 = b;
  #pragma unroll 1
  for (int s = 0; s < 8; s++) c1 = Fq::sqr(c1);
  Fe c2 = c1;
  #pragma unroll 1
  for (int s = 0; s < 8; s++) c2 = Fq::sqr(c2);
  Fe c3 = c2;
  #pragma unroll 1
  for (int s = 0; s < 8; s++) c3 = Fq::sqr(c3);
  const u32 e0 = T.dlog[Fq::to_plain(c3).l[SQRT_KEY_LIMB] & 0xffffu];
  const u32 e1 = T.dlog[Fq::to_plain(Fq::mul(c2, sqrt_tab(T, 2, e0))).l[SQRT_KEY_LIMB] & 0xffffu];
  const u32 e2 = T.dlog[Fq::to_plain(Fq::mul(Fq::mul(c1, sqrt_tab(T, 1, e0)), sqrt_tab(T, 2, e1))).l[SQRT_KEY_LIMB] & 0xffffu];
  const u32 e3 = T.dlog[Fq::to_plain(Fq::mul(Fq::mul(b, sqrt_tab(T, 0, e0)), Fq::mul(sqrt_tab(T, 1, e1), sqrt_tab(T, 2, e2)))).l[SQRT_KEY_LIMB] & 0xffffu];
  const u32 e = e0 | (e1 << 8) | (e2 << 16) | (e3 << 24);
  const u32 h = e >> 1;
  Fe z = Fq::mul(sqrt_tab(T, 0, h & 255u), sqrt_tab(T, 1, (h >> 8) & 255u));
  z = Fq::mul(z, Fq::mul(sqrt_tab(T, 2, (h >> 16) & 255u), sqrt_tab(T, 3, (h >> 24) & 255u)));
  const Fe xr = Fq::select(Fq::neg(Fq::mul(x, z)), x, e == 0 ? ~0u : 0u);
  ok = Fq::eq(Fq::sqr(xr), a);
  return xr;
}
// builds the tables once per context: thread k (< 256): dlog entry of gamma^k; thread 256 + (i*256 + k): g^(-k 2^(8i))
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_sqrt_tables_init(uint8_t* dlog, u32* npow) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid < 256) {
    Fe gam = Fq::konst(FqP::ROOT_POW2[24]);              // g^(2^24), order 256
    Fe p = Fq::one();
    for (int j = 0; j < tid; j++) p = Fq::mul(p, gam);
    dlog[Fq::to_plain(p).l[SQRT_KEY_LIMB] & 0xffffu] = (uint8_t)tid;
  } else if (tid < 256 + 1024) {
    const int i = (tid - 256) >> 8, k = (tid - 256) & 255;
    Fe base = Fq::konst(FqP::ROOT_OF_UNITY_INV);
    for (int j = 0; j < 8 * i; j++) base = Fq::sqr(base);  // g^(-2^(8i))
    Fe p = Fq::one();
    for (int j = 0; j < k; j++) p = Fq::mul(p, base);
    const Fe c = Fq::canon(p);
    _Pragma("unroll") for (int l = 0; l < NL; l++) npow[((size_t)i * 256 + k) * NL + l] = c.l[l];
  }
}
#endif  // JJ_KERNELS_BATCH

#ifdef JJ_KERNELS_BATCH
template <class P, int OP>
__global__ void __launch_bounds__(256) k_field_op(size_t n, const void* a, const void* b, void* out, uint8_t* okp, SqrtTables tabs) {
  typedef Field<P> F;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 wa[8], wb[8], wo[8];
  bool ok = true;
  Fe r;
  if constexpr (OP == OP_FROM_WIDE) {
    load8(wa, a, 2 * i); load8(wb, a, 2 * i + 1);
    r = F::from_words_wide(wa, wb);
  } else if constexpr (OP == OP_FROM_BYTES) {
    load8(wa, a, i);
    r = F::from_words_checked(wa, ok);
  } else {
    load8(wa, a, i);
    const Fe x = F::from_words(wa);
    if constexpr (OP == OP_ADD || OP == OP_SUB || OP == OP_MUL) {
      load8(wb, b, i);
      const Fe y = F::from_words(wb);
      if constexpr (OP == OP_ADD) r = F::add(x, y);
      if constexpr (OP == OP_SUB) r = F::sub(x, y);
      if constexpr (OP == OP_MUL) r = F::mul(x, y);
    }
    if constexpr (OP == OP_NEG) r = F::neg(x);
    if constexpr (OP == OP_SQUARE) r = F::sqr(x);
    if constexpr (OP == OP_DOUBLE) r = F::dbl(x);
    if constexpr (OP == OP_INVERT) { ok = !F::is_zero(x); r = F::invert(x); }
    if constexpr (OP == OP_SQRT) {
      if constexpr (P::PBITS == 255) r = fq_sqrt_fast(x, ok, tabs); else r = fr_sqrt(x, ok);
    }
  }
  F::to_words(wo, r);
  if (!ok) zero8(wo);
  store8(out, i, wo);
  if (okp) okp[i] = ok ? 1 : 0;
}
#endif  // JJ_KERNELS_BATCH

// Fr::pow / Fq::pow (reference src/fr.rs:403-414): constant-time square-and-multiply over all 256 exponent bits,
// one (base, exponent) pair per lane; the exponent is a little-endian 256-bit integer.
#ifdef JJ_KERNELS_BATCH
template <class P>
__global__ void __launch_bounds__(256) k_field_pow(size_t n, const void* a, const void* e, void* out) {
  typedef Field<P> F;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 wa[8], we[8], wo[8];
  load8(wa, a, i); load8(we, e, i);
  const Fe x = F::from_words(wa);
  Fe res = F::one();
  #pragma unroll 1
  for (int bit = 255; bit >= 0; bit--) {
    u32 word = we[0];
    _Pragma("unroll") for (int w = 1; w < 8; w++) word = ((bit >> 5) == w) ? we[w] : word;
    res = F::sqr(res);
    const Fe t = F::mul(res, x);
    res = F::select(res, t, 0u - ((word >> (bit & 31)) & 1u));
  }
  F::to_words(wo, res);
  store8(out, i, wo);
}
#endif  // JJ_KERNELS_BATCH

// PrimeFieldBits::to_le_bits (reference src/fr.rs:746-785): the canonical integer, one byte (0/1) per bit, little-endian
#ifdef JJ_KERNELS_BATCH
template <class P>
__global__ void __launch_bounds__(256) k_field_to_bits(size_t n, const void* a, void* out256) {
  typedef Field<P> F;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 wa[8], w[8];
  load8(wa, a, i);
  F::to_words(w, F::from_words(wa));
  uint4* o = reinterpret_cast<uint4*>(static_cast<uint8_t*>(out256) + i * 256);
  _Pragma("unroll") for (int v = 0; v < 16; v++) {            // 16 bits -> four words of four 0/1 bytes
    const u32 h = (w[v >> 1] >> (16 * (v & 1))) & 0xffffu;
    u32 q[4];
    _Pragma("unroll") for (int k = 0; k < 4; k++) { const u32 nib = (h >> (4 * k)) & 15u; q[k] = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21); }
    o[v] = make_uint4(q[0], q[1], q[2], q[3]);
  }
}
#endif  // JJ_KERNELS_BATCH

// ------------------------------------------------------------------------------------------------ K2: point ops
enum PointOp { PT_DOUBLE = 0, PT_ADD, PT_SUB, PT_NEG, PT_COFACTOR, PT_TO_NIELS,
               PT_IS_IDENTITY, PT_IS_SMALL_ORDER, PT_IS_ON_CURVE };

// Elementwise point ops.  Results leave as extended (U,V,Z) in the SoA buffer `ext` and are normalised by
// k_normalize (one shared inversion per chunk), except the byte-valued predicates / to_niels.
#ifdef JJ_KERNELS_BATCH
template <int OP>
__global__ void __launch_bounds__(256) k_point_op(size_t n, const void* p, const void* q, SoA ext, void* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Affine a = load_affine(p, i);
  Ext e = Curve::from_affine(a), r;
  if constexpr (OP == PT_DOUBLE) r = Curve::dbl(e);
  if constexpr (OP == PT_ADD) r = Curve::add<true>(e, Curve::to_niels(load_affine(q, i)));
  if constexpr (OP == PT_SUB) r = Curve::sub<true>(e, Curve::to_niels(load_affine(q, i)));
  if constexpr (OP == PT_NEG) r = Curve::neg(e);
  if constexpr (OP == PT_COFACTOR) r = Curve::mul_by_cofactor(e);
  if constexpr (OP <= PT_COFACTOR) { ext.put(0, i, r.u); ext.put(1, i, r.v); ext.put(2, i, r.z); }
  if constexpr (OP == PT_TO_NIELS) {
    const ANiels t = Curve::to_niels(a);
    u32 w[8];
    Fq::to_words(w, t.vpu); store8(out, 3 * i, w);
    Fq::to_words(w, t.vmu); store8(out, 3 * i + 1, w);
    Fq::to_words(w, t.t2d); store8(out, 3 * i + 2, w);
  }
  if constexpr (OP == PT_IS_IDENTITY) static_cast<uint8_t*>(out)[i] = Curve::is_identity(e);
  if constexpr (OP == PT_IS_SMALL_ORDER) static_cast<uint8_t*>(out)[i] = Curve::is_small_order(e);
  if constexpr (OP == PT_IS_ON_CURVE) static_cast<uint8_t*>(out)[i] = Curve::is_on_curve(a);
}
#endif  // JJ_KERNELS_BATCH

// decode follow-up for jj_decompress flags 4 (reject small order: U([4]P) == 0, reference src/lib.rs:699-705) and 8 (clear
// the cofactor, src/lib.rs:722-724) in one pass over the decoded points: the two tests share the first two doublings.
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_small_order_cofactor(size_t n, const void* pts, unsigned flags, SoA ext, uint8_t* ok) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ext e = Curve::dbl(Curve::dbl(Curve::from_affine(load_affine(pts, i))));
  if ((flags & 4u) && Fq::is_zero(e.u)) ok[i] = 0;
  if (flags & 8u) { e = Curve::dbl(e); ext.put(0, i, e.u); ext.put(1, i, e.v); ext.put(2, i, e.z); }
}
#endif  // JJ_KERNELS_BATCH

// ------------------------------------------------------------------------------------------------ K5: normalise
// batch_normalize (reference src/lib.rs:1084-1107, ff::BatchInverter): each lane owns CHUNK elements
// (element j of lane t is index t + j*T, so every access is coalesced), multiplies their Z's through, inverts
// once, and walks back.  Zero Z's are skipped like ff's BatchInverter (cannot occur for valid points).
// mode 0: write affine 64 B; any other mode: write compressed 32 B (AffinePoint::to_bytes, src/lib.rs:455-464).
#ifdef JJ_KERNELS_BATCH
template <int CHUNK>
__global__ void __launch_bounds__(256) k_normalize(size_t n, size_t T, SoA ext, SoA scratch, void* out, int mode) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  Fe acc = Fq::one();
  #pragma unroll 1
  for (int j = 0; j < CHUNK; j++) {
    const size_t i = t + (size_t)j * T;
    if (i >= n) break;
    scratch.put(0, i, acc);
    const Fe z = ext.get(2, i);                            // Z coordinates in the SoA are products (or the constant one): unique digits
    const u32 zz = Fq::is_zero_product(z) ? ~0u : 0u;
    acc = Fq::select(Fq::mul(acc, z), acc, zz);
  }
  // the running inverse is kept in PLAIN form (one product by the plain one per chunk): times a Montgomery-form prefix or Z it stays
  // plain, so 1/Z comes out plain and U/Z, V/Z go straight to plain integers (one product per coordinate, none for the conversion),
  // then two conditional additions of q
  Fe inv = Fq::mul(Fq::invert(acc), Fq::plain_one());
  #pragma unroll 1
  for (int j = CHUNK - 1; j >= 0; j--) {
    const size_t i = t + (size_t)j * T;
    if (i >= n) continue;
    const Fe z = ext.get(2, i);
    const u32 zz = Fq::is_zero_product(z) ? ~0u : 0u;
    const Fe zp = Fq::select(Fq::mul(inv, scratch.get(0, i)), Fq::zero(), zz);
    inv = Fq::select(Fq::mul(inv, z), inv, zz);
    u32 wu[8], wv[8];
    Fq::pack(wu, Fq::canon_plain_product(Fq::mul(ext.get(0, i), zp)));
    Fq::pack(wv, Fq::canon_plain_product(Fq::mul(ext.get(1, i), zp)));
    if (mode == 0) {
      store8(out, 2 * i, wu);
      store8(out, 2 * i + 1, wv);
    } else {
      wv[7] |= (wu[0] & 1u) << 31;
      store8(out, i, wv);
    }
  }
}
#endif  // JJ_KERNELS_BATCH

// extended SoA -> is_identity byte (reference src/lib.rs:691-696), optionally AND/ANDN into an existing ok byte
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_is_identity_ext(size_t n, SoA ext, uint8_t* out, int combine /*0 set,1 and,2 and-not*/) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool id = Fq::is_zero(ext.get(0, i)) && Fq::eq(ext.get(1, i), ext.get(2, i));
  if (combine == 0) out[i] = id;
  else if (combine == 1) out[i] = out[i] & (id ? 1 : 0);
  else out[i] = out[i] & (id ? 0 : 1);
}
#endif  // JJ_KERNELS_BATCH

// subgroup test by Tate pairing (Curve::is_torsion_free); combine: 0 set, 1 and
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_torsion_free(size_t n, const void* pts, uint8_t* out, int combine) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool tf = Curve::is_torsion_free(load_affine(pts, i));
  out[i] = combine ? (out[i] & (tf ? 1 : 0)) : (tf ? 1 : 0);
}
#endif  // JJ_KERNELS_BATCH

// user-facing batch_normalize input: 160-byte canonical (U,V,Z,T1,T2) -> SoA
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_ext160_to_soa(size_t n, const void* ext160, SoA ext) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[8];
  _Pragma("unroll") for (int c = 0; c < 3; c++) { load8(w, ext160, 5 * i + c); ext.put(c, i, Fq::from_words(w)); }
}
#endif  // JJ_KERNELS_BATCH

// ------------------------------------------------------------------------------------------------ K3: variable-base
// Signed fixed-window ladder (w = 5 by default), one scalar-mul per lane.  k (low 252 bits) is recoded as
// k = sum_i d_i 2^(w i) with signed w-bit digits d_i in [-2^(w-1), 2^(w-1)) (top digit unsigned), obtained from
// k' = k + sum 2^(w i + w - 1) as digit = window(k') - 2^(w-1).
// The lane's table {0..2^(w-1)}P (ExtendedNiels, 144 B each: 17 entries = 2448 B per lane) lives in a per-lane slot of a global
// workspace; the resident lanes hold ~320 MB of tables, far more than the 4 MB L2 of an XCD, so every entry read comes from
// HBM / Infinity Cache (14.7 KB of fabric traffic per scalar-mul, PMC-measured, hidden behind the multiply-adds: DESIGN.md 7).
// The entry for the next window is fetched before the w doublings that precede its use.
// Group element equals the reference ladder's (src/lib.rs:357-379, 831-833); negation is exact on the whole curve.
#ifndef JJ_VB_MINWAVES
#define JJ_VB_MINWAVES 2
#endif
#ifndef JJ_VB_W
#define JJ_VB_W 5
#endif
constexpr int VB_W = JJ_VB_W;                       // signed window width of the var-base ladder (4 or 5)
constexpr int VB_TABLE = 1 << (VB_W - 1);           // table entries {1 .. 2^(w-1)} P
constexpr int VB_NWIN = (253 + VB_W - 1) / VB_W;    // windows; the top one is unsigned (it holds the recoding carry)
constexpr int ENIELS_WORDS = 4 * NL;   // 36 words = 144 B

static JJ_DEV void store_eniels(u32* slot, const ENiels& n) {
  uint4* p = reinterpret_cast<uint4*>(slot);
  const Fe* c[4] = {&n.vpu, &n.vmu, &n.z2, &n.t2d};
  u32 w[ENIELS_WORDS];
  _Pragma("unroll") for (int k = 0; k < 4; k++) _Pragma("unroll") for (int l = 0; l < NL; l++) w[k * NL + l] = c[k]->l[l];
  _Pragma("unroll") for (int v = 0; v < ENIELS_WORDS / 4; v++) p[v] = make_uint4(w[4 * v], w[4 * v + 1], w[4 * v + 2], w[4 * v + 3]);
}
static JJ_DEV ENiels load_eniels(const u32* slot) {
  const uint4* p = reinterpret_cast<const uint4*>(slot);
  u32 w[ENIELS_WORDS];
  _Pragma("unroll") for (int v = 0; v < ENIELS_WORDS / 4; v++) { const uint4 x = p[v]; w[4 * v] = x.x; w[4 * v + 1] = x.y; w[4 * v + 2] = x.z; w[4 * v + 3] = x.w; }
  ENiels n;
  _Pragma("unroll") for (int l = 0; l < NL; l++) { n.vpu.l[l] = w[l]; n.vmu.l[l] = w[NL + l]; n.z2.l[l] = w[2 * NL + l]; n.t2d.l[l] = w[3 * NL + l]; }
  return n;
}
// recode: k' = (k & (2^252-1)) + sum_{i < NWIN-1} 2^(w i + w - 1); digit_i = window_i(k') - 2^(w-1), top window unsigned
static JJ_DEV void recode_signed(u32 (&k)[8]) {
  k[7] &= 0x0fffffffu;
  u64 c = 0;
  _Pragma("unroll") for (int i = 0; i < 8; i++) {
    u32 rc = 0;
    _Pragma("unroll") for (int j = 0; j < VB_NWIN - 1; j++) { const int bit = VB_W * j + VB_W - 1; if ((bit >> 5) == i) rc |= 1u << (bit & 31); }
    const u64 t = (u64)k[i] + rc + c;
    k[i] = (u32)t; c = t >> 32;
  }
}
// bits [w i, w i + w) of k'
static JJ_DEV u32 vb_window(const u32 (&k)[8], int i) {
  const int bit = VB_W * i, wi = bit >> 5, sh = bit & 31;
  u32 lo = k[0], hi = k[1];
  _Pragma("unroll") for (int q = 1; q < 8; q++) { lo = (wi == q) ? k[q] : lo; hi = (wi == q) ? (q < 7 ? k[q + 1] : 0u) : hi; }
  const u64 both = ((u64)hi << 32) | lo;
  return (u32)(both >> sh) & ((1u << VB_W) - 1u);
}
// The lane's table holds VB_SLOTS entries: slot[0] = the identity entry (digit 0), slot[j] = j P for j = 1 .. 2^(w-1): a zero
// digit is a plain table read instead of a 36-instruction select (the sign is applied inside Curve::add_signed).
constexpr int VB_SLOTS = VB_TABLE + 1;

// JJ_VB_PROBE_SHARED_READS (experiments only, WRONG results): every lane reads the table of lane 0 of its workgroup, so the reads
// hit the caches: what the ladder would cost without its table traffic (profiles/r3_varbase_traffic_probe.txt)
static JJ_DEV Ext varbase_windowed(const Affine& P, u32 (&k)[8], u32* slot) {
#ifdef JJ_VB_PROBE_SHARED_READS
  const u32* rslot = slot - (size_t)threadIdx.x * (size_t)((1 << (JJ_VB_W - 1)) + 1) * (4 * NL);
#else
  const u32* rslot = slot;
#endif
  const ANiels pn = Curve::to_niels(P);
  Ext cur = Curve::from_affine(P);
  store_eniels(slot, Curve::eniels_identity());
  store_eniels(slot + ENIELS_WORDS, Curve::to_niels<true>(cur));
  #pragma unroll 1
  for (int j = 2; j <= VB_TABLE; j++) {
    cur = Curve::add<true>(cur, pn);
    store_eniels(slot + j * ENIELS_WORDS, Curve::to_niels<true>(cur));
  }
  recode_signed(k);
  // top window: unsigned digit (0 .. 2^(253 - w (NWIN-1)))
  u32 a = vb_window(k, VB_NWIN - 1), neg = 0;
  ENiels e = load_eniels(rslot + a * ENIELS_WORDS);
  Ext acc = Curve::identity();
  #pragma unroll 1
  for (int i = VB_NWIN - 1; i >= 0; i--) {
    const ENiels s = e;
    const u32 smask = neg ? ~0u : 0u;
    if (i > 0) {                                             // fetch the next window's entry before the doublings
      const int d = (int)vb_window(k, i - 1) - VB_TABLE;
      neg = d < 0; a = (u32)(d < 0 ? -d : d);
      e = load_eniels(rslot + a * ENIELS_WORDS);
    }
    acc = Curve::add_signed<true>(acc, s, smask);
    if (i > 0) {
      // two doublings per trip so that the results can alternate between two register sets (a rolled loop copies 36
      // registers back per doubling)
      // (writing the five doublings out straight -- no loop, no register copies -- is 6.6 % SLOWER: 8 k instructions per window no
      // longer fit the instruction cache; profiles/r3_varbase_traffic_probe.txt)
      #pragma unroll 1
      for (int d = 0; d < (VB_W - 1) / 2; d++) acc = Curve::dbl(Curve::dbl(acc));
      if constexpr ((VB_W - 1) % 2) acc = Curve::dbl(acc);
      acc = Curve::dbl(acc);
    }
  }
  return acc;
}

// Persistent grid, each thread owns one table slot; waves draw their next 64 units from a global cursor (one atomic per
// wave and ladder), so the batch needs no particular relation to the grid size (a static grid-stride split of 2^20 units
// over 196 608 lanes leaves a third of them with 6 ladders and the rest with 5).
static JJ_DEV bool next_wave_units(unsigned long long* cursor, size_t n, size_t& i) {
  unsigned long long base = 0;
  if ((threadIdx.x & 63u) == 0) base = atomicAdd(cursor, 64ull);
  const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)base), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(base >> 32));
  base = ((unsigned long long)hi << 32) | lo;
  i = (size_t)base + (threadIdx.x & 63u);
  return base < n;
}
// SHARED: `scalars` is ONE 32-byte scalar for the whole batch (group::Wnaf's `scalar(..)` then many `base(..)`, reference
// src/lib.rs:1318-1336): it is read through a wave-uniform address, so the recoding and every window digit live in scalar
// registers and cost no vector instruction, and no broadcast buffer is written.
#ifdef JJ_KERNELS_BATCH
template <bool FIVE, bool SHARED>
__global__ void __launch_bounds__(256, JJ_VB_MINWAVES) k_varbase(size_t n, const void* scalars, const void* points, u32* tables, SoA ext, unsigned long long* cursor) {
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u32* slot = tables + gtid * (size_t)(VB_SLOTS * ENIELS_WORDS);
  size_t i;
  #pragma unroll 1
  while (next_wave_units(cursor, n, i)) {
    // ragged last wave: the idle lanes skip the body and meet their neighbours at the end of it, so the cursor draw at the
    // top of the loop is always executed by the whole wave (no early `continue`: that would rely on the compiler
    // reconverging the wave at the loop header)
    if (i < n) {
      u32 k[8];
      load8(k, scalars, SHARED ? (size_t)0 : i);
      const Affine P = load_affine(points, i);
      const Ext r = varbase_windowed(P, k, slot);
      ext.put(0, i, r.u); ext.put(1, i, r.v); ext.put(2, i, r.z);
      if constexpr (FIVE) { ext.put(3, i, Fq::carry(r.t1)); ext.put(4, i, Fq::carry(r.t2)); }
    }
  }
}
#endif  // JJ_KERNELS_BATCH
// the reference's exact ladder; writes all five projective coordinates canonically (160 B)
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_varbase_exact(size_t n, const void* scalars, const void* points, void* out160) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 k[8];
  load8(k, scalars, i);
  const Affine P = load_affine(points, i);
  const ENiels pn = Curve::to_niels<true>(Curve::from_affine(P));
  const ENiels zero = Curve::eniels_identity();
  Ext acc = Curve::identity();
  #pragma unroll 1
  for (int i2 = 251; i2 >= 0; i2--) {
    u32 word = k[0];
    _Pragma("unroll") for (int w = 1; w < 8; w++) word = ((i2 >> 5) == w) ? k[w] : word;
    const u32 bit = (word >> (i2 & 31)) & 1u;
    acc = Curve::dbl(acc);
    acc = Curve::add<true>(acc, Curve::select(zero, pn, 0u - bit));
  }
  u32 w[8];
  Fq::to_words(w, acc.u); store8(out160, 5 * i, w);
  Fq::to_words(w, acc.v); store8(out160, 5 * i + 1, w);
  Fq::to_words(w, acc.z); store8(out160, 5 * i + 2, w);
  Fq::to_words(w, acc.t1); store8(out160, 5 * i + 3, w);
  Fq::to_words(w, acc.t2); store8(out160, 5 * i + 4, w);
}
#endif  // JJ_KERNELS_BATCH

// Constant-time variable-base ladder (jj_varbase_mul_ct): for SECRET scalars on variable bases.  The default ladder above reads
// its per-lane table at a digit-dependent address; the reference ladder (src/lib.rs:357-379 with conditional_select, 334-343) has
// neither secret-dependent branches nor addresses.  Here the table is {P, 2P} as extended-Niels entries held in REGISTERS (there
// is no table in memory at all), windows are signed 2-bit digits d in {-2, -1, 0, 1} (k' = k + sum 2^(2i+1); the top window holds
// the recoding carry: 0 or 1), the entry |d| P is picked with bit masks, a zero digit adds the identity entry, the sign goes
// through Curve::add_signed, and the digits come off a left-aligned shift register: one fixed instruction stream, no load or store
// inside the loop.  127 additions + 252 doublings (reference: 252 + 252), same group element.
constexpr int CT_NWIN = 127;          // 126 signed 2-bit windows over bits 0..251 + the carry window
static JJ_DEV Ext varbase_ct(const Affine& P, u32 (&k)[8]) {
  const Ext p1 = Curve::from_affine(P);
  const ENiels e1 = Curve::to_niels<true>(p1);
  const ENiels e2 = Curve::to_niels<true>(Curve::dbl(p1));
  const ENiels idn = Curve::eniels_identity();
  // recode: k' = (k mod 2^252) + sum_{i<126} 2^(2i+1) = k + 0xaaa...a (252 bits)
  k[7] &= 0x0fffffffu;
  {
    u64 cy = 0;
    _Pragma("unroll") for (int j = 0; j < 8; j++) { const u64 t = (u64)k[j] + (j < 7 ? 0xaaaaaaaau : 0x0aaaaaaau) + cy; k[j] = (u32)t; cy = t >> 32; }
  }
  // top window: bit 252 (0 or 1), unsigned
  const u32 top = (k[7] >> 28) & 1u;
  Ext acc = Curve::add<true>(Curve::identity(), Curve::select(idn, e1, 0u - top));
  // left-align bit 251 at bit 255; every window is then the top two bits of ks[7]
  u32 ks[8];
  _Pragma("unroll") for (int q = 7; q >= 1; q--) ks[q] = (k[q] << 4) | (k[q - 1] >> 28);
  ks[0] = k[0] << 4;
  #pragma unroll 1
  for (int i = CT_NWIN - 2; i >= 0; i--) {
    const int d = (int)(ks[7] >> 30) - 2;                    // window - 2 in [-2, 1]
    _Pragma("unroll") for (int q = 7; q >= 1; q--) ks[q] = (ks[q] << 2) | (ks[q - 1] >> 30);
    ks[0] <<= 2;
    const u32 sgn = (u32)(d >> 31);                          // all-ones iff negative
    const u32 a = ((u32)d ^ sgn) - sgn;                      // |d| in {0, 1, 2}
    const ENiels e = Curve::select(Curve::select(idn, e1, 0u - (a & 1u)), e2, 0u - (a >> 1));
    acc = Curve::dbl(Curve::dbl(acc));
    acc = Curve::add_signed<true>(acc, e, sgn);
  }
  return acc;
}
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_varbase_ct(size_t n, const void* scalars, const void* points, SoA ext) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 k[8];
  load8(k, scalars, i);
  const Affine P = load_affine(points, i);
  const Ext r = varbase_ct(P, k);
  ext.put(0, i, r.u); ext.put(1, i, r.v); ext.put(2, i, r.z);
}
#endif  // JJ_KERNELS_BATCH

// The same discipline with signed 3-bit windows (round 5): digits d in [-4, 3] from k' = k + sum 4 * 8^i (84 windows tile the 252 bits
// exactly; bit 252 holds the recoding carry), table {P, 2P, 3P, 4P}: 84 + 1 additions instead of 126 + 1 for the same 252 doublings.  Four
// extended-Niels entries are 144 registers -- with the accumulator and a product's temporaries more than the 256 a wave may hold at two
// waves per SIMD -- so {P, 2P} stay in registers and {3P, 4P} wait in a per-lane slot of LDS (288 bytes per lane: 18 KB per wave, eight
// waves per CU).  The slot's address depends on the lane only and BOTH entries are read for every window (sixteen-byte pieces, lane-major:
// no bank conflicts, no digit in any address); the digit picks among the five candidates with bit masks.  No load, store, branch or
// address of the loop depends on the scalar.
constexpr int CT3_NWIN = 84;
constexpr int CT3_LDS_WORDS_PER_LANE = 2 * ENIELS_WORDS;                       // 3P, 4P
constexpr int CT3_LDS_BYTES_PER_BLOCK = 256 * CT3_LDS_WORDS_PER_LANE * 4;     // 72 KB per 256-thread workgroup
static JJ_DEV u32 and_or(u32 a, u32 m, u32 c) { return (a & m) | c; }          // v_and_or_b32
static JJ_DEV Ext varbase_ct3(const Affine& P, u32 (&k)[8], uint4* slot /* this lane's first piece; pieces are 64 uint4 apart */, u32* kmem, size_t kstride) {
  const Ext p1 = Curve::from_affine(P);
  const ENiels e1 = Curve::to_niels<true>(p1);
  const Ext p2 = Curve::dbl(p1);
  const ENiels e2 = Curve::to_niels<true>(p2);
  {
    const Ext p3 = Curve::add<true>(p2, e1);
    const ENiels e3 = Curve::to_niels<true>(p3), e4 = Curve::to_niels<true>(Curve::dbl(p2));
    u32 w[CT3_LDS_WORDS_PER_LANE];
    _Pragma("unroll") for (int l = 0; l < NL; l++) {
      w[l] = e3.vpu.l[l]; w[NL + l] = e3.vmu.l[l]; w[2 * NL + l] = e3.z2.l[l]; w[3 * NL + l] = e3.t2d.l[l];
      w[ENIELS_WORDS + l] = e4.vpu.l[l]; w[ENIELS_WORDS + NL + l] = e4.vmu.l[l]; w[ENIELS_WORDS + 2 * NL + l] = e4.z2.l[l]; w[ENIELS_WORDS + 3 * NL + l] = e4.t2d.l[l];
    }
    _Pragma("unroll") for (int v = 0; v < CT3_LDS_WORDS_PER_LANE / 4; v++) slot[v * 64] = make_uint4(w[4 * v], w[4 * v + 1], w[4 * v + 2], w[4 * v + 3]);
  }
  // recode: k' = (k mod 2^252) + sum_{i<84} 4 * 8^i
  k[7] &= 0x0fffffffu;
  {
    constexpr u32 RC[8] = {0x24924924u, 0x49249249u, 0x92492492u, 0x24924924u, 0x49249249u, 0x92492492u, 0x24924924u, 0x09249249u};
    u64 cy = 0;
    _Pragma("unroll") for (int j = 0; j < 8; j++) { const u64 t = (u64)k[j] + RC[j] + cy; k[j] = (u32)t; cy = t >> 32; }
  }
  const ENiels idn = Curve::eniels_identity();
  const u32 top = (k[7] >> 28) & 1u;                           // bit 252: the carry window, 0 or 1
  Ext acc = Curve::add<true>(Curve::identity(), Curve::select(idn, e1, 0u - top));
  // k' waits in memory (the unit's own words of the output array, overwritten by the result at the end): the loop reads the one or two
  // words that hold window i -- an address that depends on the unit and on i only -- instead of keeping eight registers of shift register
  _Pragma("unroll") for (int q = 0; q < 8; q++) kmem[(size_t)q * kstride] = k[q];
  #pragma unroll 1
  for (int i = CT3_NWIN - 1; i >= 0; i--) {
    const int bit = 3 * i, wi = bit >> 5, sh = bit & 31;
    const u32 lo = kmem[(size_t)wi * kstride], hi = kmem[(size_t)(wi < 7 ? wi + 1 : 7) * kstride];
    const int d = (int)((u32)((((u64)hi << 32) | lo) >> sh) & 7u) - 4;      // window - 4 in [-4, 3]
    const u32 sgn = (u32)(d >> 31);                            // all-ones iff negative
    const u32 a = ((u32)d ^ sgn) - sgn;                        // |d| in {0 .. 4}
    u32 m4 = 0u - (a >> 2), m3 = 0u - ((a >> 1) & a & 1u), m2 = 0u - ((a >> 1) & ~a & 1u), m1 = 0u - (a & ~(a >> 1) & 1u);
    asm("" : "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4));          // opaque: plain bit operations, no compare / v_cndmask rebuilt from them
    const u32 m0 = ~(m1 | m2 | m3 | m4);
    acc = Curve::dbl(Curve::dbl(acc));
    acc = Curve::dbl(acc);
    // entry |d| P: the register entries and the identity first, then the two LDS entries piece by piece (four words at a time)
    u32 sel[ENIELS_WORDS];
    _Pragma("unroll") for (int l = 0; l < NL; l++) {
      sel[l] = and_or(e1.vpu.l[l], m1, and_or(e2.vpu.l[l], m2, idn.vpu.l[l] & m0));
      sel[NL + l] = and_or(e1.vmu.l[l], m1, and_or(e2.vmu.l[l], m2, idn.vmu.l[l] & m0));
      sel[2 * NL + l] = and_or(e1.z2.l[l], m1, and_or(e2.z2.l[l], m2, idn.z2.l[l] & m0));
      sel[3 * NL + l] = and_or(e1.t2d.l[l], m1, e2.t2d.l[l] & m2);
    }
    _Pragma("unroll") for (int v = 0; v < ENIELS_WORDS / 4; v++) {
      const uint4 x3 = slot[v * 64], x4 = slot[(ENIELS_WORDS / 4 + v) * 64];
      sel[4 * v] = and_or(x3.x, m3, and_or(x4.x, m4, sel[4 * v]));
      sel[4 * v + 1] = and_or(x3.y, m3, and_or(x4.y, m4, sel[4 * v + 1]));
      sel[4 * v + 2] = and_or(x3.z, m3, and_or(x4.z, m4, sel[4 * v + 2]));
      sel[4 * v + 3] = and_or(x3.w, m3, and_or(x4.w, m4, sel[4 * v + 3]));
    }
    ENiels e;
    _Pragma("unroll") for (int l = 0; l < NL; l++) { e.vpu.l[l] = sel[l]; e.vmu.l[l] = sel[NL + l]; e.z2.l[l] = sel[2 * NL + l]; e.t2d.l[l] = sel[3 * NL + l]; }
    acc = Curve::add_signed<true>(acc, e, sgn);
  }
  return acc;
}
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256, 2) k_varbase_ct3(size_t n, const void* scalars, const void* points, SoA ext) {
  extern __shared__ __attribute__((aligned(16))) uint4 ct3_lds[];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // the wave's region: CT3_LDS_WORDS_PER_LANE / 4 pieces of 64 lanes x 16 bytes; lane L owns piece v at [v * 64 + L]
  uint4* slot = ct3_lds + (size_t)(threadIdx.x >> 6) * (CT3_LDS_WORDS_PER_LANE / 4) * 64 + (threadIdx.x & 63u);
  if (i >= n) return;
  u32 k[8];
  load8(k, scalars, i);
  const Affine P = load_affine(points, i);
  const Ext r = varbase_ct3(P, k, slot, ext.base + i, ext.n);     // (k' is parked in the unit's limbs 0..7 of the U coordinate until the result lands there)
  ext.put(0, i, r.u); ext.put(1, i, r.v); ext.put(2, i, r.z);
}
#endif  // JJ_KERNELS_BATCH

// ------------------------------------------------------------------------------------------------ K4: fixed-base
// Signed 6-bit windows: k = sum_{i<42} d_i 64^i + d_42 64^42, d_i in [-32,31], d_42 in {0,1} (k' = k + 0x820820..).
// Table[i][j] = j * 64^i * B as AffineNiels (27 limbs + 1 pad = 112 B), i < 42, j <= 32 (j = 0: the identity entry, so
// a zero digit is a plain table read), plus one entry for the top carry window.  The whole table (1387 entries,
// 152 KiB) is staged into LDS once per workgroup; every scalar-mul is then 43 mixed additions and no doubling.
constexpr int FB_W = 6;
constexpr int FB_NWIN = 42;
constexpr int FB_ENT = 33;       // entries per window: multiples 0 .. 32
constexpr int FB_ENTRIES = FB_NWIN * FB_ENT + 1;
constexpr int ANIELS_WORDS = 28;   // 27 + 1 pad, 16-byte multiple: entry stride of the LDS table
constexpr int GNIELS_WORDS = 32;   // entry stride of tables gathered from global memory: one 128-byte line per entry
constexpr int FB_LDS_BYTES = FB_ENTRIES * ANIELS_WORDS * 4;

static JJ_DEV ANiels lds_aniels(const u32* e) {
  const uint4* p = reinterpret_cast<const uint4*>(e);
  u32 w[ANIELS_WORDS];
  _Pragma("unroll") for (int v = 0; v < ANIELS_WORDS / 4; v++) { const uint4 x = p[v]; w[4 * v] = x.x; w[4 * v + 1] = x.y; w[4 * v + 2] = x.z; w[4 * v + 3] = x.w; }
  ANiels n;
  _Pragma("unroll") for (int l = 0; l < NL; l++) { n.vpu.l[l] = w[l]; n.vmu.l[l] = w[NL + l]; n.t2d.l[l] = w[2 * NL + l]; }
  return n;
}
// bits [6i, 6i+6) of the 256-bit little-endian integer k
static JJ_DEV u32 window6(const u32 (&k)[8], int i) {
  const int bit = 6 * i, wi = bit >> 5, sh = bit & 31;
  u32 lo = k[0], hi = k[1];
  _Pragma("unroll") for (int w = 1; w < 8; w++) { lo = (wi == w) ? k[w] : lo; hi = (wi == w) ? (w < 7 ? k[w + 1] : 0u) : hi; }
  const u64 both = ((u64)hi << 32) | lo;
  return (u32)(both >> sh) & 63u;
}

// CT = true: constant-time window select.  Lane L of every wave reads entry L (L <= 32; the upper lanes re-read entries
// 0..30) of the current window from LDS (a fixed pattern), and each lane then pulls the entry it needs out of its
// neighbours' registers with ds_bpermute_b32 (a crossbar shuffle: no address- or bank-dependent timing).  The sign
// is applied with bit masks.  CT = false reads the entry directly at a per-lane LDS address.
static JJ_DEV Ext soa_ext(const SoA& s, size_t i);
// chain: bit 0 = start from the point already in `ext` (sums over several fixed bases), bit 1 = also write t1, t2
#ifndef JJ_FB_THREADS
#define JJ_FB_THREADS 512
#endif
#ifndef JJ_FB_SINGLE_BUFFER
#define JJ_FB_SINGLE_BUFFER 0
#endif
constexpr int FB_THREADS = JJ_FB_THREADS;
#ifdef JJ_KERNELS_BATCH
template <bool CT>
__global__ void __launch_bounds__(FB_THREADS) k_fixedbase(size_t n, const void* scalars, const u32* table, SoA ext, int chain) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  {
    const uint4* src = reinterpret_cast<const uint4*>(table);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (int v = threadIdx.x; v < FB_LDS_BYTES / 16; v += blockDim.x) dst[v] = src[v];
  }
  __syncthreads();
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const u32 lane = threadIdx.x & 63u;
  const size_t n_round = (n + 63) & ~(size_t)63;            // whole waves stay in the loop so shuffles see all lanes
  #pragma unroll 1
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_round; idx += T) {
    const bool live = idx < n;
    u32 k[8];
    if (live) load8(k, scalars, idx); else zero8(k);
    // recode: k' = (k mod 2^252) + sum_{i<42} 32 * 64^i
    k[7] &= 0x0fffffffu;
    {
      u64 c = 0;
      _Pragma("unroll") for (int i = 0; i < 8; i++) { const u64 t = (u64)k[i] + RECODE6[i] + c; k[i] = (u32)t; c = t >> 32; }
    }
    const u32 top = (k[7] >> 28) & 1u;                      // d_42
    const ANiels idn = Curve::aniels_identity();
    Ext acc = Curve::identity();
    if ((chain & 1) && live) acc = soa_ext(ext, idx);
    acc = Curve::add<true>(acc, Curve::select(idn, lds_aniels(lds + (size_t)(FB_NWIN * FB_ENT) * ANIELS_WORDS), 0u - top));
    // The 42 windows are consumed from the top: k' is kept left-aligned (bit 251 at bit 255) and shifted by 6 per window,
    // so a digit is the top 6 bits of one register: no indexed access into k and no v_cndmask chains (a v_cndmask that
    // re-reads an old VCC issues at ~22 cycles on gfx950).  The entry of window i-1 is fetched (LDS read + shuffles) before
    // the addition of window i, so the LDS latency hides behind ~1500 multiply-adds.
    u32 ks[8];
    _Pragma("unroll") for (int q = 7; q >= 1; q--) ks[q] = (k[q] << 4) | (k[q - 1] >> 28);
    ks[0] = k[0] << 4;
    auto next_digit = [&](u32& j, u32& negmask) {
      const int d = (int)(ks[7] >> 26) - 32;                 // window - 32 in [-32, 31]
      _Pragma("unroll") for (int q = 7; q >= 1; q--) ks[q] = (ks[q] << 6) | (ks[q - 1] >> 26);
      ks[0] <<= 6;
      const u32 sgn = (u32)(d >> 31);                        // all-ones iff negative
      j = ((u32)d ^ sgn) - sgn;                              // |d| = table index (0 = identity entry)
      negmask = sgn;
    };
    auto fetch = [&](int i, u32 j) -> ANiels {
      if constexpr (CT) {
        const ANiels mine = lds_aniels(lds + ((size_t)i * FB_ENT + (lane < (u32)FB_ENT ? lane : lane - (u32)FB_ENT)) * ANIELS_WORDS);
        const int src = (int)(j << 2);                       // byte address of lane j
        ANiels e;
        _Pragma("unroll") for (int l = 0; l < NL; l++) {
          e.vpu.l[l] = (u32)__builtin_amdgcn_ds_bpermute(src, (int)mine.vpu.l[l]);
          e.vmu.l[l] = (u32)__builtin_amdgcn_ds_bpermute(src, (int)mine.vmu.l[l]);
          e.t2d.l[l] = (u32)__builtin_amdgcn_ds_bpermute(src, (int)mine.t2d.l[l]);
        }
        return e;
      } else {
        return lds_aniels(lds + ((size_t)i * FB_ENT + j) * ANIELS_WORDS);
      }
    };
    // two windows per trip with two entry registers sets (e0, e1), so that "the entry fetched last trip" needs no copy and
    // the wait for the shuffles sits behind a whole addition
    static_assert(FB_NWIN % 2 == 0, "the window loop is unrolled by two");
    u32 j, neg0, neg1;
#if JJ_FB_SINGLE_BUFFER
    #pragma unroll 1
    for (int i = FB_NWIN - 1; i >= 0; i--) {                  // one entry register set (see JJ_FBC_SINGLE_BUFFER)
      next_digit(j, neg0);
      const ANiels e = fetch(i, j);
      acc = Curve::add_signed<true>(acc, e, neg0);
    }
    (void)neg1;
#else
    next_digit(j, neg0);
    ANiels e0 = fetch(FB_NWIN - 1, j), e1;
    #pragma unroll 1
    for (int i = FB_NWIN - 1; i >= 1; i -= 2) {
      next_digit(j, neg1);
      e1 = fetch(i - 1, j);
      acc = Curve::add_signed<true>(acc, e0, neg0);
      if (i > 1) { next_digit(j, neg0); e0 = fetch(i - 2, j); }
      acc = Curve::add_signed<true>(acc, e1, neg1);
    }
#endif
    if (live) {
      ext.put(0, idx, acc.u); ext.put(1, idx, acc.v); ext.put(2, idx, acc.z);
      if (chain & 2) { ext.put(3, idx, Fq::carry(acc.t1)); ext.put(4, idx, Fq::carry(acc.t2)); }
    }
  }
}
#endif  // JJ_KERNELS_BATCH
// ---- Signed comb (jj_fixedbase_table_create window_bits = 7; the default): 8 teeth 32 bits apart, 8 column blocks.
// k (252 bits) is made odd, kk = k | 1, and written with digits +-1 only: kk = sum_{p<256} s_p 2^p, s_255 = +1, s_p = +1 iff bit
// p + 1 of kk is set (p < 255).  Column j (0..31) collects the eight signs s_{j + 32 i}, i.e. bit j of the eight 32-bit words of
// kk >> 1 | 2^255: its value is +-(2^224 + sum_{i<7} +-2^(32 i)) = sign x one of 128 table entries T[idx],
// idx_i = (s_{j+32i} == s_{j+224}).  Columns are grouped in 8 blocks of four (j = 4 j1 + j0) with their own tables
// T_{j1} = 2^(4 j1) T:   kk B = sum_{j0<4} 2^j0 sum_{j1<8} +-T_{j1}[idx_{4 j1 + j0}]      -- 32 mixed additions and 3 doublings
// (signed 6-bit windows above: 43 additions).  For an even k the last addition (column 0) takes its entry from T_0 -+ B instead of
// T_0 (two more tables), which removes the B that kk = k + 1 added: no extra addition.  10 tables x 128 entries x 112 B = 140 KiB
// of LDS.  Constant-time select: a table is staged as two halves of 64 entries, lane L holds entries L and 64 + L, both halves
// are shuffled with ds_bpermute and the top index bit picks one with bit masks.  Same group element as the reference's
// AffineNielsPoint::multiply (src/lib.rs:272-310) for every base point of the curve (only sums of multiples of B are formed; no
// assumption on its order).
constexpr int FBC_TEETH = 8, FBC_SPACING = 32, FBC_BLOCKS = 8, FBC_COLS = 4;      // 8 x 32 = 256 signed bits; 8 blocks x 4 columns
constexpr int FBC_TENT = 128;                                // entries per table: 2^(teeth - 1)
constexpr int FBC_TABLES = FBC_BLOCKS + 2;                   // + T_0 - B, T_0 + B
constexpr int FBC_ENTRIES = FBC_TABLES * FBC_TENT;
constexpr int FBC_LDS_BYTES = FBC_ENTRIES * ANIELS_WORDS * 4;
// Workgroup size and entry buffering of the comb kernel.  Round 3: 512 threads (two waves per SIMD) and two entry register sets (the entry
// of step t + 1 shuffled in before the addition of step t: 174 VGPRs).  Round 4 (profiles/r4_fixedbase_select_pmc.txt, LDS probe): the
// shuffles' LDS time is not hidden at two waves per SIMD (+14 % in the probe, +4 % at three), so: 768 threads, and ONE entry register
// set (145 VGPRs: three waves per SIMD fit without a spill; the other waves cover the shuffle latency instead of the software prefetch).
// Same box: 604-612 -> 620-625 M/s (512 threads + one set: 611-618; 1024 threads: 128 VGPRs, 30 spilled, 601-605).
#ifndef JJ_FBC_THREADS
#define JJ_FBC_THREADS 768
#endif
#ifndef JJ_FBC_SINGLE_BUFFER
#define JJ_FBC_SINGLE_BUFFER 1
#endif
constexpr int FBC_THREADS = JJ_FBC_THREADS;                  // one workgroup per CU (the table fills the LDS): waves per SIMD = FBC_THREADS / 256
#ifdef JJ_KERNELS_BATCH
template <bool CT>
__global__ void __launch_bounds__(FBC_THREADS) k_fixedbase_comb(size_t n, const void* scalars, const u32* table, SoA ext, int chain) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  {
    const uint4* src = reinterpret_cast<const uint4*>(table);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (int v = threadIdx.x; v < FBC_LDS_BYTES / 16; v += blockDim.x) dst[v] = src[v];
  }
  __syncthreads();
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const u32 lane = threadIdx.x & 63u;
  const size_t n_round = (n + 63) & ~(size_t)63;            // whole waves stay in the loop so shuffles see all lanes
  #pragma unroll 1
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_round; idx += T) {
    const bool live = idx < n;
    u32 k[8];
    if (live) load8(k, scalars, idx); else zero8(k);
    k[7] &= 0x0fffffffu;
    const u32 even = 0u - ((k[0] & 1u) ^ 1u);                // all-ones iff k is even
    // sw = (kk >> 1) | 2^255, kk = k | 1: tooth i is word i, column j is bit j of every word
    u32 sw[8];
    _Pragma("unroll") for (int q = 0; q < 7; q++) sw[q] = (k[q] >> 1) | (k[q + 1] << 31);
    sw[7] = (k[7] >> 1) | 0x80000000u;
    // 8 x 32 bit-matrix transpose (three masked-swap stages on the eight words: every byte lane is an 8 x 8 transpose): afterwards
    // column j is byte j >> 3 of word j & 7, bit i of the byte = tooth i
    u32 a[8];
    _Pragma("unroll") for (int i = 0; i < 8; i++) a[i] = sw[i];
    _Pragma("unroll") for (int i = 0; i < 8; i += 2) { const u32 t = ((a[i] >> 1) ^ a[i + 1]) & 0x55555555u; a[i + 1] ^= t; a[i] ^= t << 1; }
    _Pragma("unroll") for (int i = 0; i < 8; i++) { if (i & 2) continue; const u32 t = ((a[i] >> 2) ^ a[i + 2]) & 0x33333333u; a[i + 2] ^= t; a[i] ^= t << 2; }
    _Pragma("unroll") for (int i = 0; i < 4; i++) { const u32 t = ((a[i] >> 4) ^ a[i + 4]) & 0x0f0f0f0fu; a[i + 4] ^= t; a[i] ^= t << 4; }
    // a column whose top tooth is -1 is minus the entry of the complemented bits: complement the low 7 bits of those bytes; the top
    // bit stays (1 = the entry is added, 0 = subtracted)
    _Pragma("unroll") for (int i = 0; i < 8; i++) { const u32 m = (~a[i] >> 7) & 0x01010101u; a[i] ^= (m << 7) - m; }
    // digits in the order the steps consume them, step 0 in the top byte of pk[7]: phase ph adds the columns j0 = 3 - ph of the
    // blocks j1 = 7 .. 0 (column 4 j1 + j0 = byte j1 >> 1 of word j0 + 4 (j1 & 1)), a doubling between the phases
    u32 pk[8];
    _Pragma("unroll") for (int ph = 0; ph < 4; ph++) {
      const u32 hi = a[(3 - ph) + 4], lo = a[3 - ph];
      pk[7 - 2 * ph] = (hi & 0xff000000u) | ((lo >> 8) & 0x00ff0000u) | ((hi >> 8) & 0x0000ff00u) | ((lo >> 16) & 0x000000ffu);
      pk[6 - 2 * ph] = ((hi << 16) & 0xff000000u) | ((lo << 8) & 0x00ff0000u) | ((hi << 8) & 0x0000ff00u) | (lo & 0x000000ffu);
    }
    auto next_digit = [&](u32& j, u32& negmask) {
      const u32 d = pk[7] >> 24;
      _Pragma("unroll") for (int q = 7; q >= 1; q--) pk[q] = (pk[q] << 8) | (pk[q - 1] >> 24);
      pk[0] <<= 8;
      j = d & 127u;
      negmask = (d >> 7) - 1u;                                // top tooth +: 0, -: all-ones
    };
    auto bperm = [&](const ANiels& mine, u32 j) -> ANiels {
      const int src = (int)((j & 63u) << 2);                 // byte address of lane j mod 64
      ANiels e;
      _Pragma("unroll") for (int l = 0; l < NL; l++) {
        e.vpu.l[l] = (u32)__builtin_amdgcn_ds_bpermute(src, (int)mine.vpu.l[l]);
        e.vmu.l[l] = (u32)__builtin_amdgcn_ds_bpermute(src, (int)mine.vmu.l[l]);
        e.t2d.l[l] = (u32)__builtin_amdgcn_ds_bpermute(src, (int)mine.t2d.l[l]);
      }
      return e;
    };
    auto fetch = [&](int tab, u32 j) -> ANiels {
      const u32* tb = lds + (size_t)tab * FBC_TENT * ANIELS_WORDS;
#if defined(JJ_EXPERIMENTS) && defined(JJ_FBC_PROBE)
      // WRONG results, probe builds only (tools/fixedbase_floor.sh): what the comb costs without its select (1: the lane's own staged entry) and with
      // one shuffle round instead of two (2: the low half of the table only)
      if (JJ_FBC_PROBE == 1) return lds_aniels(tb + (size_t)lane * ANIELS_WORDS);
      if (JJ_FBC_PROBE == 2) return bperm(lds_aniels(tb + (size_t)lane * ANIELS_WORDS), j);
#endif
      if constexpr (CT) {
        const ANiels lo = bperm(lds_aniels(tb + (size_t)lane * ANIELS_WORDS), j);
        return Curve::select(lo, bperm(lds_aniels(tb + (size_t)(64 + lane) * ANIELS_WORDS), j), 0u - (j >> 6));
      } else return lds_aniels(tb + (size_t)j * ANIELS_WORDS);
    };
    // column 0: T_0 for an odd k, T_0 - B (+ digit) or T_0 + B (- digit) for an even one: sign * (T_0 -+ B) = +-T_0 - B
    auto fetch_last = [&](u32 j, u32 negmask) -> ANiels {
      if constexpr (CT) {
        // the choice of table depends on THIS lane's scalar while the staged entries belong to the lanes they came from: every
        // lane stages and shuffles all three tables in turn (fixed addresses), keeping the one its parity and sign call for
        ANiels e = fetch(0, j);
        e = Curve::select(e, fetch(FBC_BLOCKS, j), even & ~negmask);
        e = Curve::select(e, fetch(FBC_BLOCKS + 1, j), even & negmask);
        return e;
      } else {
        const u32 tab = even ? (negmask ? (u32)FBC_BLOCKS + 1u : (u32)FBC_BLOCKS) : 0u;
        return lds_aniels(lds + ((size_t)tab * FBC_TENT + j) * ANIELS_WORDS);
      }
    };
    Ext acc = Curve::identity();
    u32 j, neg0, neg1;
#if JJ_FBC_SINGLE_BUFFER
    // one entry register set: the entry of a step is fetched right before its addition and the other waves of the SIMD cover the shuffles'
    // latency (three waves per SIMD with 768-thread workgroups need the 27 registers the second set takes)
    #pragma unroll 1
    for (int t = 0; t < 32; t++) {
      next_digit(j, neg0);
      const ANiels e = (t == 31) ? fetch_last(j, neg0) : fetch(FBC_BLOCKS - 1 - (t & 7), j);
      acc = Curve::add_signed<true>(acc, e, neg0);
      if (((t + 1) & 7) == 0 && t + 1 < 32) acc = Curve::dbl(acc);
    }
    (void)neg1;
#else
    ANiels e0, e1;
    next_digit(j, neg0);
    e0 = fetch(FBC_BLOCKS - 1, j);
    // 32 steps, two per trip with two entry register sets (the entry of step t + 1 is fetched before the addition of step t); step t
    // uses table 7 - (t & 7); after every 8 steps (one column of every block) the sum is doubled
    #pragma unroll 1
    for (int t = 0; t < 32; t += 2) {
      next_digit(j, neg1);
      e1 = (t + 1 == 31) ? fetch_last(j, neg1) : fetch(FBC_BLOCKS - 1 - ((t + 1) & 7), j);
      acc = Curve::add_signed<true>(acc, e0, neg0);
      if (t + 2 < 32) { next_digit(j, neg0); e0 = fetch(FBC_BLOCKS - 1 - ((t + 2) & 7), j); }
      acc = Curve::add_signed<true>(acc, e1, neg1);
      if (((t + 2) & 7) == 0 && t + 2 < 32) acc = Curve::dbl(acc);
    }
#endif
    // sums over several bases (chain): the previous sum is added after the comb -- the doublings between the phases must not touch it
    if ((chain & 1) && live) acc = Curve::add<true>(acc, Curve::to_niels<false>(soa_ext(ext, idx)));
    if (live) {
      ext.put(0, idx, acc.u); ext.put(1, idx, acc.v); ext.put(2, idx, acc.z);
      if (chain & 2) { ext.put(3, idx, Fq::carry(acc.t1)); ext.put(4, idx, Fq::carry(acc.t2)); }
    }
  }
}
#endif  // JJ_KERNELS_BATCH
// ---- Several fixed bases with SHORT scalars in ONE pass over ONE LDS table set (Pedersen-style sums
// sum_b k_b B_b built from AffineNielsPoint::multiply_bits, reference src/lib.rs:297-301).  The windows of k_fixedbase need not
// belong to one base: a composite table gives window slots [off_b, off_b + W_b) to base b (entries j 64^(i - off_b) B_b), and
// the short scalars are packed into one 252-bit virtual scalar, field b at bit 6 off_b.  With 6 W_b >= bits_b + 2 the signed
// recoding of a field never carries into the next field (k' of the field stays below 64^W_b), so k_fixedbase -- one accumulator
// per lane, 43 additions, constant-time shuffle select -- computes sum_b (k_b mod 2^bits_b) B_b unchanged.
constexpr int FBX_MAX_BASES = 21;      // 42 window slots, at least two per base
struct FbxParams { int nb; int off[FBX_MAX_BASES]; int bits[FBX_MAX_BASES]; };
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_pack_composite(size_t n, const void* scalars, FbxParams fx, void* virt32) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 v[8];
  zero8(v);
  #pragma unroll 1
  for (int b = 0; b < fx.nb; b++) {
    u32 k[8];
    load8(k, scalars, (size_t)b * n + i);
    const int bits = fx.bits[b], sh = 6 * fx.off[b], ws = sh >> 5, bs = sh & 31;
    _Pragma("unroll") for (int q = 0; q < 8; q++) {                  // keep the low `bits` bits
      const int lo = 32 * q;
      const u32 m = bits >= lo + 32 ? ~0u : (bits <= lo ? 0u : ((1u << (bits - lo)) - 1u));
      k[q] &= m;
    }
    _Pragma("unroll") for (int q = 0; q < 8; q++) {                  // v |= k << sh   (a field never reaches bit 252)
      const int src = q - ws;
      u32 lo = 0, hi = 0;
      _Pragma("unroll") for (int r = 0; r < 8; r++) { lo = (r == src) ? k[r] : lo; hi = (r == src - 1) ? k[r] : hi; }
      v[q] |= (lo << bs) | (bs ? (hi >> (32 - bs)) : 0u);
    }
  }
  store8(virt32, i, v);
}
#endif  // JJ_KERNELS_BATCH

// Wide-window variant: table of (j+1) * 2^(w i) * B for w = 8..12 (0.5 - 5 MB) kept in global memory, L2-resident;
// each lane gathers its entry (7 x dwordx4 of one 128-byte line) one window ahead of its use.  Fewer additions than the LDS
// kernel (w = 10: 26 instead of 43) at the price of a secret-dependent address (documented as variable-time).
struct FbParams {
  int w, W;          // window bits, number of windows = ceil(253 / w)
  u32 E;             // largest multiple per window = 2^(w-1); a window holds E + 1 entries (multiples 0 .. E)
  u32 recode[8];     // sum_{i<W-1} 2^(w i + w - 1)
};
static JJ_DEV u32 fb_window(const u32 (&k)[8], int w, int i) {
  const int bit = w * i, wi = bit >> 5, sh = bit & 31;
  u32 lo = k[0], hi = k[1];
  _Pragma("unroll") for (int q = 1; q < 8; q++) { lo = (wi == q) ? k[q] : lo; hi = (wi == q) ? (q < 7 ? k[q + 1] : 0u) : hi; }
  const u64 both = ((u64)hi << 32) | lo;
  return (u32)(both >> sh) & ((1u << w) - 1u);
}
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_fixedbase_gather(size_t n, const void* scalars, const u32* table, FbParams fp, SoA ext, int chain) {
  const size_t T = (size_t)gridDim.x * blockDim.x;
  #pragma unroll 1
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += T) {
    u32 k[8];
    load8(k, scalars, idx);
    k[7] &= 0x0fffffffu;
    {
      u64 c = 0;
      _Pragma("unroll") for (int i = 0; i < 8; i++) { const u64 t = (u64)k[i] + fp.recode[i] + c; k[i] = (u32)t; c = t >> 32; }
    }
    Ext acc = Curve::identity();
    if (chain & 1) acc = soa_ext(ext, idx);
    // top window: unsigned digit
    u32 a = fb_window(k, fp.w, fp.W - 1), neg = 0;
    ANiels e = lds_aniels(table + ((size_t)(fp.W - 1) * (fp.E + 1) + a) * GNIELS_WORDS);
    #pragma unroll 1
    for (int i = fp.W - 1; i >= 0; i--) {
      const ANiels s = e;
      const u32 smask = neg ? ~0u : 0u;
      if (i > 0) {                                           // fetch the next window's entry before this addition
        const int d = (int)fb_window(k, fp.w, i - 1) - (int)fp.E;
        neg = d < 0; a = (u32)(d < 0 ? -d : d);
        e = lds_aniels(table + ((size_t)(i - 1) * (fp.E + 1) + a) * GNIELS_WORDS);
      }
      acc = Curve::add_signed<true>(acc, s, smask);
    }
    ext.put(0, idx, acc.u); ext.put(1, idx, acc.v); ext.put(2, idx, acc.z);
    if (chain & 2) { ext.put(3, idx, Fq::carry(acc.t1)); ext.put(4, idx, Fq::carry(acc.t2)); }
  }
}
#endif  // JJ_KERNELS_BATCH
// affine points (64 B canonical) -> table entries (AffineNiels limbs, 112 B)
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_affine_to_table(size_t n, const void* pts, u32* table, int stride) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const ANiels t = Curve::to_niels(load_affine(pts, i));
  u32* e = table + i * (size_t)stride;
  // a deterministic function of the base point: v+u carried, v-u with signed limbs, t2d a product
  const Fe a = t.vpu, b = t.vmu, c = t.t2d;
  _Pragma("unroll") for (int l = 0; l < NL; l++) { e[l] = a.l[l]; e[NL + l] = b.l[l]; e[2 * NL + l] = c.l[l]; }
  for (int l = 27; l < stride; l++) e[l] = 0;
}
#endif  // JJ_KERNELS_BATCH

// ------------------------------------------------------------------------------------------------ sums (MSM tail, point_sum)
// Each lane folds FOLD strided extended points (Ext + Ext = to_niels + add, reference src/lib.rs:992-999).
#ifdef JJ_KERNELS_BATCH
template <int FOLD>
__global__ void __launch_bounds__(256) k_sum_pass(size_t n, size_t T, SoA in, SoA out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  Ext acc = Curve::identity();
  #pragma unroll 1
  for (int j = 0; j < FOLD; j++) {
    const size_t i = t + (size_t)j * T;
    if (i >= n) break;
    Ext e; e.u = in.get(0, i); e.v = in.get(1, i); e.z = in.get(2, i); e.t1 = in.get(3, i); e.t2 = in.get(4, i);
    acc = Curve::add<true>(acc, Curve::to_niels<true>(e));
  }
  out.put(0, t, acc.u); out.put(1, t, acc.v); out.put(2, t, acc.z); out.put(3, t, Fq::carry(acc.t1)); out.put(4, t, Fq::carry(acc.t2));
}
__global__ void __launch_bounds__(256) k_affine_to_soa5(size_t n, const void* pts, SoA ext) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Affine a = load_affine(pts, i);
  ext.put(0, i, a.u); ext.put(1, i, a.v); ext.put(2, i, Fq::one()); ext.put(3, i, a.u); ext.put(4, i, a.v);
}
#endif  // JJ_KERNELS_BATCH
// ------------------------------------------------------------------------------------------------ K6: decompress
// AffinePoint::from_bytes_inner / batch_from_bytes (reference src/lib.rs:492-534, 541-627): u^2 = (v^2-1)/(1+d v^2).
// Like batch_from_bytes the denominators share one inversion: each lane owns CHUNK strided encodings, multiplies
// their (never-zero) denominators through, inverts once and walks back, recomputing v, v^2 on the way (cheaper
// than storing them).  The square root is fq_sqrt_fast; the sign bit picks the root (lib.rs:518-520) and the
// ZIP-216 rule rejects u = 0 with the sign bit set (lib.rs:522-531).
static JJ_DEV void decode_v(const void* in32, size_t i, Fe& v, Fe& v2, Fe& den, u32& sign, bool& ok, u32 (&w)[8]) {
  load8(w, in32, i);
  sign = w[7] >> 31;
  w[7] &= 0x7fffffffu;
  v = Fq::from_words_checked(w, ok);
  v2 = Fq::sqr(v);
  den = Fq::carry(Fq::add(Fq::one(), Fq::mul(Fq::konst(FqP::D), v2)));
}
#ifdef JJ_KERNELS_BATCH
template <int CHUNK>
__global__ void __launch_bounds__(256) k_decompress(size_t n, size_t T, const void* in32, unsigned flags, SoA scratch, SqrtTables tabs,
                                                     void* out64, uint8_t* okp) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  Fe acc = Fq::one();
  #pragma unroll 1
  for (int j = 0; j < CHUNK; j++) {
    const size_t i = t + (size_t)j * T;
    if (i >= n) break;
    Fe v, v2, den; u32 sign, wv[8]; bool ok;
    decode_v(in32, i, v, v2, den, sign, ok, wv);
    scratch.put(0, i, acc);
    acc = Fq::mul(acc, den);
  }
  Fe inv = Fq::invert(acc);
  #pragma unroll 1
  for (int j = CHUNK - 1; j >= 0; j--) {
    const size_t i = t + (size_t)j * T;
    if (i >= n) continue;
    Fe v, v2, den; u32 sign, wv[8]; bool ok;
    decode_v(in32, i, v, v2, den, sign, ok, wv);         // wv: the encoding without its sign bit = the canonical bytes of v when ok
    const Fe deninv = Fq::mul(inv, scratch.get(0, i));
    inv = Fq::mul(inv, den);
    const Fe u2 = Fq::mul(Fq::sub(v2, Fq::one()), deninv);
    bool sq_ok;
    const Fe u = fq_sqrt_fast(u2, sq_ok, tabs);
    ok = ok && sq_ok;
    // canonical u once; -u is q - u on the plain limbs (no second conversion product)
    const Fe up = Fq::to_plain(u);
    u32 nzl = 0; _Pragma("unroll") for (int k = 0; k < NL; k++) nzl |= up.l[k];
    const u32 flip = (up.l[0] ^ sign) & 1u;
    if ((flags & 1u) && nzl == 0 && flip) ok = false;
    Fe un; _Pragma("unroll") for (int k = 0; k < NL; k++) un.l[k] = (nzl ? FqP::P[k] : 0u) - up.l[k];
    u32 wu[8];
    Fq::pack(wu, Fq::select(up, Fq::carry_full(un), flip ? ~0u : 0u));
    _Pragma("unroll") for (int k = 0; k < 8; k++) { if (!ok) { wu[k] = 0; wv[k] = 0; } }
    store8(out64, 2 * i, wu);
    store8(out64, 2 * i + 1, wv);
    okp[i] = ok ? 1 : 0;
  }
}
#endif  // JJ_KERNELS_BATCH
// AffinePoint::to_bytes (reference src/lib.rs:455-464) for affine input
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_compress(size_t n, const void* pts, void* out32) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 wu[8], wv[8], t[8];
  load8(t, pts, 2 * i); Fq::to_words(wu, Fq::from_words(t));
  load8(t, pts, 2 * i + 1); Fq::to_words(wv, Fq::from_words(t));
  wv[7] |= (wu[0] & 1u) << 31;
  store8(out32, i, wv);
}
#endif  // JJ_KERNELS_BATCH
// zero the 64-byte outputs whose ok byte is 0
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_mask_outputs(size_t n, void* out64, const uint8_t* ok) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!ok[i]) { u32 z[8]; zero8(z); store8(out64, 2 * i, z); store8(out64, 2 * i + 1, z); }
}
__global__ void __launch_bounds__(256) k_fill_scalar(size_t n, void* scalars, const uint8_t* pattern32) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[8];
  load8(w, pattern32, 0);
  store8(scalars, i, w);
}
__global__ void __launch_bounds__(256) k_and_bytes(size_t n, uint8_t* a, const uint8_t* b, int negate_b) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  a[i] = a[i] & (negate_b ? (b[i] ^ 1) : b[i]);
}
#endif  // JJ_KERNELS_BATCH

// ------------------------------------------------------------------------------------------------ point records + quad-lane point operations
// (shared by the small-batch ladder and the MSM kernels in jj_msm_kernels.h)
static JJ_DEV void soa_put_ext(const SoA& s, size_t i, const Ext& e);
// Buckets and chunk heads are written from divergent code (a lane flushes whenever its run of equal buckets ends), so
// they are kept as one 192-byte record per point (U V Z T1 T2, 9 limbs each, 3 words of padding): 12 16-byte
// stores from one base address instead of 45 strided dword stores.
constexpr int EXT_AOS_WORDS = 48;
struct ExtAoS { u32* p; };
static JJ_DEV void aos_put_ext(const ExtAoS& a, size_t i, const Ext& e) {
  const Fe t1 = Fq::carry(e.t1), t2 = Fq::carry(e.t2);
  u32 w[EXT_AOS_WORDS];
  _Pragma("unroll") for (int l = 0; l < NL; l++) { w[l] = e.u.l[l]; w[NL + l] = e.v.l[l]; w[2 * NL + l] = e.z.l[l]; w[3 * NL + l] = t1.l[l]; w[4 * NL + l] = t2.l[l]; }
  w[45] = w[46] = w[47] = 0;
  uint4* d = reinterpret_cast<uint4*>(a.p + i * EXT_AOS_WORDS);
  _Pragma("unroll") for (int v = 0; v < EXT_AOS_WORDS / 4; v++) d[v] = make_uint4(w[4 * v], w[4 * v + 1], w[4 * v + 2], w[4 * v + 3]);
}
static JJ_DEV Ext aos_ext(const ExtAoS& a, size_t i) {
  const uint4* d = reinterpret_cast<const uint4*>(a.p + i * EXT_AOS_WORDS);
  u32 w[EXT_AOS_WORDS];
  _Pragma("unroll") for (int v = 0; v < EXT_AOS_WORDS / 4; v++) { const uint4 x = d[v]; w[4 * v] = x.x; w[4 * v + 1] = x.y; w[4 * v + 2] = x.z; w[4 * v + 3] = x.w; }
  Ext e;
  _Pragma("unroll") for (int l = 0; l < NL; l++) { e.u.l[l] = w[l]; e.v.l[l] = w[NL + l]; e.z.l[l] = w[2 * NL + l]; e.t1.l[l] = w[3 * NL + l]; e.t2.l[l] = w[4 * NL + l]; }
  return e;
}
static JJ_DEV Ext soa_ext(const SoA& s, size_t i) { Ext e; e.u = s.get(0, i); e.v = s.get(1, i); e.z = s.get(2, i); e.t1 = s.get(3, i); e.t2 = s.get(4, i); return e; }
static JJ_DEV void soa_put_ext(const SoA& s, size_t i, const Ext& e) {
  s.put(0, i, e.u); s.put(1, i, e.v); s.put(2, i, e.z); s.put(3, i, Fq::carry(e.t1)); s.put(4, i, Fq::carry(e.t2));
}
// The tails (bucket reduce, folds, big-bucket fix-up) are serial chains of point operations bound by the instructions one wave
// has to issue (experiments/lone_wave), so each point operation is spread over the four lanes of a quad: lane r squares {U, V, Z, U-V}[r], the four
// squares are broadcast inside the quad with DPP quad_perm moves, every lane forms the completed point, and lane r
// multiplies one of (cu*ct, cv*cz, cz*ct).  Same formulas as Curve::dbl (reference src/lib.rs:739-828), ~2.4x fewer
// instructions on the critical path.
template <int K>
static JJ_DEV Fe quad_bcast(const Fe& a) {   // value held by lane K of each quad -> all four lanes
  Fe r;
  _Pragma("unroll") for (int l = 0; l < NL; l++) r.l[l] = (u32)__builtin_amdgcn_mov_dpp((int)a.l[l], K * 0x55, 0xf, 0xf, false);
  return r;
}
static JJ_DEV Fe role_select4(const Fe& a0, const Fe& a1, const Fe& a2, const Fe& a3, u32 role) {
  Fe r = Fq::select(a0, a1, role == 1 ? ~0u : 0u);
  r = Fq::select(r, a2, role == 2 ? ~0u : 0u);
  return Fq::select(r, a3, role == 3 ? ~0u : 0u);
}
// the doubling's completed point from the four squares UU, VV, ZZ, (U-V)^2 (same steps as Curve::dbl)
static JJ_DEV void quad_dbl_completed(const Fe& sq, Fe& cu, Fe& vpu, Fe& vmu, Fe& ct) {
  const Fe uu = quad_bcast<0>(sq), vv = quad_bcast<1>(sq), zz = quad_bcast<2>(sq), s = quad_bcast<3>(sq);
  vpu = Fq::add(vv, uu);
  vmu = Fq::sub(vv, uu);
  cu = Fq::sub(vpu, s);                                   // 2UV
  ct = Fq::carry(Fq::sub(Fq::add(zz, zz), vmu));          // 2Z^2 - (VV-UU)
}
static JJ_DEV Ext quad_dbl(const Ext& p, u32 role) {
  const Fe sq = Fq::sqr(role_select4(p.u, p.v, p.z, Fq::sub(p.u, p.v), role));
  Fe cu, vpu, vmu, ct;
  quad_dbl_completed(sq, cu, vpu, vmu, ct);
  // lane 0: cu*ct   lane 1: cv*cz   lane 2,3: cz*ct
  const Fe pr = Fq::mul(role_select4(cu, vpu, vmu, vmu, role), role_select4(ct, vmu, ct, ct, role));
  Ext r;
  r.u = quad_bcast<0>(pr); r.v = quad_bcast<1>(pr); r.z = quad_bcast<2>(pr); r.t1 = cu; r.t2 = vpu;
  return r;
}
// Extended + Extended on a quad (reference src/lib.rs:992-999 = to_niels + Add<&ExtendedNielsPoint>, 883-920):
// four multiplication rounds instead of ten serial multiplications.
static JJ_DEV Ext quad_add_ext(const Ext& p, const Ext& q, u32 role) {
  // round 1: lane0 p.t1*p.t2, lane1 q.t1*q.t2, lane2/3 p.z*q.z
  const Fe r1 = Fq::mul(role_select4(Fq::carry(p.t1), Fq::carry(q.t1), p.z, p.z, role), role_select4(p.t2, q.t2, q.z, q.z, role));
  const Fe ttp = quad_bcast<0>(r1), ttq = quad_bcast<1>(r1), zz = quad_bcast<2>(r1);
  // round 2: lane0 (Vp-Up)(Vq-Uq), lane1 (Vp+Up)(Vq+Uq), lane2/3 ttp*ttq
  const Fe r2 = Fq::mul(role_select4(Fq::sub(p.v, p.u), Fq::add(p.v, p.u), ttp, ttp, role),
                        role_select4(Fq::sub(q.v, q.u), Fq::carry(Fq::add(q.v, q.u)), ttq, ttq, role));
  const Fe a = quad_bcast<0>(r2), b = quad_bcast<1>(r2), tpq = quad_bcast<2>(r2);
  // round 3: c = 2d * ttp * ttq  (every lane)
  const Fe c = Fq::mul(tpq, Fq::konst(FqP::D2));
  const Fe d = Fq::add(zz, zz);
  const Fe cu = Fq::sub(b, a), cv = Fq::add(b, a), cz = Fq::carry(Fq::add(d, c)), ct = Fq::sub(d, c);
  // round 4: lane0 cu*ct, lane1 cv*cz, lane2/3 cz*ct
  const Fe r4 = Fq::mul(role_select4(cu, cv, cz, cz, role), role_select4(ct, cz, ct, ct, role));
  Ext r;
  r.u = quad_bcast<0>(r4); r.v = quad_bcast<1>(r4); r.z = quad_bcast<2>(r4); r.t1 = cu; r.t2 = cv;
  return r;
}
// ---- quad variants that keep T = t1*t2 alongside the point, for adding table entries in two rounds
// doubling; lane 3 multiplies carry(cu)*cv so the caller also gets T of the result
static JJ_DEV Ext quad_dbl_t(const Ext& p, u32 role, Fe& T) {
  const Fe sq = Fq::sqr(role_select4(p.u, p.v, p.z, Fq::sub(p.u, p.v), role));
  Fe cu, vpu, vmu, ct;
  quad_dbl_completed(sq, cu, vpu, vmu, ct);
  const Fe pr = Fq::mul(role_select4(cu, vpu, vmu, Fq::carry(cu), role), role_select4(ct, vmu, ct, vpu, role));
  Ext r;
  r.u = quad_bcast<0>(pr); r.v = quad_bcast<1>(pr); r.z = quad_bcast<2>(pr); r.t1 = cu; r.t2 = vpu;
  T = quad_bcast<3>(pr);
  return r;
}
// shared second half of the quad additions: a, b, c, d -> result and its T (lane 3: cu*cv; cu = b - a is small)
static JJ_DEV Ext quad_add_finish(const Fe& a, const Fe& b, const Fe& c, const Fe& d, u32 role, Fe& Tout) {
  const Fe cu = Fq::sub(b, a), cv = Fq::add(b, a), cz = Fq::carry(Fq::add(d, c)), ct = Fq::sub(d, c);
  const Fe r2 = Fq::mul(role_select4(cu, cv, cz, cu, role), role_select4(ct, cz, ct, cv, role));
  Ext r;
  r.u = quad_bcast<0>(r2); r.v = quad_bcast<1>(r2); r.z = quad_bcast<2>(r2); r.t1 = cu; r.t2 = cv;
  Tout = quad_bcast<3>(r2);
  return r;
}
// p (+/-) n for an extended-Niels operand, given T = p.t1*p.t2 (reference src/lib.rs:883-940): round 1 a, b, c = T*t2d,
// d = Z*2Z' on the four lanes; round 2 U, V, Z, T.  neg selects the subtraction formula (swap v+u / v-u, negate c).
static JJ_DEV Ext quad_add_eniels(const Ext& p, const Fe& T, const ENiels& n, u32 neg, u32 role, Fe& Tout) {
  const u32 nm = neg ? ~0u : 0u;
  const Fe fa = Fq::select(n.vmu, n.vpu, nm), fb = Fq::select(n.vpu, n.vmu, nm);
  const Fe r1 = Fq::mul(role_select4(Fq::sub(p.v, p.u), Fq::add(p.v, p.u), T, p.z, role), role_select4(fa, fb, n.t2d, n.z2, role));
  const Fe a = quad_bcast<0>(r1), b = quad_bcast<1>(r1), c = Fq::cneg(quad_bcast<2>(r1), nm), d = quad_bcast<3>(r1);
  return quad_add_finish(a, b, c, d, role, Tout);
}
// p + n for an affine-Niels operand (Z2 = 1: d = 2 Z1; reference src/lib.rs:944-968)
static JJ_DEV Ext quad_add_aniels(const Ext& p, const Fe& T, const ANiels& n, u32 role, Fe& Tout) {
  const Fe r1 = Fq::mul(role_select4(Fq::sub(p.v, p.u), Fq::add(p.v, p.u), T, T, role), role_select4(n.vmu, n.vpu, n.t2d, n.t2d, role));
  const Fe a = quad_bcast<0>(r1), b = quad_bcast<1>(r1), c = quad_bcast<2>(r1);
  return quad_add_finish(a, b, c, Fq::add(p.z, p.z), role, Tout);
}
// Extended + Extended with both T = t1*t2 known: round 1 a, b, T1*T2, Z1*Z2; round 2 c = 2d*T1*T2 (lane 0 used; lane 1
// does a side product sa*sb for the caller, typically T of the NEXT operand); round 3 U, V, Z, T.  Three rounds
// instead of the four of quad_add_ext.
static JJ_DEV Ext quad_add_ext_t(const Ext& p, const Fe& Tp, const Ext& q, const Fe& Tq, u32 role, Fe& Tout,
                                 const Fe& sa, const Fe& sb, Fe& sout) {
  const Fe r1 = Fq::mul(role_select4(Fq::sub(p.v, p.u), Fq::add(p.v, p.u), Tp, p.z, role),
                        role_select4(Fq::sub(q.v, q.u), Fq::carry(Fq::add(q.v, q.u)), Tq, q.z, role));
  const Fe a = quad_bcast<0>(r1), b = quad_bcast<1>(r1), tt = quad_bcast<2>(r1), zz = quad_bcast<3>(r1);
  const Fe r2 = Fq::mul(role_select4(tt, sa, tt, tt, role), role_select4(Fq::konst(FqP::D2), sb, Fq::konst(FqP::D2), Fq::konst(FqP::D2), role));
  const Fe c = quad_bcast<0>(r2);
  sout = quad_bcast<1>(r2);
  return quad_add_finish(a, b, c, Fq::add(zz, zz), role, Tout);
}
// The same with TWO side products (lanes 1 and 2 of the second round, which only lane 0 needs): sout1 = sa1 * sb1, sout2 = sa2 * sb2.
static JJ_DEV Ext quad_add_ext_t2(const Ext& p, const Fe& Tp, const Ext& q, const Fe& Tq, u32 role, Fe& Tout,
                                  const Fe& sa1, const Fe& sb1, Fe& sout1, const Fe& sa2, const Fe& sb2, Fe& sout2) {
  const Fe r1 = Fq::mul(role_select4(Fq::sub(p.v, p.u), Fq::add(p.v, p.u), Tp, p.z, role),
                        role_select4(Fq::sub(q.v, q.u), Fq::carry(Fq::add(q.v, q.u)), Tq, q.z, role));
  const Fe a = quad_bcast<0>(r1), b = quad_bcast<1>(r1), tt = quad_bcast<2>(r1), zz = quad_bcast<3>(r1);
  const Fe r2 = Fq::mul(role_select4(tt, sa1, sa2, tt, role), role_select4(Fq::konst(FqP::D2), sb1, sb2, Fq::konst(FqP::D2), role));
  const Fe c = quad_bcast<0>(r2);
  sout1 = quad_bcast<1>(r2);
  sout2 = quad_bcast<2>(r2);
  return quad_add_finish(a, b, c, Fq::add(zz, zz), role, Tout);
}
// Small batches: one scalar multiplication per quad of lanes.  Same signed-window ladder and table as
// varbase_windowed, but every point operation is two multiplication rounds on four lanes (12 rounds per 5-bit window
// instead of 43 dependent products), which is what matters when the batch cannot fill the SIMDs anyway.
// All four lanes hold the same point; each lane stores / loads whole table entries itself.
static JJ_DEV Ext varbase_windowed_quad(const Affine& P, u32 (&k)[8], u32* slot, u32 role) {
  const ANiels pn = Curve::to_niels(P);
  Ext cur = Curve::from_affine(P);
  Fe T = Fq::mul(P.u, P.v);
  ENiels en;
  en.vpu = pn.vpu; en.vmu = pn.vmu; en.z2 = Fq::add(Fq::one(), Fq::one()); en.t2d = pn.t2d;
  store_eniels(slot, Curve::eniels_identity());
  store_eniels(slot + ENIELS_WORDS, en);
  #pragma unroll 1
  for (int j = 2; j <= VB_TABLE; j++) {
    cur = quad_add_aniels(cur, T, pn, role, T);
    en.vpu = Fq::carry(Fq::add(cur.v, cur.u)); en.vmu = Fq::sub(cur.v, cur.u); en.z2 = Fq::add(cur.z, cur.z);
    en.t2d = Fq::mul(T, Fq::konst(FqP::D2));
    store_eniels(slot + j * ENIELS_WORDS, en);
  }
  recode_signed(k);
  u32 a = vb_window(k, VB_NWIN - 1), neg = 0;
  ENiels e = load_eniels(slot + a * ENIELS_WORDS);
  Ext acc = Curve::identity();
  T = Fq::zero();
  #pragma unroll 1
  for (int i = VB_NWIN - 1; i >= 0; i--) {
    const ENiels s = e;
    const u32 sneg = neg;
    if (i > 0) {
      const int d = (int)vb_window(k, i - 1) - VB_TABLE;
      neg = d < 0; a = (u32)(d < 0 ? -d : d);
      e = load_eniels(slot + a * ENIELS_WORDS);
    }
    acc = quad_add_eniels(acc, T, s, sneg, role, T);
    if (i > 0) {
      #pragma unroll 1
      for (int d = 0; d < VB_W - 1; d++) acc = quad_dbl(acc, role);
      acc = quad_dbl_t(acc, role, T);
    }
  }
  return acc;
}
// The constant-time ladder for small batches: one scalar multiplication per QUAD of lanes, signed 3-bit windows as k_varbase_ct3.  In a quad
// addition lane r multiplies ONE coordinate of the table entry (lane 0: v-u or v+u, lane 1: the other one, lane 2: 2d T, lane 3: 2Z), so every lane
// keeps just that coordinate of {P, 2P, 3P, 4P} -- for an addition and for a subtraction: 8 x 9 registers, no table in memory or LDS -- and picks
// with bit masks.  84 windows x (3 doublings + 1 addition) x 2 multiplication rounds; no load, store, branch or address depends on the scalar.
static JJ_DEV Ext varbase_ct3_quad(const Affine& P, u32 (&k)[8], u32 role) {
  Fe pos[4], neg[4];                                           // this lane's coordinate of j P (j = 1 .. 4) for +j P / -j P
  auto keep = [&](int j, const Ext& e, const Fe& T) {
    const Fe vpu = Fq::carry(Fq::add(e.v, e.u)), vmu = Fq::sub(e.v, e.u), z2 = Fq::add(e.z, e.z), t2d = Fq::mul(T, Fq::konst(FqP::D2));
    pos[j] = role_select4(vmu, vpu, t2d, z2, role);
    neg[j] = role_select4(vpu, vmu, t2d, z2, role);
  };
  const Ext p1 = Curve::from_affine(P);
  const Fe T1 = Fq::mul(P.u, P.v);
  keep(0, p1, T1);
  ENiels e1;
  e1.vpu = Fq::carry(Fq::add(p1.v, p1.u)); e1.vmu = Fq::sub(p1.v, p1.u); e1.z2 = Fq::add(p1.z, p1.z); e1.t2d = Fq::mul(T1, Fq::konst(FqP::D2));
  Fe T2, T3, T4;
  const Ext p2 = quad_dbl_t(p1, role, T2);
  keep(1, p2, T2);
  const Ext p3 = quad_add_eniels(p2, T2, e1, 0u, role, T3);
  keep(2, p3, T3);
  const Ext p4 = quad_dbl_t(p2, role, T4);
  keep(3, p4, T4);
  // the identity entry as this lane sees it: 1, 1, 0, 2
  const Fe one = Fq::one();
  const Fe idn = role_select4(one, one, Fq::zero(), Fq::add(one, one), role);
  k[7] &= 0x0fffffffu;
  {
    constexpr u32 RC[8] = {0x24924924u, 0x49249249u, 0x92492492u, 0x24924924u, 0x49249249u, 0x92492492u, 0x24924924u, 0x09249249u};
    u64 cy = 0;
    _Pragma("unroll") for (int j = 0; j < 8; j++) { const u64 t = (u64)k[j] + RC[j] + cy; k[j] = (u32)t; cy = t >> 32; }
  }
  const u32 top = 0u - ((k[7] >> 28) & 1u);                    // bit 252: the carry window, 0 or 1 -> the ladder starts from O or P
  const Ext idp = Curve::identity();
  Ext acc;
  acc.u = Fq::select(idp.u, p1.u, top); acc.v = Fq::select(idp.v, p1.v, top); acc.z = Fq::select(idp.z, p1.z, top);
  acc.t1 = Fq::select(idp.t1, p1.t1, top); acc.t2 = Fq::select(idp.t2, p1.t2, top);
  Fe T = Fq::select(Fq::zero(), T1, top);
  u32 ks[8];
  _Pragma("unroll") for (int q = 7; q >= 1; q--) ks[q] = (k[q] << 4) | (k[q - 1] >> 28);
  ks[0] = k[0] << 4;
  #pragma unroll 1
  for (int i = CT3_NWIN - 1; i >= 0; i--) {
    const int d = (int)(ks[7] >> 29) - 4;
    _Pragma("unroll") for (int q = 7; q >= 1; q--) ks[q] = (ks[q] << 3) | (ks[q - 1] >> 29);
    ks[0] <<= 3;
    const u32 sgn = (u32)(d >> 31);
    const u32 a = ((u32)d ^ sgn) - sgn;
    u32 m4 = 0u - (a >> 2), m3 = 0u - ((a >> 1) & a & 1u), m2 = 0u - ((a >> 1) & ~a & 1u), m1 = 0u - (a & ~(a >> 1) & 1u);
    asm("" : "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4));
    const u32 m0 = ~(m1 | m2 | m3 | m4);
    acc = quad_dbl(acc, role);
    acc = quad_dbl(acc, role);
    acc = quad_dbl_t(acc, role, T);
    Fe op;
    _Pragma("unroll") for (int l = 0; l < NL; l++) {
      const u32 pp = and_or(pos[0].l[l], m1, and_or(pos[1].l[l], m2, and_or(pos[2].l[l], m3, and_or(pos[3].l[l], m4, idn.l[l] & m0))));
      const u32 nn = and_or(neg[0].l[l], m1, and_or(neg[1].l[l], m2, and_or(neg[2].l[l], m3, and_or(neg[3].l[l], m4, idn.l[l] & m0))));
      op.l[l] = (nn & sgn) | (pp & ~sgn);
    }
    // quad_add_eniels with this lane's operand already picked: round 1 a, b, c = T * 2dT', d = Z * 2Z'; round 2 U, V, Z, T
    const Fe r1 = Fq::mul(role_select4(Fq::sub(acc.v, acc.u), Fq::add(acc.v, acc.u), T, acc.z, role), op);
    const Fe ra = quad_bcast<0>(r1), rb = quad_bcast<1>(r1), rc = Fq::cneg(quad_bcast<2>(r1), sgn), rd = quad_bcast<3>(r1);
    acc = quad_add_finish(ra, rb, rc, rd, role, T);
  }
  return acc;
}
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_varbase_ct_quad(size_t n, const void* scalars, const void* points, SoA ext) {
  const size_t q = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const u32 role = threadIdx.x & 3u;
  if (q >= n) return;                                       // whole quads leave together
  u32 k[8];
  load8(k, scalars, q);
  const Affine P = load_affine(points, q);
  const Ext r = varbase_ct3_quad(P, k, role);
  if (role == 0) { ext.put(0, q, r.u); ext.put(1, q, r.v); ext.put(2, q, r.z); }
}
#endif  // JJ_KERNELS_BATCH
#ifdef JJ_KERNELS_BATCH
template <bool FIVE>
__global__ void __launch_bounds__(256) k_varbase_quad(size_t n, const void* scalars, const void* points, u32* tables, SoA ext) {
  const size_t q = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const u32 role = threadIdx.x & 3u;
  if (q >= n) return;                                       // whole quads leave together
  u32 k[8];
  load8(k, scalars, q);
  const Affine P = load_affine(points, q);
  const Ext r = varbase_windowed_quad(P, k, tables + q * (size_t)(VB_SLOTS * ENIELS_WORDS), role);
  if (role == 0) {
    ext.put(0, q, r.u); ext.put(1, q, r.v); ext.put(2, q, r.z);
    if constexpr (FIVE) { ext.put(3, q, Fq::carry(r.t1)); ext.put(4, q, Fq::carry(r.t2)); }
  }
}
#endif  // JJ_KERNELS_BATCH


#ifdef JJ_KERNELS_MSM
#include "jj_msm_kernels.h"
#endif

// ------------------------------------------------------------------------------------------------ synthetic inputs
// Counter-based generator of SURVEY 8(d): word j of unit i is splitmix64(seed + i * stride + j), so any index can be
// produced on any GPU or on the host (oracle/jubjub_ref.py synth_scalar / synth_point restate the same streams).
static JJ_DEV u64 splitmix64(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  u64 z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// scalar_i = four PRNG words, top 4 bits cleared, minus r if >= r (uniform-ish in [0, r)); raw != 0: the 32 PRNG bytes
// as they are (arbitrary bit patterns: inputs >= q, sign-bit noise for the decoder)
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_synth_scalars(size_t n, u64 seed, u64 first, int raw, void* out32) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[8];
  _Pragma("unroll") for (int j = 0; j < 4; j++) { const u64 x = splitmix64(seed + (first + i) * 4 + j); w[2 * j] = (u32)x; w[2 * j + 1] = (u32)(x >> 32); }
  if (!raw) {
    w[7] &= 0x0fffffffu;
    u32 d[8]; int64_t borrow = 0;
    _Pragma("unroll") for (int j = 0; j < 8; j++) { const int64_t t = (int64_t)w[j] - (int64_t)FR_MODULUS_W[j] + borrow; d[j] = (u32)t; borrow = t >> 32; }
    if (borrow == 0) { _Pragma("unroll") for (int j = 0; j < 8; j++) w[j] = d[j]; }
  }
  store8(out32, i, w);
}
#endif  // JJ_KERNELS_BATCH
// Group::random (reference src/lib.rs:1244-1267; SubgroupPoint::random 1290-1298 with `subgroup`): rejection sampling
//   loop { v = Fq::random (64 PRNG bytes, from_bytes_wide); flip = next_u32 % 2; u = sqrt((v^2-1)/(1+d v^2)) or retry;
//          p = (flip ? -u : u, v); retry if identity; [subgroup: p = [8]p; retry if identity] }
// The reference draws from one sequential RNG; here every unit owns the counter range [(first+i) << 16, ...) and attempt t
// reads words 16 t .. 16 t + 8 of it (8 for v, 1 for the flip), so the result is a pure function of (seed, index).
constexpr int RANDOM_POINT_STRIDE_LOG2 = 16;
#ifdef JJ_KERNELS_BATCH
__global__ void __launch_bounds__(256) k_random_points(size_t n, u64 seed, u64 first, int subgroup, SqrtTables tabs, void* out64, u32* attempts) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 base = seed + ((first + i) << RANDOM_POINT_STRIDE_LOG2);
  u32 t = 0;
  bool done = false;
  #pragma unroll 1
  while (!done) {                                              // divergent: lanes leave as their sample is accepted
    u32 lo[8], hi[8];
    _Pragma("unroll") for (int j = 0; j < 4; j++) {
      const u64 a = splitmix64(base + 16ull * t + j), b = splitmix64(base + 16ull * t + 4 + j);
      lo[2 * j] = (u32)a; lo[2 * j + 1] = (u32)(a >> 32); hi[2 * j] = (u32)b; hi[2 * j + 1] = (u32)(b >> 32);
    }
    const u32 flip = (u32)splitmix64(base + 16ull * t + 8) & 1u;
    t++;
    const Fe v = Fq::mul(Fq::from_words_wide(lo, hi), Fq::one());          // one carried representative of the 512-bit value mod q
    const Fe v2 = Fq::sqr(v);
    const Fe den = Fq::carry(Fq::add(Fq::one(), Fq::mul(Fq::konst(FqP::D), v2)));
    const Fe u2 = Fq::mul(Fq::sub(v2, Fq::one()), Fq::invert(den));       // invert(0) = 0, as unwrap_or(zero)
    bool ok;
    Fe u = fq_sqrt_fast(u2, ok, tabs);
    if (!ok) continue;
    u = Fq::cneg(u, flip ? ~0u : 0u);
    Affine a; a.u = Fq::mul(u, Fq::one()); a.v = v;
    Ext e = Curve::from_affine(a);
    if (Curve::is_identity(e)) continue;
    if (subgroup) {
      e = Curve::mul_by_cofactor(e);
      if (Curve::is_identity(e)) continue;
      const Fe zi = Fq::invert(e.z);
      store_affine(out64, i, Fq::mul(e.u, zi), Fq::mul(e.v, zi));
    } else {
      store_affine(out64, i, a.u, a.v);
    }
    done = true;
  }
  if (attempts) attempts[i] = t;
}
#endif  // JJ_KERNELS_BATCH

// ------------------------------------------------------------------------------------------------ roofline probe
// 8 independent v_mad_u64_u32 chains per lane, 64 mads per loop iteration: the measured peak IMAD32 rate.
#ifdef JJ_KERNELS_PROBE
__global__ void __launch_bounds__(256) k_peak_mad(u32* out, int iters, u32 seed) {
  u64 acc[8];
  const u32 a = seed * 2654435761u + threadIdx.x, b = (seed ^ (blockIdx.x * 40503u)) | 1u;
  _Pragma("unroll") for (int k = 0; k < 8; k++) acc[k] = (((u64)a << 32) | b) + k * 77u;
  for (int it = 0; it < iters; it++) {
    _Pragma("unroll") for (int r = 0; r < 8; r++) {
      _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
    }
  }
  u64 s = 0;
  _Pragma("unroll") for (int k = 0; k < 8; k++) s ^= acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}
#endif  // JJ_KERNELS_PROBE

}  // namespace jj
