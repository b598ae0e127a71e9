/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C CPU restatement of the zkcrypto/jubjub reference algorithms (crate jubjub 0.10.0,
 * /root/reference), using the reference's own data layout: field elements are 4 x u64
 * little-endian limbs in Montgomery form with R = 2^256 (src/fr.rs:19-23), multiply-accumulate
 * through unsigned __int128 exactly like src/util.rs:1-20.  Each function cites the reference
 * file:line it follows.  The same template serves Fq (= bls12_381::Scalar 0.8.0, not on disk,
 * Cargo.lock:50-53; same 4x64 Montgomery structure) and Fr (src/fr.rs).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * It is the checker and the reported CPU baseline ("kind":"port"), never the product.
 *
 * Parity pinning: validated against the reference's known-answer vectors through
 * tests/test_oracle_c.py (golden vectors in tests/golden/reference_vectors.json) and
 * cross-checked against the Python big-int oracle on random inputs.  Fq::sqrt root-sign is
 * "parity unpinned" (no reference test stores a raw sqrt output); decompression is pinned.
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC oracle/jubjub_oracle.c -o oracle/libjj_oracle.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;
typedef struct { u64 l[4]; } fe;            /* Montgomery form */

typedef struct {
  fe modulus, r, r2, r3;
  u64 inv;                                   /* -(p^-1) mod 2^64, src/fr.rs:213-214 */
} field_t;

/* ---- limb primitives: src/util.rs:1-20 ---- */
static inline u64 adc(u64 a, u64 b, u64 carry, u64 *out_carry) {
  u128 ret = (u128)a + (u128)b + (u128)carry;
  *out_carry = (u64)(ret >> 64);
  return (u64)ret;
}
static inline u64 sbb(u64 a, u64 b, u64 borrow, u64 *out_borrow) {
  u128 ret = (u128)a - ((u128)b + (u128)(borrow >> 63));
  *out_borrow = (u64)(ret >> 64);
  return (u64)ret;
}
static inline u64 mac(u64 a, u64 b, u64 c, u64 carry, u64 *out_carry) {
  u128 ret = (u128)a + (u128)b * (u128)c + (u128)carry;
  *out_carry = (u64)(ret >> 64);
  return (u64)ret;
}

/* ---- constants ---- */
static const field_t FQ = {
  {{0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL}},
  {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}},
  {{0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL}},
  {{0xc62c1807439b73afULL, 0x1b3e0d188cf06990ULL, 0x73d13c71c7b5f418ULL, 0x6e2a5bb9c8db33e9ULL}},
  0xfffffffeffffffffULL};
/* src/fr.rs:77-82, 217-238, 214 */
static const field_t FR = {
  {{0xd0970e5ed6f72cb7ULL, 0xa6682093ccc81082ULL, 0x06673b0101343b00ULL, 0x0e7db4ea6533afa9ULL}},
  {{0x25f80bb3b99607d9ULL, 0xf315d62f66b6e750ULL, 0x932514eeeb8814f4ULL, 0x09a6fc6f479155c6ULL}},
  {{0x67719aa495e57731ULL, 0x51b0cef09ce3fc26ULL, 0x69dab7fac026e9a5ULL, 0x04f6547b8d127688ULL}},
  {{0xe0d6c6563d830544ULL, 0x323e3883598d0f85ULL, 0xf0fea3004c2e2ba8ULL, 0x05874f84946737ecULL}},
  0x1ba3a358ef788ef9ULL};

/* d and 2d in Montgomery form (canonical values at src/lib.rs:399-412; converted by jjo_selftest) */
static const fe FQ_D  = {{0x2a522455b974f6b0ULL, 0xfc6cc9ef0d9acab3ULL, 0x7a08fb94c27628d1ULL, 0x57f8f6a8fe0e262eULL}};
static const fe FQ_D2 = {{0x54a448ac72e9ed5fULL, 0xa51befdb1b373967ULL, 0xc0d81f217b4a799eULL, 0x3c0445fed27ecf14ULL}};
/* Fq 2^32-th root of unity 7^t, Montgomery form */
static const fe FQ_ROOT = {{0xb9b58d8c5f0e466aULL, 0x5b1b4c801819d7ecULL, 0x0af53ae352a31e64ULL, 0x5bf3adda19e9b27bULL}};
/* (t-1)/2 for Fq, plain integer */
static const u64 FQ_TM1D2[4] = {0x7fff2dff7fffffffULL, 0x04d0ec02a9ded201ULL, 0x94cebea4199cec04ULL, 0x0000000039f6d3a9ULL};
/* src/lib.rs:73-76 */
static const uint8_t FR_MODULUS_BYTES[32] = {183, 44, 247, 214, 94, 14, 151, 208, 130, 16, 200, 204, 147, 32, 104, 166,
                                             0, 59, 52, 1, 1, 59, 103, 6, 169, 175, 51, 101, 234, 180, 125, 14};

/* ---- field ops: src/fr.rs ---- */
/* src/fr.rs:620-634 */
static inline fe f_sub(const field_t *F, const fe *a, const fe *b) {
  u64 borrow = 0, carry = 0; fe d;
  d.l[0] = sbb(a->l[0], b->l[0], 0, &borrow);
  d.l[1] = sbb(a->l[1], b->l[1], borrow, &borrow);
  d.l[2] = sbb(a->l[2], b->l[2], borrow, &borrow);
  d.l[3] = sbb(a->l[3], b->l[3], borrow, &borrow);
  d.l[0] = adc(d.l[0], F->modulus.l[0] & borrow, 0, &carry);
  d.l[1] = adc(d.l[1], F->modulus.l[1] & borrow, carry, &carry);
  d.l[2] = adc(d.l[2], F->modulus.l[2] & borrow, carry, &carry);
  d.l[3] = adc(d.l[3], F->modulus.l[3] & borrow, carry, &carry);
  return d;
}
/* src/fr.rs:638-647 */
static inline fe f_add(const field_t *F, const fe *a, const fe *b) {
  u64 carry = 0; fe d;
  d.l[0] = adc(a->l[0], b->l[0], 0, &carry);
  d.l[1] = adc(a->l[1], b->l[1], carry, &carry);
  d.l[2] = adc(a->l[2], b->l[2], carry, &carry);
  d.l[3] = adc(a->l[3], b->l[3], carry, &carry);
  return f_sub(F, &d, &F->modulus);
}
/* src/fr.rs:651-665 */
static inline fe f_neg(const field_t *F, const fe *a) {
  u64 borrow = 0; fe d;
  d.l[0] = sbb(F->modulus.l[0], a->l[0], 0, &borrow);
  d.l[1] = sbb(F->modulus.l[1], a->l[1], borrow, &borrow);
  d.l[2] = sbb(F->modulus.l[2], a->l[2], borrow, &borrow);
  d.l[3] = sbb(F->modulus.l[3], a->l[3], borrow, &borrow);
  u64 mask = (u64)((a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0) - 1;
  d.l[0] &= mask; d.l[1] &= mask; d.l[2] &= mask; d.l[3] &= mask;
  return d;
}
static inline fe f_double(const field_t *F, const fe *a) { return f_add(F, a, a); } /* src/fr.rs:261-263 */

/* src/fr.rs:544-588 */
static inline fe f_montgomery_reduce(const field_t *F, u64 r0, u64 r1, u64 r2, u64 r3, u64 r4, u64 r5, u64 r6, u64 r7) {
  const u64 *m = F->modulus.l; u64 carry, carry2, k;
  k = r0 * F->inv;
  (void)mac(r0, k, m[0], 0, &carry);
  r1 = mac(r1, k, m[1], carry, &carry);
  r2 = mac(r2, k, m[2], carry, &carry);
  r3 = mac(r3, k, m[3], carry, &carry);
  r4 = adc(r4, 0, carry, &carry2);
  k = r1 * F->inv;
  (void)mac(r1, k, m[0], 0, &carry);
  r2 = mac(r2, k, m[1], carry, &carry);
  r3 = mac(r3, k, m[2], carry, &carry);
  r4 = mac(r4, k, m[3], carry, &carry);
  r5 = adc(r5, carry2, carry, &carry2);
  k = r2 * F->inv;
  (void)mac(r2, k, m[0], 0, &carry);
  r3 = mac(r3, k, m[1], carry, &carry);
  r4 = mac(r4, k, m[2], carry, &carry);
  r5 = mac(r5, k, m[3], carry, &carry);
  r6 = adc(r6, carry2, carry, &carry2);
  k = r3 * F->inv;
  (void)mac(r3, k, m[0], 0, &carry);
  r4 = mac(r4, k, m[1], carry, &carry);
  r5 = mac(r5, k, m[2], carry, &carry);
  r6 = mac(r6, k, m[3], carry, &carry);
  r7 = adc(r7, carry2, carry, &carry2);
  fe t = {{r4, r5, r6, r7}};
  return f_sub(F, &t, &F->modulus);
}
/* src/fr.rs:592-616 */
static inline fe f_mul(const field_t *F, const fe *a, const fe *b) {
  u64 carry, r0, r1, r2, r3, r4, r5, r6, r7;
  r0 = mac(0, a->l[0], b->l[0], 0, &carry);
  r1 = mac(0, a->l[0], b->l[1], carry, &carry);
  r2 = mac(0, a->l[0], b->l[2], carry, &carry);
  r3 = mac(0, a->l[0], b->l[3], carry, &r4);
  r1 = mac(r1, a->l[1], b->l[0], 0, &carry);
  r2 = mac(r2, a->l[1], b->l[1], carry, &carry);
  r3 = mac(r3, a->l[1], b->l[2], carry, &carry);
  r4 = mac(r4, a->l[1], b->l[3], carry, &r5);
  r2 = mac(r2, a->l[2], b->l[0], 0, &carry);
  r3 = mac(r3, a->l[2], b->l[1], carry, &carry);
  r4 = mac(r4, a->l[2], b->l[2], carry, &carry);
  r5 = mac(r5, a->l[2], b->l[3], carry, &r6);
  r3 = mac(r3, a->l[3], b->l[0], 0, &carry);
  r4 = mac(r4, a->l[3], b->l[1], carry, &carry);
  r5 = mac(r5, a->l[3], b->l[2], carry, &carry);
  r6 = mac(r6, a->l[3], b->l[3], carry, &r7);
  return f_montgomery_reduce(F, r0, r1, r2, r3, r4, r5, r6, r7);
}
/* src/fr.rs:353-381 */
static inline fe f_square(const field_t *F, const fe *a) {
  u64 carry, r0, r1, r2, r3, r4, r5, r6, r7;
  r1 = mac(0, a->l[0], a->l[1], 0, &carry);
  r2 = mac(0, a->l[0], a->l[2], carry, &carry);
  r3 = mac(0, a->l[0], a->l[3], carry, &r4);
  r3 = mac(r3, a->l[1], a->l[2], 0, &carry);
  r4 = mac(r4, a->l[1], a->l[3], carry, &r5);
  r5 = mac(r5, a->l[2], a->l[3], 0, &r6);
  r7 = r6 >> 63;
  r6 = (r6 << 1) | (r5 >> 63);
  r5 = (r5 << 1) | (r4 >> 63);
  r4 = (r4 << 1) | (r3 >> 63);
  r3 = (r3 << 1) | (r2 >> 63);
  r2 = (r2 << 1) | (r1 >> 63);
  r1 = r1 << 1;
  r0 = mac(0, a->l[0], a->l[0], 0, &carry);
  r1 = adc(0, r1, carry, &carry);
  r2 = mac(r2, a->l[1], a->l[1], carry, &carry);
  r3 = adc(0, r3, carry, &carry);
  r4 = mac(r4, a->l[2], a->l[2], carry, &carry);
  r5 = adc(0, r5, carry, &carry);
  r6 = mac(r6, a->l[3], a->l[3], carry, &carry);
  r7 = adc(0, r7, carry, &carry);
  return f_montgomery_reduce(F, r0, r1, r2, r3, r4, r5, r6, r7);
}
static inline int f_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int f_eq(const fe *a, const fe *b) {               /* src/fr.rs:48-55 */
  return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline fe f_select(const fe *a, const fe *b, int choice) { /* src/fr.rs:64-73: choice ? b : a */
  u64 m = (u64)0 - (u64)(choice & 1); fe r;
  for (int i = 0; i < 4; i++) r.l[i] = a->l[i] ^ (m & (a->l[i] ^ b->l[i]));
  return r;
}
/* src/fr.rs:422-434 */
static fe f_pow_vartime(const field_t *F, const fe *a, const u64 by[4]) {
  fe res = F->r;
  for (int e = 3; e >= 0; e--)
    for (int i = 63; i >= 0; i--) {
      res = f_square(F, &res);
      if ((by[e] >> i) & 1) res = f_mul(F, &res, a);
    }
  return res;
}
/* invert = a^(p-2); src/fr.rs:438-540 (Fr uses an addition chain with the same value); ok = a != 0 */
static fe f_invert(const field_t *F, const fe *a, int *ok) {
  u64 e[4]; u64 borrow = 0;
  e[0] = sbb(F->modulus.l[0], 2, 0, &borrow);
  e[1] = sbb(F->modulus.l[1], 0, borrow, &borrow);
  e[2] = sbb(F->modulus.l[2], 0, borrow, &borrow);
  e[3] = sbb(F->modulus.l[3], 0, borrow, &borrow);
  *ok = !f_is_zero(a);
  return f_pow_vartime(F, a, e);
}
/* src/fr.rs:268-292 */
static fe f_from_bytes(const field_t *F, const uint8_t b[32], int *ok) {
  fe t; u64 borrow = 0;
  memcpy(t.l, b, 32);                         /* little-endian host */
  (void)sbb(t.l[0], F->modulus.l[0], 0, &borrow);
  (void)sbb(t.l[1], F->modulus.l[1], borrow, &borrow);
  (void)sbb(t.l[2], F->modulus.l[2], borrow, &borrow);
  (void)sbb(t.l[3], F->modulus.l[3], borrow, &borrow);
  *ok = (int)(borrow & 1);
  return f_mul(F, &t, &F->r2);
}
/* src/fr.rs:296-308 */
static void f_to_bytes(const field_t *F, const fe *a, uint8_t out[32]) {
  fe t = f_montgomery_reduce(F, a->l[0], a->l[1], a->l[2], a->l[3], 0, 0, 0, 0);
  memcpy(out, t.l, 32);
}
/* src/fr.rs:312-343 */
static fe f_from_bytes_wide(const field_t *F, const uint8_t b[64]) {
  fe d0, d1; memcpy(d0.l, b, 32); memcpy(d1.l, b + 32, 32);
  fe x = f_mul(F, &d0, &F->r2), y = f_mul(F, &d1, &F->r3);
  return f_add(F, &x, &y);
}
/* src/fr.rs:347-349 */
static fe f_from_raw(const field_t *F, const u64 v[4]) { fe t; memcpy(t.l, v, 32); return f_mul(F, &t, &F->r2); }

/* src/fr.rs:384-399 */
static fe fr_sqrt(const fe *a, int *ok) {
  static const u64 e[4] = {0xb425c397b5bdcb2eULL, 0x299a0824f3320420ULL, 0x4199cec0404d0ec0ULL, 0x039f6d3a994cebeaULL};
  fe s = f_pow_vartime(&FR, a, e), ss = f_square(&FR, &s);
  *ok = f_eq(&ss, a);
  return s;
}
/* Fq::sqrt = ff::helpers::sqrt_tonelli_shanks(self, (t-1)/2)  (bls12_381 0.8.0 / ff 0.13.1; call sites
 * src/lib.rs:515,610,1253). S = 32. */
static fe fq_sqrt(const fe *a, int *ok) {
  const field_t *F = &FQ;
  fe w = f_pow_vartime(F, a, FQ_TM1D2);
  int v = 32;
  fe x = f_mul(F, a, &w), b = f_mul(F, &x, &w), z = FQ_ROOT;
  for (int max_v = 32; max_v >= 1; max_v--) {
    int k = 1;
    fe tmp = f_square(F, &b);
    int j_less_than_v = 1;
    for (int j = 2; j < max_v; j++) {
      int tmp_is_one = f_eq(&tmp, &F->r);
      fe sel = f_select(&tmp, &z, tmp_is_one);
      fe squared = f_square(F, &sel);
      tmp = f_select(&squared, &tmp, tmp_is_one);
      fe new_z = f_select(&z, &squared, tmp_is_one);
      j_less_than_v &= (j != v);
      k = tmp_is_one ? k : j;
      z = f_select(&z, &new_z, j_less_than_v);
    }
    fe result = f_mul(F, &x, &z);
    x = f_select(&result, &x, f_eq(&b, &F->r));
    z = f_square(F, &z);
    b = f_mul(F, &b, &z);
    v = k;
  }
  fe xx = f_square(F, &x);
  *ok = f_eq(&xx, a);
  return x;
}

/* ---- points (all coordinates Fq, Montgomery) ---- */
typedef struct { fe u, v; } affine_t;                 /* src/lib.rs:80-84   */
typedef struct { fe u, v, z, t1, t2; } ext_t;         /* src/lib.rs:138-145 */
typedef struct { fe vpu, vmu, t2d; } aniels_t;        /* src/lib.rs:254-259 */
typedef struct { fe vpu, vmu, z, t2d; } eniels_t;     /* src/lib.rs:326-332 */
typedef struct { fe u, v, z, t; } completed_t;        /* src/lib.rs:1036-1041 */

#define Q (&FQ)
static inline fe qmul(const fe *a, const fe *b) { return f_mul(Q, a, b); }
static inline fe qsq(const fe *a) { return f_square(Q, a); }
static inline fe qadd(const fe *a, const fe *b) { return f_add(Q, a, b); }
static inline fe qsub(const fe *a, const fe *b) { return f_sub(Q, a, b); }

static ext_t ext_identity(void) { ext_t p; memset(&p, 0, sizeof p); p.v = FQ.r; p.z = FQ.r; return p; } /* lib.rs:680-688 */
static aniels_t aniels_identity(void) { aniels_t n; n.vpu = FQ.r; n.vmu = FQ.r; memset(&n.t2d, 0, 32); return n; } /* 263-269 */
static eniels_t eniels_identity(void) { eniels_t n; n.vpu = FQ.r; n.vmu = FQ.r; n.z = FQ.r; memset(&n.t2d, 0, 32); return n; } /* 347-354 */
static ext_t affine_to_ext(const affine_t *a) { ext_t p = {a->u, a->v, FQ.r, a->u, a->v}; return p; } /* lib.rs:640-648 */
/* src/lib.rs:1052-1060 */
static inline ext_t into_extended(const completed_t *c) {
  ext_t p; p.u = qmul(&c->u, &c->t); p.v = qmul(&c->v, &c->z); p.z = qmul(&c->z, &c->t); p.t1 = c->u; p.t2 = c->v; return p;
}
/* src/lib.rs:739-828 */
static ext_t ext_double(const ext_t *p) {
  fe uu = qsq(&p->u), vv = qsq(&p->v), zz = qsq(&p->z), zz2 = f_double(Q, &zz);
  fe upv = qadd(&p->u, &p->v), uv2 = qsq(&upv);
  fe vpu = qadd(&vv, &uu), vmu = qsub(&vv, &uu);
  completed_t c; c.u = qsub(&uv2, &vpu); c.v = vpu; c.z = vmu; c.t = qsub(&zz2, &vmu);
  return into_extended(&c);
}
/* src/lib.rs:883-920 */
static ext_t ext_add_eniels(const ext_t *p, const eniels_t *n) {
  fe vmu = qsub(&p->v, &p->u), vpu = qadd(&p->v, &p->u);
  fe a = qmul(&vmu, &n->vmu), b = qmul(&vpu, &n->vpu);
  fe tt = qmul(&p->t1, &p->t2), c = qmul(&tt, &n->t2d);
  fe zz = qmul(&p->z, &n->z), d = f_double(Q, &zz);
  completed_t r; r.u = qsub(&b, &a); r.v = qadd(&b, &a); r.z = qadd(&d, &c); r.t = qsub(&d, &c);
  return into_extended(&r);
}
/* src/lib.rs:944-968 */
static ext_t ext_add_aniels(const ext_t *p, const aniels_t *n) {
  fe vmu = qsub(&p->v, &p->u), vpu = qadd(&p->v, &p->u);
  fe a = qmul(&vmu, &n->vmu), b = qmul(&vpu, &n->vpu);
  fe tt = qmul(&p->t1, &p->t2), c = qmul(&tt, &n->t2d);
  fe d = f_double(Q, &p->z);
  completed_t r; r.u = qsub(&b, &a); r.v = qadd(&b, &a); r.z = qadd(&d, &c); r.t = qsub(&d, &c);
  return into_extended(&r);
}
/* src/lib.rs:652-658 */
static aniels_t affine_to_niels(const affine_t *a) {
  aniels_t n; n.vpu = qadd(&a->v, &a->u); n.vmu = qsub(&a->v, &a->u);
  fe uv = qmul(&a->u, &a->v); n.t2d = qmul(&uv, &FQ_D2); return n;
}
/* src/lib.rs:728-735 */
static eniels_t ext_to_niels(const ext_t *p) {
  eniels_t n; n.vpu = qadd(&p->v, &p->u); n.vmu = qsub(&p->v, &p->u); n.z = p->z;
  fe tt = qmul(&p->t1, &p->t2); n.t2d = qmul(&tt, &FQ_D2); return n;
}
static inline int ladder_bit(const uint8_t by[32], int i) { return (by[i >> 3] >> (i & 7)) & 1; }
/* src/lib.rs:357-379: bits 251..0, double then add select(identity, self, bit) */
static ext_t eniels_multiply(const eniels_t *n, const uint8_t by[32]) {
  eniels_t zero = eniels_identity(); ext_t acc = ext_identity();
  for (int i = 251; i >= 0; i--) {
    acc = ext_double(&acc);
    int bit = ladder_bit(by, i);
    eniels_t s;
    s.vpu = f_select(&zero.vpu, &n->vpu, bit); s.vmu = f_select(&zero.vmu, &n->vmu, bit);
    s.z = f_select(&zero.z, &n->z, bit); s.t2d = f_select(&zero.t2d, &n->t2d, bit);
    acc = ext_add_eniels(&acc, &s);
  }
  return acc;
}
/* src/lib.rs:272-295 */
static ext_t aniels_multiply(const aniels_t *n, const uint8_t by[32]) {
  aniels_t zero = aniels_identity(); ext_t acc = ext_identity();
  for (int i = 251; i >= 0; i--) {
    acc = ext_double(&acc);
    int bit = ladder_bit(by, i);
    aniels_t s;
    s.vpu = f_select(&zero.vpu, &n->vpu, bit); s.vmu = f_select(&zero.vmu, &n->vmu, bit);
    s.t2d = f_select(&zero.t2d, &n->t2d, bit);
    acc = ext_add_aniels(&acc, &s);
  }
  return acc;
}
static ext_t ext_multiply(const ext_t *p, const uint8_t by[32]) { eniels_t n = ext_to_niels(p); return eniels_multiply(&n, by); } /* 831-833 */
static int ext_is_identity(const ext_t *p) { return f_is_zero(&p->u) & f_eq(&p->v, &p->z); }                    /* 691-696 */
static int ext_is_small_order(const ext_t *p) { ext_t a = ext_double(p), b = ext_double(&a); return f_is_zero(&b.u); } /* 699-705 */
static int ext_is_torsion_free(const ext_t *p) { ext_t m = ext_multiply(p, FR_MODULUS_BYTES); return ext_is_identity(&m); } /* 709-711 */
static ext_t ext_mul_by_cofactor(const ext_t *p) { ext_t a = ext_double(p), b = ext_double(&a); return ext_double(&b); } /* 722-724 */
/* src/lib.rs:227-243 */
static affine_t ext_to_affine(const ext_t *p) {
  int ok; fe zinv = f_invert(Q, &p->z, &ok); affine_t a; a.u = qmul(&p->u, &zinv); a.v = qmul(&p->v, &zinv); return a;
}
/* src/lib.rs:455-464 */
static void affine_to_bytes(const affine_t *a, uint8_t out[32]) {
  uint8_t ub[32]; f_to_bytes(Q, &a->v, out); f_to_bytes(Q, &a->u, ub); out[31] |= (uint8_t)(ub[0] << 7);
}
/* src/lib.rs:492-534; `den_inv` lets batch_from_bytes share one inversion (lib.rs:596-600) */
static int affine_from_bytes_core(const uint8_t in[32], int zip216, const fe *den_inv, affine_t *out) {
  uint8_t b[32]; memcpy(b, in, 32);
  int sign = b[31] >> 7; b[31] &= 0x7f;
  int ok; fe v = f_from_bytes(Q, b, &ok);
  memset(out, 0, sizeof *out);
  if (!ok) return 0;
  fe v2 = qsq(&v), num = qsub(&v2, &FQ.r), inv;
  if (den_inv) inv = *den_inv;
  else { fe dv2 = qmul(&FQ_D, &v2), den = qadd(&FQ.r, &dv2); int iok; inv = f_invert(Q, &den, &iok); if (!iok) memset(&inv, 0, 32); }
  fe u2 = qmul(&num, &inv); int sok; fe u = fq_sqrt(&u2, &sok);
  if (!sok) return 0;
  uint8_t ub[32]; f_to_bytes(Q, &u, ub);
  int flip = (ub[0] ^ sign) & 1;
  fe un = f_neg(Q, &u), fu = f_select(&u, &un, flip);
  if (zip216 && f_is_zero(&u) && flip) return 0;
  out->u = fu; out->v = v; return 1;
}

/* ================= exported batch API (canonical little-endian bytes on the wire) ================= */
#define API __attribute__((visibility("default")))

static affine_t load_affine(const uint8_t *p) { /* from_raw_unchecked semantics: reduce mod q like from_raw */
  affine_t a; u64 t[4]; memcpy(t, p, 32); a.u = f_from_raw(Q, t); memcpy(t, p + 32, 32); a.v = f_from_raw(Q, t); return a;
}
static void store_affine(const affine_t *a, uint8_t *p) { f_to_bytes(Q, &a->u, p); f_to_bytes(Q, &a->v, p + 32); }

/* which: 0 = Fq, 1 = Fr.  op: 0 add 1 sub 2 mul 3 neg 4 square 5 double 6 invert 7 sqrt.  Elements 32B canonical LE. */
API int jjo_field_op(int which, int op, size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out, uint8_t *ok) {
  const field_t *F = which ? &FR : &FQ;
  #pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    u64 t[4]; memcpy(t, a + 32 * i, 32); fe x = f_from_raw(F, t), y = x, r; int good = 1;
    if (b) { memcpy(t, b + 32 * i, 32); y = f_from_raw(F, t); }
    switch (op) {
      case 0: r = f_add(F, &x, &y); break;
      case 1: r = f_sub(F, &x, &y); break;
      case 2: r = f_mul(F, &x, &y); break;
      case 3: r = f_neg(F, &x); break;
      case 4: r = f_square(F, &x); break;
      case 5: r = f_double(F, &x); break;
      case 6: r = f_invert(F, &x, &good); if (!good) memset(&r, 0, 32); break;
      case 7: r = which ? fr_sqrt(&x, &good) : fq_sqrt(&x, &good); if (!good) memset(&r, 0, 32); break;
      default: memset(&r, 0, 32); good = 0;
    }
    f_to_bytes(F, &r, out + 32 * i);
    if (ok) ok[i] = (uint8_t)good;
  }
  return 0;
}
/* Montgomery-limb level access so tests can check the reference's golden limbs (fr.rs / lib.rs:1758-1776) */
API void jjo_mont_mul(int which, const u64 a[4], const u64 b[4], u64 out[4]) {
  fe x, y; memcpy(x.l, a, 32); memcpy(y.l, b, 32); fe r = f_mul(which ? &FR : &FQ, &x, &y); memcpy(out, r.l, 32);
}
API void jjo_to_mont(int which, const uint8_t in[32], u64 out[4]) { u64 t[4]; memcpy(t, in, 32); fe r = f_from_raw(which ? &FR : &FQ, t); memcpy(out, r.l, 32); }
API void jjo_from_mont(int which, const u64 in[4], uint8_t out[32]) { fe x; memcpy(x.l, in, 32); f_to_bytes(which ? &FR : &FQ, &x, out); }
API int jjo_from_bytes(int which, size_t n, const uint8_t *in, uint8_t *out, uint8_t *ok) {
  const field_t *F = which ? &FR : &FQ;
  for (size_t i = 0; i < n; i++) { int good; fe x = f_from_bytes(F, in + 32 * i, &good); if (!good) memset(&x, 0, 32); f_to_bytes(F, &x, out + 32 * i); ok[i] = (uint8_t)good; }
  return 0;
}
API int jjo_from_bytes_wide(int which, size_t n, const uint8_t *in64, uint8_t *out) {
  const field_t *F = which ? &FR : &FQ;
  for (size_t i = 0; i < n; i++) { fe x = f_from_bytes_wide(F, in64 + 64 * i); f_to_bytes(F, &x, out + 32 * i); }
  return 0;
}

/* ExtendedPoint * scalar-bytes through the exact 252-step ladder, then to_affine (lib.rs:873-879, 831-833, 227-243) */
API int jjo_varbase_mul(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t *out_affine) {
  #pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    affine_t a = load_affine(points + 64 * i); ext_t p = affine_to_ext(&a);
    ext_t r = ext_multiply(&p, scalars + 32 * i); affine_t o = ext_to_affine(&r); store_affine(&o, out_affine + 64 * i);
  }
  return 0;
}
/* AffinePoint * scalar via AffineNielsPoint::multiply (lib.rs:1109-1115, 272-295); one shared base point */
API int jjo_fixedbase_mul(size_t n, const uint8_t *scalars, const uint8_t *base_affine, uint8_t *out_affine) {
  affine_t b = load_affine(base_affine); aniels_t nb = affine_to_niels(&b);
  #pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    ext_t r = aniels_multiply(&nb, scalars + 32 * i); affine_t o = ext_to_affine(&r); store_affine(&o, out_affine + 64 * i);
  }
  return 0;
}
/* Full projective result of the exact ladder, 5 x 32B canonical (for coordinate-exact checks) */
API int jjo_varbase_mul_ext(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t *out160) {
  for (size_t i = 0; i < n; i++) {
    affine_t a = load_affine(points + 64 * i); ext_t p = affine_to_ext(&a); ext_t r = ext_multiply(&p, scalars + 32 * i);
    f_to_bytes(Q, &r.u, out160 + 160 * i); f_to_bytes(Q, &r.v, out160 + 160 * i + 32); f_to_bytes(Q, &r.z, out160 + 160 * i + 64);
    f_to_bytes(Q, &r.t1, out160 + 160 * i + 96); f_to_bytes(Q, &r.t2, out160 + 160 * i + 128);
  }
  return 0;
}
/* op: 0 double, 1 add (Ext+Affine, lib.rs:1012-1019), 2 sub, 3 neg, 4 mul_by_cofactor; affine in/out */
API int jjo_point_op(int op, size_t n, const uint8_t *pa, const uint8_t *pb, uint8_t *out_affine) {
  #pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    affine_t a = load_affine(pa + 64 * i); ext_t p = affine_to_ext(&a), r;
    if (op == 0) r = ext_double(&p);
    else if (op == 1 || op == 2) {
      affine_t b = load_affine(pb + 64 * i); if (op == 2) b.u = f_neg(Q, &b.u);   /* P - B = P + (-B), lib.rs:92-104 */
      aniels_t nb = affine_to_niels(&b); r = ext_add_aniels(&p, &nb);
    } else if (op == 3) { r = p; r.u = f_neg(Q, &p.u); r.t1 = f_neg(Q, &p.t1); }
    else r = ext_mul_by_cofactor(&p);
    affine_t o = ext_to_affine(&r); store_affine(&o, out_affine + 64 * i);
  }
  return 0;
}
API int jjo_to_niels(size_t n, const uint8_t *pa, uint8_t *out96) {
  for (size_t i = 0; i < n; i++) { affine_t a = load_affine(pa + 64 * i); aniels_t t = affine_to_niels(&a);
    f_to_bytes(Q, &t.vpu, out96 + 96 * i); f_to_bytes(Q, &t.vmu, out96 + 96 * i + 32); f_to_bytes(Q, &t.t2d, out96 + 96 * i + 64); }
  return 0;
}
/* what: 0 is_identity 1 is_small_order 2 is_torsion_free 3 is_prime_order 4 is_on_curve */
API int jjo_predicate(int what, size_t n, const uint8_t *pa, uint8_t *out) {
  #pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    affine_t a = load_affine(pa + 64 * i); ext_t p = affine_to_ext(&a); int r = 0;
    if (what == 0) r = ext_is_identity(&p);
    else if (what == 1) r = ext_is_small_order(&p);
    else if (what == 2) r = ext_is_torsion_free(&p);
    else if (what == 3) r = ext_is_torsion_free(&p) & !ext_is_identity(&p);
    else { fe u2 = qsq(&a.u), v2 = qsq(&a.v), l = qsub(&v2, &u2), uv = qmul(&u2, &v2), duv = qmul(&FQ_D, &uv), rr = qadd(&FQ.r, &duv); r = f_eq(&l, &rr); }
    out[i] = (uint8_t)r;
  }
  return 0;
}
API int jjo_compress(size_t n, const uint8_t *pa, uint8_t *out32) {
  for (size_t i = 0; i < n; i++) { affine_t a = load_affine(pa + 64 * i); affine_to_bytes(&a, out32 + 32 * i); }
  return 0;
}
/* flags: bit0 zip216, bit1 require torsion-free, bit2 reject small order, bit3 clear cofactor (multiply output by 8) */
API int jjo_decompress(size_t n, const uint8_t *in32, int flags, uint8_t *out_affine, uint8_t *ok) {
  #pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    affine_t a; int good = affine_from_bytes_core(in32 + 32 * i, flags & 1, NULL, &a);
    if (good) {
      ext_t p = affine_to_ext(&a);
      if ((flags & 2) && !ext_is_torsion_free(&p)) good = 0;
      if ((flags & 4) && ext_is_small_order(&p)) good = 0;
      if (good && (flags & 8)) { ext_t c = ext_mul_by_cofactor(&p); a = ext_to_affine(&c); }
    }
    if (!good) memset(&a, 0, sizeof a);
    if (good) store_affine(&a, out_affine + 64 * i); else memset(out_affine + 64 * i, 0, 64);
    ok[i] = (uint8_t)good;
  }
  return 0;
}
/* batch_from_bytes with one shared inversion (lib.rs:541-627) — single-threaded, exact structure */
API int jjo_batch_from_bytes(size_t n, const uint8_t *in32, uint8_t *out_affine, uint8_t *ok, uint8_t *scratch /* n*64 bytes */) {
  fe *den = (fe *)scratch, *pre = den + n; fe acc = FQ.r;
  for (size_t i = 0; i < n; i++) {
    uint8_t b[32]; memcpy(b, in32 + 32 * i, 32); b[31] &= 0x7f; int good; fe v = f_from_bytes(Q, b, &good);
    if (good) { fe v2 = qsq(&v), dv2 = qmul(&FQ_D, &v2); den[i] = qadd(&FQ.r, &dv2); } else memset(&den[i], 0, 32);
    pre[i] = acc; if (!f_is_zero(&den[i])) acc = qmul(&acc, &den[i]);       /* zero elements are skipped (ff BatchInvert) */
  }
  int iok; fe inv = f_invert(Q, &acc, &iok);
  for (size_t i = n; i-- > 0;) {
    if (f_is_zero(&den[i])) continue;
    fe di = qmul(&inv, &pre[i]); inv = qmul(&inv, &den[i]); den[i] = di;
  }
  for (size_t i = 0; i < n; i++) {
    affine_t a; int good = affine_from_bytes_core(in32 + 32 * i, 1, &den[i], &a);
    if (good) store_affine(&a, out_affine + 64 * i); else memset(out_affine + 64 * i, 0, 64);
    ok[i] = (uint8_t)good;
  }
  return 0;
}
/* batch_normalize (lib.rs:1084-1107): n x 160B extended canonical -> n x 64B affine; one inversion, zeros skipped */
API int jjo_batch_normalize(size_t n, const uint8_t *ext160, uint8_t *out_affine, uint8_t *scratch /* n*64 bytes */) {
  fe *z = (fe *)scratch, *pre = z + n; fe acc = FQ.r; u64 t[4];
  for (size_t i = 0; i < n; i++) { memcpy(t, ext160 + 160 * i + 64, 32); z[i] = f_from_raw(Q, t); pre[i] = acc; if (!f_is_zero(&z[i])) acc = qmul(&acc, &z[i]); }
  int iok; fe inv = f_invert(Q, &acc, &iok);
  for (size_t i = n; i-- > 0;) { if (f_is_zero(&z[i])) continue; fe zi = qmul(&inv, &pre[i]); inv = qmul(&inv, &z[i]); z[i] = zi; }
  for (size_t i = 0; i < n; i++) {
    memcpy(t, ext160 + 160 * i, 32); fe u = f_from_raw(Q, t); memcpy(t, ext160 + 160 * i + 32, 32); fe v = f_from_raw(Q, t);
    affine_t a; a.u = qmul(&u, &z[i]); a.v = qmul(&v, &z[i]); store_affine(&a, out_affine + 64 * i);
  }
  return 0;
}
/* MSM oracle semantics: fold of acc + (P_i * k_i) (lib.rs:183-193, 873-879).  Parallel partial folds are
 * combined in order; the group element is identical. */
API int jjo_msm(size_t n, const uint8_t *scalars, const uint8_t *points, uint8_t *out_affine) {
  ext_t total = ext_identity();
  #pragma omp parallel
  {
    ext_t local = ext_identity();
    #pragma omp for schedule(static) nowait
    for (size_t i = 0; i < n; i++) {
      affine_t a = load_affine(points + 64 * i); ext_t p = affine_to_ext(&a); ext_t r = ext_multiply(&p, scalars + 32 * i);
      eniels_t rn = ext_to_niels(&r); local = ext_add_eniels(&local, &rn);
    }
    #pragma omp critical
    { eniels_t ln = ext_to_niels(&local); total = ext_add_eniels(&total, &ln); }
  }
  affine_t o = ext_to_affine(&total); store_affine(&o, out_affine);
  return 0;
}
/* The same sum by the bucket method (Pippenger), for the CPU baseline of the MSM workload ("CPU Pippenger at 2^20 beside
 * the naive fold").  The reference has NO multi-scalar algorithm (its semantics are the fold above, lib.rs:183-193); this is the
 * textbook method written with the reference's own point formulas only (mixed addition lib.rs:956-968, addition 883-920, doubling
 * 739-828): unsigned c-bit windows of the low 252 scalar bits (the ladder ignores the top four, lib.rs:357-379), one bucket array per
 * window, windows in parallel (OpenMP), running-sum bucket reduction, Horner over the windows.  Same group element as jjo_msm. */
API int jjo_msm_pippenger(size_t n, const uint8_t *scalars, const uint8_t *points, int c, uint8_t *out_affine) {
  if (c < 1 || c > 20) return -1;
  const int W = (252 + c - 1) / c;
  const size_t nb = ((size_t)1 << c) - 1;
  aniels_t *pn = (aniels_t *)malloc((n ? n : 1) * sizeof(aniels_t));
  ext_t *wsum = (ext_t *)malloc((size_t)W * sizeof(ext_t));
  if (!pn || !wsum) { free(pn); free(wsum); return -2; }
  #pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) { affine_t a = load_affine(points + 64 * i); pn[i] = affine_to_niels(&a); }
  int fail = 0;
  #pragma omp parallel for schedule(dynamic, 1)
  for (int w = 0; w < W; w++) {
    ext_t *bk = (ext_t *)malloc(nb * sizeof(ext_t));
    if (!bk) { fail = 1; wsum[w] = ext_identity(); continue; }
    for (size_t j = 0; j < nb; j++) bk[j] = ext_identity();
    for (size_t i = 0; i < n; i++) {
      unsigned d = 0;
      for (int b = 0; b < c; b++) { const int bit = w * c + b; if (bit < 252) d |= (unsigned)ladder_bit(scalars + 32 * i, bit) << b; }
      if (d) bk[d - 1] = ext_add_aniels(&bk[d - 1], &pn[i]);
    }
    ext_t running = ext_identity(), sum = ext_identity();
    for (size_t j = nb; j-- > 0;) {
      eniels_t bn = ext_to_niels(&bk[j]); running = ext_add_eniels(&running, &bn);
      eniels_t rn = ext_to_niels(&running); sum = ext_add_eniels(&sum, &rn);
    }
    wsum[w] = sum;
    free(bk);
  }
  ext_t total = ext_identity();
  for (int w = W - 1; w >= 0; w--) {
    for (int b = 0; b < c; b++) total = ext_double(&total);
    eniels_t sn = ext_to_niels(&wsum[w]); total = ext_add_eniels(&total, &sn);
  }
  affine_t o = ext_to_affine(&total); store_affine(&o, out_affine);
  free(pn); free(wsum);
  return fail ? -2 : 0;
}
/* Sum of affine points (lib.rs:183-193) */
API int jjo_sum(size_t n, const uint8_t *points, uint8_t *out_affine) {
  ext_t acc = ext_identity();
  for (size_t i = 0; i < n; i++) { affine_t a = load_affine(points + 64 * i); aniels_t na = affine_to_niels(&a); acc = ext_add_aniels(&acc, &na); }
  affine_t o = ext_to_affine(&acc); store_affine(&o, out_affine);
  return 0;
}
/* config-1 plumbing (benches/fq_bench.rs:25-33, point_bench.rs:6-11 analogue): timed loops on pre-converted data */
API int jjo_bench_fq_mul(size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out, int reps) {
  for (size_t i = 0; i < n; i++) {
    u64 t[4]; memcpy(t, a + 32 * i, 32); fe x = f_from_raw(Q, t); memcpy(t, b + 32 * i, 32); fe y = f_from_raw(Q, t);
    for (int r = 0; r < reps; r++) x = qmul(&x, &y);
    f_to_bytes(Q, &x, out + 32 * i);
  }
  return 0;
}
API int jjo_bench_double(size_t n, const uint8_t *pa, uint8_t *out_affine, int reps) {
  for (size_t i = 0; i < n; i++) {
    affine_t a = load_affine(pa + 64 * i); ext_t p = affine_to_ext(&a);
    for (int r = 0; r < reps; r++) p = ext_double(&p);
    affine_t o = ext_to_affine(&p); store_affine(&o, out_affine + 64 * i);
  }
  return 0;
}
/* constants self-check: d, 2d, root of unity are the Montgomery forms of the canonical values */
API int jjo_selftest(void) {
  static const u64 d_raw[4] = {0x01065fd6d6343eb1ULL, 0x292d7f6d37579d26ULL, 0xf5fd9207e6bd7fd4ULL, 0x2a9318e74bfa2b48ULL};   /* lib.rs:399-404 */
  static const u64 d2_raw[4] = {0x020cbfadac687d62ULL, 0x525afeda6eaf3a4cULL, 0xebfb240fcd7affa8ULL, 0x552631ce97f45691ULL};  /* lib.rs:407-412 */
  fe d = f_from_raw(Q, d_raw), d2 = f_from_raw(Q, d2_raw);
  if (!f_eq(&d, &FQ_D)) return 1;
  if (!f_eq(&d2, &FQ_D2)) return 2;
  fe dd = f_double(Q, &d); if (!f_eq(&dd, &d2)) return 3;
  /* 7^t */
  static const u64 seven[4] = {7, 0, 0, 0}; fe s7 = f_from_raw(Q, seven);
  u64 t[4] = {0xfffe5bfeffffffffULL, 0x09a1d80553bda402ULL, 0x299d7d483339d808ULL, 0x0000000073eda753ULL};
  fe root = f_pow_vartime(Q, &s7, t); if (!f_eq(&root, &FQ_ROOT)) return 4;
  fe one = FQ.r, r1 = f_mul(Q, &one, &one); if (!f_eq(&r1, &one)) return 5;
  fe rr = f_mul(Q, &FQ.r2, &one); (void)rr;
  u64 onel[4] = {1, 0, 0, 0}; fe o1 = f_from_raw(Q, onel); if (!f_eq(&o1, &FQ.r)) return 6;
  fe o2 = f_from_raw(&FR, onel); if (!f_eq(&o2, &FR.r)) return 7;
  if (FQ.modulus.l[0] * FQ.inv != (u64)-1) return 8;
  if (FR.modulus.l[0] * FR.inv != (u64)-1) return 9;
  return 0;
}
