// The Horner chain of the MSM host tail (jj_host_tail.h WindowSums::finish: 252 dependent point doublings, one addition per window) on
// AVX-512 IFMA: the four coordinates (U, V, Z, T) of the running point are the four 64-bit lanes of 256-bit vectors, a field element is
// five 52-bit limbs (one vector per limb), and a point doubling is TWO four-lane Montgomery products -- [U U, V V, Z Z, U V], then
// [E F, G H, F G, E H] -- instead of seven scalar 4 x 64-bit products one after the other.  ~4x the scalar chain on a Zen 5 / Ice Lake
// core; used when the CPU has avx512ifma + avx512vl (checked at run time; jj_ctx_set_option(NULL, "host_tail_scalar", 1) forces the scalar chain).
//
// Form.  Montgomery radix 2^260 here, 2^256 in the records and in jj_host_tail.h.  The integers of a record are used AS THEY ARE: read
// in this radix they are the coordinates times 2^-4, and a projective point may be scaled by any constant (U, V, Z and T by the same
// one).  Every formula below is homogeneous (each output of a round is a product of two values of the previous round), so the four
// coordinates always carry the same factor and the factor cancels in u = U / Z, v = V / Z.  The only constant that meets a
// coordinate, 2d, is kept in the radix-2^260 form.
//
// Ranges.  A product returns a value below a b / 2^260 + q with normalised limbs (each below 2^52, as the IFMA instructions read only
// the low 52 bits of an operand).  Sums and differences get multiples of q added so that they are non-negative, then a signed carry
// pass; with 2^260 = 35.3 q the ranges close: coordinates stay below 2.2 q, operands of a product below 6.1 q < 2^258 (worked out at
// each step below).  Formulas: the same completed-point formulas as jj_host_tail.h / jj_curve.h (reference src/lib.rs:739-828 double,
// 883-920 add).
#pragma once
// (inside namespace jjhost: included from the middle of jj_host_tail.h, after the scalar field and point arithmetic and <immintrin.h>)

#define JJ_IFMA __attribute__((target("avx512f,avx512vl,avx512ifma")))

namespace ifma {

constexpr uint64_t M52 = (1ull << 52) - 1;
struct V4 { __m256i l[5]; };      // four field elements: limb j (bits 52 j .. 52 j + 51) of all four in l[j]

static inline void split52(const uint64_t a[4], uint64_t o[5]) {
  o[0] = a[0] & M52; o[1] = ((a[0] >> 52) | (a[1] << 12)) & M52; o[2] = ((a[1] >> 40) | (a[2] << 24)) & M52;
  o[3] = ((a[2] >> 28) | (a[3] << 36)) & M52; o[4] = a[3] >> 16;
}
static inline void join52(const uint64_t l[5], uint64_t a[4]) {      // value below 2^256
  a[0] = l[0] | (l[1] << 52); a[1] = (l[1] >> 12) | (l[2] << 40); a[2] = (l[2] >> 24) | (l[3] << 28); a[3] = (l[3] >> 36) | (l[4] << 16);
}
struct Consts52 {
  uint64_t q[5], qinv, d2[5], r260[5], kq[8][5];      // q; -1/q mod 2^52; 2d 2^260 mod q; 2^260 mod q (the 1 of this radix); k q for k < 8
  // per limb, the four lanes of: the multiples of q a doubling adds [0, 2q, 5q, 3q]; those of an addition [3q,0,0,0], [2q,0,0,0], [2q,0,2q,0];
  // the constant operand [1, 1, 1, 2d] of an addition
  alignas(32) uint64_t k_dbl[5][4], k_addl[5][4], k_addr[5][4], k_addw[5][4], c_add[5][4];
};
static inline const Consts52& consts52() {
  static const Consts52 K = [] {
    Consts52 k;
    split52(QL, k.q);
    uint64_t x = 1;                                   // -q^-1 mod 2^52 by Newton iteration on the low limb
    for (int i = 0; i < 6; i++) x *= 2 - k.q[0] * x;
    k.qinv = ((uint64_t)0 - x) & M52;
    Fe d2 = consts().d2;                              // 2d 2^256 mod q -> 2d 2^260 mod q
    for (int i = 0; i < 4; i++) d2 = dbl(d2);
    split52(d2.l, k.d2);
    Fe one = consts().one;                            // 2^256 mod q -> 2^260 mod q
    for (int i = 0; i < 4; i++) one = dbl(one);
    split52(one.l, k.r260);
    for (int m = 0; m < 8; m++) {
      uint64_t cy = 0;
      for (int j = 0; j < 5; j++) { const uint64_t t = (uint64_t)m * k.q[j] + cy; k.kq[m][j] = j < 4 ? (t & M52) : t; cy = t >> 52; }
    }
    for (int j = 0; j < 5; j++) {
      const uint64_t dbl_[4] = {0, k.kq[2][j], k.kq[5][j], k.kq[3][j]}, addl[4] = {k.kq[3][j], 0, 0, 0}, addr[4] = {k.kq[2][j], 0, 0, 0},
                     addw[4] = {k.kq[2][j], 0, k.kq[2][j], 0}, cadd[4] = {k.r260[j], k.r260[j], k.r260[j], k.d2[j]};
      memcpy(k.k_dbl[j], dbl_, 32); memcpy(k.k_addl[j], addl, 32); memcpy(k.k_addr[j], addr, 32); memcpy(k.k_addw[j], addw, 32); memcpy(k.c_add[j], cadd, 32);
    }
    return k;
  }();
  return K;
}

// a b / 2^260 mod q on all four lanes; operands: normalised limbs, values below 2^258; result: normalised limbs, below a b / 2^260 + q.
// The chain of point operations waits for every product, so what counts is the LATENCY of one: the 25 limb products are formed first
// (independent of the reduction, low columns first), then five reduction
// steps whose critical path is m = c_i q' (one IFMA), the two IFMAs that reach column i + 1 side by side, and two additions.  Column i
// itself is never completed: c_i + low(m q_0) is 0 or 2^52, i.e. its carry is (c_i >> 52) + (low 52 bits of c_i != 0).
JJ_IFMA static inline V4 mul(const V4& a, const V4& b) {
  const Consts52& k = consts52();
  const __m256i z = _mm256_setzero_si256(), qi = _mm256_set1_epi64x((long long)k.qinv), mk = _mm256_set1_epi64x((long long)M52), one = _mm256_set1_epi64x(1);
  const __m256i q0 = _mm256_set1_epi64x((long long)k.q[0]), q1 = _mm256_set1_epi64x((long long)k.q[1]), q2 = _mm256_set1_epi64x((long long)k.q[2]),
                q3 = _mm256_set1_epi64x((long long)k.q[3]), q4 = _mm256_set1_epi64x((long long)k.q[4]);
  // column sums, low columns first (the reduction below starts on them while the high ones are still summed); written out: no loop for
  // the compiler to keep rolled with the accumulators in memory
  __m256i c0 = z, c1 = z, c2 = z, c3 = z, c4 = z, c5 = z, c6 = z, c7 = z, c8 = z, c9 = z;
  c0 = _mm256_madd52lo_epu64(c0, a.l[0], b.l[0]); c1 = _mm256_madd52hi_epu64(c1, a.l[0], b.l[0]);
  c1 = _mm256_madd52lo_epu64(c1, a.l[1], b.l[0]); c2 = _mm256_madd52hi_epu64(c2, a.l[1], b.l[0]);
  c1 = _mm256_madd52lo_epu64(c1, a.l[0], b.l[1]); c2 = _mm256_madd52hi_epu64(c2, a.l[0], b.l[1]);
  c2 = _mm256_madd52lo_epu64(c2, a.l[2], b.l[0]); c3 = _mm256_madd52hi_epu64(c3, a.l[2], b.l[0]);
  c2 = _mm256_madd52lo_epu64(c2, a.l[1], b.l[1]); c3 = _mm256_madd52hi_epu64(c3, a.l[1], b.l[1]);
  c2 = _mm256_madd52lo_epu64(c2, a.l[0], b.l[2]); c3 = _mm256_madd52hi_epu64(c3, a.l[0], b.l[2]);
  c3 = _mm256_madd52lo_epu64(c3, a.l[3], b.l[0]); c4 = _mm256_madd52hi_epu64(c4, a.l[3], b.l[0]);
  c3 = _mm256_madd52lo_epu64(c3, a.l[2], b.l[1]); c4 = _mm256_madd52hi_epu64(c4, a.l[2], b.l[1]);
  c3 = _mm256_madd52lo_epu64(c3, a.l[1], b.l[2]); c4 = _mm256_madd52hi_epu64(c4, a.l[1], b.l[2]);
  c3 = _mm256_madd52lo_epu64(c3, a.l[0], b.l[3]); c4 = _mm256_madd52hi_epu64(c4, a.l[0], b.l[3]);
  c4 = _mm256_madd52lo_epu64(c4, a.l[4], b.l[0]); c5 = _mm256_madd52hi_epu64(c5, a.l[4], b.l[0]);
  c4 = _mm256_madd52lo_epu64(c4, a.l[3], b.l[1]); c5 = _mm256_madd52hi_epu64(c5, a.l[3], b.l[1]);
  c4 = _mm256_madd52lo_epu64(c4, a.l[2], b.l[2]); c5 = _mm256_madd52hi_epu64(c5, a.l[2], b.l[2]);
  c4 = _mm256_madd52lo_epu64(c4, a.l[1], b.l[3]); c5 = _mm256_madd52hi_epu64(c5, a.l[1], b.l[3]);
  c4 = _mm256_madd52lo_epu64(c4, a.l[0], b.l[4]); c5 = _mm256_madd52hi_epu64(c5, a.l[0], b.l[4]);
  c5 = _mm256_madd52lo_epu64(c5, a.l[4], b.l[1]); c6 = _mm256_madd52hi_epu64(c6, a.l[4], b.l[1]);
  c5 = _mm256_madd52lo_epu64(c5, a.l[3], b.l[2]); c6 = _mm256_madd52hi_epu64(c6, a.l[3], b.l[2]);
  c5 = _mm256_madd52lo_epu64(c5, a.l[2], b.l[3]); c6 = _mm256_madd52hi_epu64(c6, a.l[2], b.l[3]);
  c5 = _mm256_madd52lo_epu64(c5, a.l[1], b.l[4]); c6 = _mm256_madd52hi_epu64(c6, a.l[1], b.l[4]);
  c6 = _mm256_madd52lo_epu64(c6, a.l[4], b.l[2]); c7 = _mm256_madd52hi_epu64(c7, a.l[4], b.l[2]);
  c6 = _mm256_madd52lo_epu64(c6, a.l[3], b.l[3]); c7 = _mm256_madd52hi_epu64(c7, a.l[3], b.l[3]);
  c6 = _mm256_madd52lo_epu64(c6, a.l[2], b.l[4]); c7 = _mm256_madd52hi_epu64(c7, a.l[2], b.l[4]);
  c7 = _mm256_madd52lo_epu64(c7, a.l[4], b.l[3]); c8 = _mm256_madd52hi_epu64(c8, a.l[4], b.l[3]);
  c7 = _mm256_madd52lo_epu64(c7, a.l[3], b.l[4]); c8 = _mm256_madd52hi_epu64(c8, a.l[3], b.l[4]);
  c8 = _mm256_madd52lo_epu64(c8, a.l[4], b.l[4]); c9 = _mm256_madd52hi_epu64(c9, a.l[4], b.l[4]);
  __m256i m, cy, x, y;
  m = _mm256_madd52lo_epu64(z, c0, qi);                                        // c_0 + m q = 0 mod 2^52
  cy = _mm256_srli_epi64(c0, 52); cy = _mm256_mask_add_epi64(cy, _mm256_test_epi64_mask(c0, mk), cy, one);
  x = _mm256_madd52hi_epu64(_mm256_add_epi64(c1, cy), m, q0); y = _mm256_madd52lo_epu64(z, m, q1); c1 = _mm256_add_epi64(x, y);
  c2 = _mm256_madd52hi_epu64(c2, m, q1); c2 = _mm256_madd52lo_epu64(c2, m, q2);
  c3 = _mm256_madd52hi_epu64(c3, m, q2); c3 = _mm256_madd52lo_epu64(c3, m, q3);
  c4 = _mm256_madd52hi_epu64(c4, m, q3); c4 = _mm256_madd52lo_epu64(c4, m, q4);
  c5 = _mm256_madd52hi_epu64(c5, m, q4);
  m = _mm256_madd52lo_epu64(z, c1, qi);                                        // c_1 + m q = 0 mod 2^52
  cy = _mm256_srli_epi64(c1, 52); cy = _mm256_mask_add_epi64(cy, _mm256_test_epi64_mask(c1, mk), cy, one);
  x = _mm256_madd52hi_epu64(_mm256_add_epi64(c2, cy), m, q0); y = _mm256_madd52lo_epu64(z, m, q1); c2 = _mm256_add_epi64(x, y);
  c3 = _mm256_madd52hi_epu64(c3, m, q1); c3 = _mm256_madd52lo_epu64(c3, m, q2);
  c4 = _mm256_madd52hi_epu64(c4, m, q2); c4 = _mm256_madd52lo_epu64(c4, m, q3);
  c5 = _mm256_madd52hi_epu64(c5, m, q3); c5 = _mm256_madd52lo_epu64(c5, m, q4);
  c6 = _mm256_madd52hi_epu64(c6, m, q4);
  m = _mm256_madd52lo_epu64(z, c2, qi);                                        // c_2 + m q = 0 mod 2^52
  cy = _mm256_srli_epi64(c2, 52); cy = _mm256_mask_add_epi64(cy, _mm256_test_epi64_mask(c2, mk), cy, one);
  x = _mm256_madd52hi_epu64(_mm256_add_epi64(c3, cy), m, q0); y = _mm256_madd52lo_epu64(z, m, q1); c3 = _mm256_add_epi64(x, y);
  c4 = _mm256_madd52hi_epu64(c4, m, q1); c4 = _mm256_madd52lo_epu64(c4, m, q2);
  c5 = _mm256_madd52hi_epu64(c5, m, q2); c5 = _mm256_madd52lo_epu64(c5, m, q3);
  c6 = _mm256_madd52hi_epu64(c6, m, q3); c6 = _mm256_madd52lo_epu64(c6, m, q4);
  c7 = _mm256_madd52hi_epu64(c7, m, q4);
  m = _mm256_madd52lo_epu64(z, c3, qi);                                        // c_3 + m q = 0 mod 2^52
  cy = _mm256_srli_epi64(c3, 52); cy = _mm256_mask_add_epi64(cy, _mm256_test_epi64_mask(c3, mk), cy, one);
  x = _mm256_madd52hi_epu64(_mm256_add_epi64(c4, cy), m, q0); y = _mm256_madd52lo_epu64(z, m, q1); c4 = _mm256_add_epi64(x, y);
  c5 = _mm256_madd52hi_epu64(c5, m, q1); c5 = _mm256_madd52lo_epu64(c5, m, q2);
  c6 = _mm256_madd52hi_epu64(c6, m, q2); c6 = _mm256_madd52lo_epu64(c6, m, q3);
  c7 = _mm256_madd52hi_epu64(c7, m, q3); c7 = _mm256_madd52lo_epu64(c7, m, q4);
  c8 = _mm256_madd52hi_epu64(c8, m, q4);
  m = _mm256_madd52lo_epu64(z, c4, qi);                                        // c_4 + m q = 0 mod 2^52
  cy = _mm256_srli_epi64(c4, 52); cy = _mm256_mask_add_epi64(cy, _mm256_test_epi64_mask(c4, mk), cy, one);
  x = _mm256_madd52hi_epu64(_mm256_add_epi64(c5, cy), m, q0); y = _mm256_madd52lo_epu64(z, m, q1); c5 = _mm256_add_epi64(x, y);
  c6 = _mm256_madd52hi_epu64(c6, m, q1); c6 = _mm256_madd52lo_epu64(c6, m, q2);
  c7 = _mm256_madd52hi_epu64(c7, m, q2); c7 = _mm256_madd52lo_epu64(c7, m, q3);
  c8 = _mm256_madd52hi_epu64(c8, m, q3); c8 = _mm256_madd52lo_epu64(c8, m, q4);
  c9 = _mm256_madd52hi_epu64(c9, m, q4);
  V4 r;
  __m256i t = c5; cy = _mm256_srli_epi64(t, 52); r.l[0] = _mm256_and_si256(t, mk);
  t = _mm256_add_epi64(c6, cy); cy = _mm256_srli_epi64(t, 52); r.l[1] = _mm256_and_si256(t, mk);
  t = _mm256_add_epi64(c7, cy); cy = _mm256_srli_epi64(t, 52); r.l[2] = _mm256_and_si256(t, mk);
  t = _mm256_add_epi64(c8, cy); cy = _mm256_srli_epi64(t, 52); r.l[3] = _mm256_and_si256(t, mk);
  r.l[4] = _mm256_add_epi64(c9, cy);
  return r;
}
// signed carry pass: limbs of any sign (a non-negative value below 2^260) -> normalised limbs
JJ_IFMA static inline void carry(V4& x) {
  const __m256i mk = _mm256_set1_epi64x((long long)M52);
  for (int j = 0; j < 4; j++) {
    const __m256i c = _mm256_srai_epi64(x.l[j], 52);
    x.l[j] = _mm256_and_si256(x.l[j], mk);
    x.l[j + 1] = _mm256_add_epi64(x.l[j + 1], c);
  }
}
#define JJ_LANES(a, b, c, d) (((d) << 6) | ((c) << 4) | ((b) << 2) | (a))      /* lane 0 <- a, 1 <- b, 2 <- c, 3 <- d */

// from w = [E, G, F, H]: [E F, G H, F G, E H] = [U, V, Z, T] of the result, or (AGAIN: a doubling follows, which reads no T)
// [E F, G H, F G, E F] = [U, V, Z, U], the left operand of that doubling's first product as it is
template <bool AGAIN>
JJ_IFMA static inline V4 second_round(const V4& w) {
  V4 l, r;
  for (int j = 0; j < 5; j++) {
    l.l[j] = _mm256_permute4x64_epi64(w.l[j], JJ_LANES(0, 1, 2, 0));                        // E G F E
    r.l[j] = _mm256_permute4x64_epi64(w.l[j], AGAIN ? JJ_LANES(2, 3, 1, 2) : JJ_LANES(2, 3, 1, 3));     // AGAIN: F H G F (lane 3 = E F = U again)  |  else: F H G H (lane 3 = E H = T)
  }
  return mul(l, r);
}
// 2 P.  UFORM: p = [U, V, Z, U] (what second_round<true> left), else [U, V, Z, T] (T is not read: a doubling needs none).
// Coordinates below 2.2 q in, below 1.6 q out.
template <bool UFORM, bool AGAIN>
JJ_IFMA static inline V4 point_dbl(const V4& p) {
  const Consts52& k = consts52();
  V4 a, b;
  for (int j = 0; j < 5; j++) {
    a.l[j] = UFORM ? p.l[j] : _mm256_mask_permutex_epi64(p.l[j], 0x8, p.l[j], JJ_LANES(0, 0, 0, 0));      // U V Z U
    b.l[j] = _mm256_mask_permutex_epi64(p.l[j], 0x8, p.l[j], JJ_LANES(0, 0, 0, 1));                       // U V Z V
  }
  const V4 s = mul(a, b);                                               // UU VV ZZ UV, each below 2.2^2 / 35.3 + 1 = 1.14 q
  // E = 2 UV (< 2.3 q), G = VV - UU + 2 q (< 3.2 q), F = G - 2 ZZ + 3 q (< 6.2 q), H = 3 q - UU - VV (<= 3 q)
  V4 w;
  for (int j = 0; j < 5; j++) {
    const __m256i x = s.l[j];
    const __m256i p1 = _mm256_maskz_permutex_epi64(0x7, x, JJ_LANES(3, 1, 1, 0));            //  UV  VV  VV   0
    const __m256i p2 = _mm256_maskz_permutex_epi64(0xe, x, JJ_LANES(0, 0, 0, 0));            //   0  UU  UU  UU
    const __m256i p3 = _mm256_maskz_permutex_epi64(0xc, x, JJ_LANES(0, 0, 2, 1));            //   0   0  ZZ  VV
    __m256i t = _mm256_add_epi64(p1, _mm256_load_si256((const __m256i*)k.k_dbl[j]));
    t = _mm256_mask_add_epi64(t, 0x1, t, p1);                                               // + UV once more
    t = _mm256_sub_epi64(t, _mm256_add_epi64(p2, p3));
    w.l[j] = _mm256_mask_sub_epi64(t, 0x4, t, p3);                                          // - ZZ once more
  }
  carry(w);
  return second_round<AGAIN>(w);                                        // EF < 1.4 q, GH < 1.3 q, FG < 1.6 q, EH < 1.2 q
}
// P1 + P2 for points [U, V, Z, T] with coordinates below 2.2 q; result below 1.5 q
JJ_IFMA static inline V4 point_add(const V4& p1, const V4& p2) {
  const Consts52& k = consts52();
  V4 c;                                                                 // [1, 1, 1, 2d] in this radix's form: p2k = [U2, V2, Z2, 2d T2], below 1.1 q
  for (int j = 0; j < 5; j++) c.l[j] = _mm256_load_si256((const __m256i*)k.c_add[j]);
  const V4 p2k = mul(p2, c);
  // left  = [V1 - U1 + 3 q, V1 + U1, T1, Z1]         (< 5.2 q, 4.4 q, 2.2 q, 2.2 q)
  // right = [V2 - U2 + 2 q, V2 + U2, 2d T2, 2 Z2]    (< 3.1 q, 2.2 q, 1.1 q, 2.2 q)
  V4 l, r;
  for (int j = 0; j < 5; j++) {
    const __m256i x = p1.l[j], y = p2k.l[j];
    __m256i t = _mm256_permute4x64_epi64(x, JJ_LANES(1, 1, 3, 2));                          //  V1  V1  T1  Z1
    t = _mm256_add_epi64(t, _mm256_maskz_permutex_epi64(0x2, x, JJ_LANES(0, 0, 0, 0)));     //   0 +U1   0   0
    t = _mm256_sub_epi64(t, _mm256_maskz_permutex_epi64(0x1, x, JJ_LANES(0, 0, 0, 0)));     // -U1   0   0   0
    l.l[j] = _mm256_add_epi64(t, _mm256_load_si256((const __m256i*)k.k_addl[j]));
    __m256i u = _mm256_permute4x64_epi64(y, JJ_LANES(1, 1, 3, 2));                          //  V2  V2 2dT2 Z2
    u = _mm256_add_epi64(u, _mm256_maskz_permutex_epi64(0xa, y, JJ_LANES(0, 0, 0, 2)));     //   0 +U2   0 +Z2
    u = _mm256_sub_epi64(u, _mm256_maskz_permutex_epi64(0x1, y, JJ_LANES(0, 0, 0, 0)));     // -U2   0   0   0
    r.l[j] = _mm256_add_epi64(u, _mm256_load_si256((const __m256i*)k.k_addr[j]));
  }
  carry(l); carry(r);
  const V4 s = mul(l, r);                                               // A < 1.5 q, B < 1.3 q, C < 1.1 q, D < 1.2 q
  // E = B - A + 2 q, G = D + C, F = D - C + 2 q, H = B + A
  V4 w;
  for (int j = 0; j < 5; j++) {
    const __m256i x = s.l[j];
    const __m256i bd = _mm256_permute4x64_epi64(x, JJ_LANES(1, 3, 3, 1)), ac = _mm256_permute4x64_epi64(x, JJ_LANES(0, 2, 2, 0));
    __m256i t = _mm256_add_epi64(bd, _mm256_maskz_mov_epi64(0xa, ac));
    t = _mm256_sub_epi64(t, _mm256_maskz_mov_epi64(0x5, ac));
    w.l[j] = _mm256_add_epi64(t, _mm256_load_si256((const __m256i*)k.k_addw[j]));
  }
  carry(w);
  return second_round<false>(w);
}

// a point [U, V, Z, T] in memory (limb-major, as the vectors hold it)
struct P4 { alignas(32) uint64_t l[5][4]; };
static inline void pack_point(P4& o, const Fe& u, const Fe& v, const Fe& z, const Fe& t) {
  uint64_t a[4][5];
  split52(u.l, a[0]); split52(v.l, a[1]); split52(z.l, a[2]); split52(t.l, a[3]);
  for (int j = 0; j < 5; j++) for (int i = 0; i < 4; i++) o.l[j][i] = a[i][j];
}
JJ_IFMA static inline V4 ld(const P4& p) { V4 r; for (int j = 0; j < 5; j++) r.l[j] = _mm256_load_si256((const __m256i*)p.l[j]); return r; }
JJ_IFMA static inline void st(P4& p, const V4& v) { for (int j = 0; j < 5; j++) _mm256_store_si256((__m256i*)p.l[j], v.l[j]); }
// window sums of several records meet here: acc (+)= the point with canonical coordinates c[0..3] = U, V, Z, T of a record
JJ_IFMA static inline void accumulate(P4& acc, bool have, const Fe* c) {
  P4 n;
  pack_point(n, c[0], c[1], c[2], c[3]);
  if (!have) { acc = n; return; }
  st(acc, point_add(ld(acc), ld(n)));
}
// and back to the scalar code: every coordinate (below 2.2 q) is brought below q; T travels as t1 with t2 = 1 (as ext_from_record leaves it)
static inline Ext unpack_point(const P4& p) {
  Fe c[4];
  for (int i = 0; i < 4; i++) {
    const uint64_t l[5] = {p.l[0][i], p.l[1][i], p.l[2][i], p.l[3][i], p.l[4][i]};
    join52(l, c[i].l);
    while (geq_q(c[i].l)) sub_q(c[i].l);
  }
  return Ext{c[0], c[1], c[2], c[3], consts().one};
}
// sum_w 2^(start_w) S_w by Horner from the top window (WindowSums::finish)
JJ_IFMA static inline Ext horner(int W, const bool* have, const P4* sum) {
  V4 acc;
  bool any = false;
  for (int w = W - 1; w >= 0; w--) {
    if (any) {
      // width(w) doublings; all but the last pass [U, V, Z, U] on; the last one forms T when an addition (or the end) follows
      const int n = win_width(W, w);
      const bool t_needed = have[w] || w == 0;
      if (n == 1) acc = t_needed ? point_dbl<false, false>(acc) : point_dbl<false, true>(acc);
      else {
        acc = point_dbl<false, true>(acc);
        for (int i = 1; i < n - 1; i++) acc = point_dbl<true, true>(acc);
        acc = t_needed ? point_dbl<true, false>(acc) : point_dbl<true, true>(acc);
      }
      // (without an addition the window leaves [U, V, Z, U]: wrong only in T, which the next window's first doubling does not read)
    }
    if (have[w]) { const V4 s = ld(sum[w]); acc = any ? point_add(acc, s) : s; any = true; }
  }
  if (!any) return identity();
  P4 out;
  st(out, acc);
  return unpack_point(out);
}
// jj_ctx_set_option(NULL, "host_tail_scalar", 1): the scalar 4 x 64-bit chain even where the IFMA chain is available (tests, measurements)
inline std::atomic<int>& force_scalar() { static std::atomic<int> v{0}; return v; }
static inline bool available() {
  static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512ifma");
  return ok && !force_scalar().load(std::memory_order_relaxed);
}

}  // namespace ifma
