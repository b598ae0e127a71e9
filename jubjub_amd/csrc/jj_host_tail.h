// Host-side finish of a Pippenger MSM: the Horner combination of the W window sums (c doublings per window, about
// 250 dependent point doublings in all) and the final conversion to affine.  That chain has no parallelism a GPU
// could use -- a lone wavefront issues one VALU instruction per ~9 cycles, 1.5 us per doubling -- while a host core
// does the same doubling in ~0.2 us, so the 16-17 window sums (160 bytes each) are copied back and finished here.
//
// Plain 4 x 64-bit Montgomery arithmetic (radix 2^256) modulo q; every constant is derived at start-up from q and
// d = -10240/10241, nothing is tabulated.  Point formulas: the same completed-point formulas as jj_curve.h
// (reference src/lib.rs:739-828 double, 883-920 add, 1052-1060 into_extended).
#pragma once
#include <stdint.h>
#include <string.h>

namespace jjhost {

typedef unsigned __int128 u128;
struct Fe { uint64_t l[4]; };

static const uint64_t QL[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};

static inline bool geq_q(const uint64_t* a) {
  for (int i = 3; i >= 0; i--) { if (a[i] != QL[i]) return a[i] > QL[i]; }
  return true;
}
static inline void sub_q(uint64_t* a) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) { const u128 t = (u128)a[i] - QL[i] - (uint64_t)br; a[i] = (uint64_t)t; br = (t >> 64) & 1; }
}
static inline Fe add(const Fe& a, const Fe& b) {
  Fe r; u128 cy = 0;
  for (int i = 0; i < 4; i++) { cy += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)cy; cy >>= 64; }
  if (cy || geq_q(r.l)) sub_q(r.l);      // q < 2^255: a + b < 2^256, cy is always 0; kept for clarity
  return r;
}
static inline Fe sub(const Fe& a, const Fe& b) {
  Fe r; u128 br = 0;
  for (int i = 0; i < 4; i++) { const u128 t = (u128)a.l[i] - b.l[i] - (uint64_t)br; r.l[i] = (uint64_t)t; br = (t >> 64) & 1; }
  if (br) { u128 cy = 0; for (int i = 0; i < 4; i++) { cy += (u128)r.l[i] + QL[i]; r.l[i] = (uint64_t)cy; cy >>= 64; } }
  return r;
}
static inline uint64_t ninv64() {      // -q^-1 mod 2^64 by Newton iteration
  uint64_t x = 1;
  for (int i = 0; i < 6; i++) x *= 2 - QL[0] * x;
  return (uint64_t)0 - x;
}
// Montgomery product a*b/2^256 mod q (operand scanning, one reduction step per word)
static inline Fe mul(const Fe& a, const Fe& b) {
  static const uint64_t NINV = ninv64();
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 cy = 0;
    for (int j = 0; j < 4; j++) { cy += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)cy; cy >>= 64; }
    cy += t[4]; t[4] = (uint64_t)cy; t[5] = (uint64_t)(cy >> 64);
    const uint64_t m = t[0] * NINV;
    cy = ((u128)m * QL[0] + t[0]) >> 64;
    for (int j = 1; j < 4; j++) { cy += (u128)m * QL[j] + t[j]; t[j - 1] = (uint64_t)cy; cy >>= 64; }
    cy += t[4]; t[3] = (uint64_t)cy; t[4] = t[5] + (uint64_t)(cy >> 64);
  }
  Fe r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || geq_q(r.l)) sub_q(r.l);
  return r;
}
static inline Fe sqr(const Fe& a) { return mul(a, a); }
static inline Fe dbl(const Fe& a) { return add(a, a); }

struct Consts { Fe one, r2, d2; };
static inline Fe pow_q2(const Fe& a, const Fe& one);
static inline const Consts& consts() {
  static const Consts K = [] {
    Consts k;
    Fe x = {{1, 0, 0, 0}};                       // 2^256 mod q by 256 modular doublings, 2^512 by 256 more
    for (int i = 0; i < 256; i++) x = dbl(x);
    k.one = x;
    for (int i = 0; i < 256; i++) x = dbl(x);
    k.r2 = x;
    const Fe a = mul(Fe{{10240, 0, 0, 0}}, k.r2), b = mul(Fe{{10241, 0, 0, 0}}, k.r2);
    const Fe d = sub(Fe{{0, 0, 0, 0}}, mul(a, pow_q2(b, k.one)));      // d = -10240/10241 (reference src/lib.rs:399-404)
    k.d2 = dbl(d);
    return k;
  }();
  return K;
}
// a^(q-2): plain square-and-multiply, once per MSM
static inline Fe pow_q2(const Fe& a, const Fe& one) {
  uint64_t e[4] = {QL[0] - 2, QL[1], QL[2], QL[3]};
  Fe r = one;
  for (int i = 254; i >= 0; i--) { r = sqr(r); if ((e[i >> 6] >> (i & 63)) & 1) r = mul(r, a); }
  return r;
}
static inline Fe from_canon(const uint8_t* p) { Fe a; memcpy(a.l, p, 32); return mul(a, consts().r2); }
static inline void to_canon(uint8_t* p, const Fe& a) { const Fe c = mul(a, Fe{{1, 0, 0, 0}}); memcpy(p, c.l, 32); }

struct Ext { Fe u, v, z, t1, t2; };
static inline Ext into_extended(const Fe& cu, const Fe& cv, const Fe& cz, const Fe& ct) {
  return Ext{mul(cu, ct), mul(cv, cz), mul(cz, ct), cu, cv};
}
static inline Ext point_dbl(const Ext& p) {
  const Fe uu = sqr(p.u), vv = sqr(p.v), zz2 = dbl(sqr(p.z)), uv2 = sqr(add(p.u, p.v));
  const Fe vpu = add(vv, uu), vmu = sub(vv, uu);
  return into_extended(sub(uv2, vpu), vpu, vmu, sub(zz2, vmu));
}
static inline Ext point_add(const Ext& p, const Ext& q) {
  const Fe a = mul(sub(p.v, p.u), sub(q.v, q.u)), b = mul(add(p.v, p.u), add(q.v, q.u));
  const Fe c = mul(mul(mul(p.t1, p.t2), mul(q.t1, q.t2)), consts().d2), d = dbl(mul(p.z, q.z));
  return into_extended(sub(b, a), add(b, a), add(d, c), sub(d, c));
}
static inline Ext ext_from_canon160(const uint8_t* p) {
  return Ext{from_canon(p), from_canon(p + 32), from_canon(p + 64), from_canon(p + 96), from_canon(p + 128)};
}
static inline Ext identity() { const Fe z = {{0, 0, 0, 0}}; return Ext{z, consts().one, consts().one, z, z}; }
// sum_w 2^(c w) * win[w]; win = W canonical 160-byte extended points
static inline Ext horner(const uint8_t* win160, int W, int c) {
  Ext acc = ext_from_canon160(win160 + (size_t)160 * (W - 1));
  for (int w = W - 2; w >= 0; w--) {
    for (int i = 0; i < c; i++) acc = point_dbl(acc);
    acc = point_add(acc, ext_from_canon160(win160 + (size_t)160 * w));
  }
  return acc;
}
// canonical (u, v), 64 bytes
static inline void to_affine64(uint8_t* out, const Ext& p) {
  const Fe zi = pow_q2(p.z, consts().one);
  to_canon(out, mul(p.u, zi));
  to_canon(out + 32, mul(p.v, zi));
}

}  // namespace jjhost
