// Host-side finish of an MSM: the device leaves a RECORD of window sums (jj_msm_kernels.h: 64-byte header, then one 128-byte
// point per window: U, V, Z, T = T1 T2, already in the 4 x 64-bit Montgomery form used here); the window sums of several records
// (passes, devices, ranks) are added window by window, the windows are combined by Horner (252 dependent point doublings in all)
// and the result is converted to affine.  That chain has no parallelism a GPU could use -- a quad of lanes needs two rounds of
// ~300 instructions, 1.4 us, per doubling -- while a host core does the same doubling in ~0.1 us (scalar code below) or ~0.04 us
// (jj_host_tail_ifma.h: the four coordinates in the lanes of AVX-512 IFMA vectors, two products per doubling; taken at run time when
// the CPU has it, for the Horner chain and the window-by-window sums; this file's scalar code is the fallback and the reference the
// tests hold it to).
//
// Plain 4 x 64-bit Montgomery arithmetic (radix 2^256) modulo q; every constant is derived at start-up from q and
// d = -10240/10241, nothing is tabulated.  Point formulas: the same completed-point formulas as jj_curve.h
// (reference src/lib.rs:739-828 double, 883-920 add, 1052-1060 into_extended).
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>

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
// a^(q-2) with 4-bit fixed windows (252 squarings + 63 + 14 multiplications), once per MSM
static inline Fe pow_q2(const Fe& a, const Fe& one) {
  const uint64_t e[4] = {QL[0] - 2, QL[1], QL[2], QL[3]};
  Fe tab[16];
  tab[0] = one; tab[1] = a;
  for (int i = 2; i < 16; i++) tab[i] = mul(tab[i - 1], a);
  Fe r = tab[(e[3] >> 60) & 15];
  for (int i = 62; i >= 0; i--) {
    r = sqr(sqr(sqr(sqr(r))));
    const unsigned nib = (unsigned)(e[i >> 4] >> (4 * (i & 15))) & 15u;
    if (nib) r = mul(r, tab[nib]);
  }
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
// ---- records of partial window sums (layout: jj_msm_kernels.h MSM_REC_*; include/jubjub_hip.h jj_msm_partial)
constexpr uint32_t REC_MAGIC = 0x504D4A4Au;       // "JJMP"
constexpr int REC_HDR_BYTES = 64, REC_PT_BYTES = 128, REC_MAX_W = 64;
constexpr size_t REC_MAX_BYTES = REC_HDR_BYTES + (size_t)REC_MAX_W * REC_PT_BYTES;
// width of window w when W windows tile the 253 bits of a recoded scalar: 253 = W c + r, the r low windows are one bit wider
static inline int win_width(int W, int w) { const int c = 253 / W, r = 253 % W; return c + (w < r ? 1 : 0); }
struct RecHeader { uint32_t magic, version, W, nblk; uint64_t mask, n; };
static inline bool rec_header(const uint8_t* rec, RecHeader* h) {
  memcpy(h, rec, sizeof(RecHeader));
  return h->magic == REC_MAGIC && h->version == 2 && h->W >= 1 && h->W <= (uint32_t)REC_MAX_W && h->nblk == 1;
}
static inline size_t rec_bytes(int W) { return REC_HDR_BYTES + (size_t)W * REC_PT_BYTES; }
#if defined(__x86_64__) && !defined(JJ_NO_IFMA)
#define JJ_HAVE_IFMA_TAIL 1
}  // namespace jjhost (closed around the system header the next file includes)
#include <immintrin.h>
namespace jjhost {
#include "jj_host_tail_ifma.h"        // ifma::available(), ifma::horner(): the Horner chain on AVX-512 IFMA, chosen at run time
#endif
// a window's point: four field elements in Montgomery form, each below q (checked: a damaged record must not reach the arithmetic)
static inline bool ext_from_record(const uint8_t* p, Ext* out) {
  Fe c[4];
  memcpy(c, p, 128);
  for (int i = 0; i < 4; i++) if (geq_q(c[i].l)) return false;
  *out = Ext{c[0], c[1], c[2], c[3], consts().one};           // t1 * t2 = T
  return true;
}
// Window sums of one window layout; records with the same W (every pass / rank that saw the same number of terms) meet here
// window by window, so that the Horner chain runs once for all of them.
struct WindowSums {
  int W = 0;
  bool have[REC_MAX_W];
  Ext sum[REC_MAX_W];                 // scalar path
#ifdef JJ_HAVE_IFMA_TAIL
  ifma::P4 vsum[REC_MAX_W];           // AVX-512 IFMA path: the same sums as [U, V, Z, T] limb vectors
#endif
  bool add_record(const uint8_t* rec) {
    RecHeader h;
    if (!rec_header(rec, &h)) return false;
    if (W == 0) { W = (int)h.W; for (int w = 0; w < W; w++) have[w] = false; }
    if ((int)h.W != W) return false;
    for (int w = 0; w < W; w++) {
      if (!((h.mask >> w) & 1)) continue;
      const uint8_t* pt = rec + REC_HDR_BYTES + (size_t)w * REC_PT_BYTES;
#ifdef JJ_HAVE_IFMA_TAIL
      if (ifma::available()) {
        Fe c[4];
        memcpy(c, pt, 128);
        for (int i = 0; i < 4; i++) if (geq_q(c[i].l)) return false;
        ifma::accumulate(vsum[w], have[w], c);
        have[w] = true;
        continue;
      }
#endif
      Ext p;
      if (!ext_from_record(pt, &p)) return false;
      if (have[w]) sum[w] = point_add(sum[w], p); else { sum[w] = p; have[w] = true; }
    }
    return true;
  }
  // sum_w 2^(start_w) S_w by Horner from the top window: width(w) doublings, then + S_w
  Ext finish() const {
#ifdef JJ_HAVE_IFMA_TAIL
    if (ifma::available()) return ifma::horner(W, have, vsum);
#endif
    return finish_scalar();
  }
  Ext finish_scalar() const {
    Ext acc = identity();
    bool any = false;
    for (int w = W - 1; w >= 0; w--) {
      if (any) for (int i = 0; i < win_width(W, w); i++) acc = point_dbl(acc);
      if (have[w]) { acc = any ? point_add(acc, sum[w]) : sum[w]; any = true; }
    }
    return acc;
  }
};
// `count` records, `stride` bytes apart (any mix of window layouts) -> the sum of the MSMs they stand for; false on a bad header
static inline bool combine_records(const uint8_t* recs, size_t count, size_t stride, Ext* out) {
  WindowSums groups[4];
  int ng = 0;
  Ext extra = identity();
  for (size_t i = 0; i < count; i++) {
    const uint8_t* rec = recs + i * stride;
    RecHeader h;
    if (!rec_header(rec, &h)) return false;
    int g = 0;
    while (g < ng && groups[g].W != (int)h.W) g++;
    if (g == ng) {
      if (ng == 4) { WindowSums one; if (!one.add_record(rec)) return false; extra = point_add(extra, one.finish()); continue; }   // a fifth layout: on its own
      ng++;
    }
    if (!groups[g].add_record(rec)) return false;
  }
  Ext total = extra;
  for (int g = 0; g < ng; g++) total = point_add(total, groups[g].finish());
  *out = total;
  return true;
}
// canonical (u, v), 64 bytes
static inline void to_affine64(uint8_t* out, const Ext& p) {
  const Fe zi = pow_q2(p.z, consts().one);
  to_canon(out, mul(p.u, zi));
  to_canon(out + 32, mul(p.v, zi));
}

}  // namespace jjhost
