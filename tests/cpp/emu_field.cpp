// Host emulation of the device field / curve arithmetic (jubjub_amd/csrc/jj_field.h, jj_curve.h compiled with
// -DJJ_HOST_EMU): the SAME formulas and limb arithmetic the HIP kernels run, executed on the CPU with a 128-bit shadow
// of every 64-bit column accumulator.  Test infrastructure only (tests/test_emu_field.py drives it through ctypes and
// compares with the oracle); it lets the signed lazy-reduction bounds and the point formulas be checked without a GPU.
// It is NOT a CPU fallback of the product: nothing in jubjub_amd/ links or loads it.
#include <stdint.h>
#include <string.h>
#define JJ_HOST_EMU 1
#include "../../jubjub_amd/csrc/jj_curve.h"

using namespace jj;

static int g_overflow = 0;
extern "C" void jj_emu_overflow(const char*) { g_overflow++; }
extern "C" int emu_overflow_count(void) { return g_overflow; }
extern "C" void emu_overflow_reset(void) { g_overflow = 0; }

static void ld(u32 (&w)[8], const uint8_t* p) { memcpy(w, p, 32); }
static void st(uint8_t* p, const u32 (&w)[8]) { memcpy(p, w, 32); }

template <class F>
static void field_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, uint8_t* ok) {
  u32 wa[8], wb[8], wo[8];
  ld(wa, a);
  if (b) ld(wb, b);
  Fe x = F::from_words(wa), y = b ? F::from_words(wb) : F::zero(), r = F::zero();
  *ok = 1;
  switch (op) {
    case 0: r = F::add(x, y); break;
    case 1: r = F::sub(x, y); break;
    case 2: r = F::mul(x, y); break;
    case 3: r = F::neg(x); break;
    case 4: r = F::sqr(x); break;
    case 5: r = F::dbl(x); break;
    case 6: *ok = !F::is_zero(x); r = F::invert(x); break;
    case 7: r = F::sqr2(x); break;
    case 8: r = F::canon(F::sub(F::dbl(x), y)); break;   // canon(): Montgomery-form representative in [0, p)
    case 9: *ok = F::eq(x, y); r = x; break;
  }
  F::to_words(wo, r);
  st(out, wo);
}
extern "C" void emu_fq_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, uint8_t* ok) { field_op<Fq>(op, a, b, out, ok); }
extern "C" void emu_fr_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, uint8_t* ok) { field_op<Fr>(op, a, b, out, ok); }
// from_bytes (checked) and from_bytes_wide
extern "C" void emu_from_bytes(int fr, const uint8_t* a, uint8_t* out, uint8_t* ok) {
  u32 wa[8], wo[8]; ld(wa, a); bool k;
  if (fr) { Fe r = Fr::from_words_checked(wa, k); Fr::to_words(wo, r); } else { Fe r = Fq::from_words_checked(wa, k); Fq::to_words(wo, r); }
  if (!k) memset(wo, 0, 32);
  *ok = k; st(out, wo);
}
extern "C" void emu_from_wide(int fr, const uint8_t* a64, uint8_t* out) {
  u32 lo[8], hi[8], wo[8]; ld(lo, a64); ld(hi, a64 + 32);
  if (fr) Fr::to_words(wo, Fr::from_words_wide(lo, hi)); else Fq::to_words(wo, Fq::from_words_wide(lo, hi));
  st(out, wo);
}

static Affine load_affine(const uint8_t* p) { u32 wu[8], wv[8]; ld(wu, p); ld(wv, p + 32); Affine a; a.u = Fq::from_words(wu); a.v = Fq::from_words(wv); return a; }
static void store_affine(uint8_t* out, const Ext& e) {
  const Fe zi = Fq::invert(e.z);
  u32 w[8];
  Fq::to_words(w, Fq::mul(e.u, zi)); st(out, w);
  Fq::to_words(w, Fq::mul(e.v, zi)); st(out + 32, w);
}
static u32 window(const u32 (&k)[8], int w, int i) {
  const int bit = w * i, wi = bit >> 5, sh = bit & 31;
  uint64_t both = k[wi];
  if (wi < 7) both |= (uint64_t)k[wi + 1] << 32;
  return (u32)(both >> sh) & ((1u << w) - 1u);
}
// the signed 5-bit window ladder of k_varbase (jj_kernels.h varbase_windowed), same operation order
extern "C" void emu_varbase(const uint8_t* scalar, const uint8_t* point, uint8_t* out64) {
  constexpr int W = 5, TAB = 16, NWIN = (253 + W - 1) / W;
  u32 k[8]; ld(k, scalar);
  const Affine P = load_affine(point);
  const ANiels pn = Curve::to_niels(P);
  ENiels tab[TAB];
  Ext cur = Curve::from_affine(P);
  tab[0] = Curve::to_niels<true>(cur);
  for (int j = 1; j < TAB; j++) { cur = Curve::add<true>(cur, pn); tab[j] = Curve::to_niels<true>(cur); }
  k[7] &= 0x0fffffffu;
  { uint64_t c = 0; for (int i = 0; i < 8; i++) { u32 rc = 0; for (int j = 0; j < NWIN - 1; j++) { const int bit = W * j + W - 1; if ((bit >> 5) == i) rc |= 1u << (bit & 31); } const uint64_t t = (uint64_t)k[i] + rc + c; k[i] = (u32)t; c = t >> 32; } }
  Ext acc = Curve::identity();
  for (int i = NWIN - 1; i >= 0; i--) {
    int d = (int)window(k, W, i);
    if (i != NWIN - 1) d -= TAB;
    const u32 neg = d < 0 ? ~0u : 0u, a = (u32)(d < 0 ? -d : d);
    const ENiels e = Curve::select(tab[a ? a - 1 : 0], Curve::eniels_identity(), a == 0 ? ~0u : 0u);
    acc = Curve::add_signed<true>(acc, e, neg);
    if (i > 0) for (int s = 0; s < W; s++) acc = Curve::dbl(acc);
  }
  store_affine(out64, acc);
}
// the reference's exact ladder (k_varbase_exact): all five coordinates, canonical
extern "C" void emu_varbase_exact(const uint8_t* scalar, const uint8_t* point, uint8_t* out160) {
  u32 k[8]; ld(k, scalar);
  const ENiels pn = Curve::to_niels(Curve::from_affine(load_affine(point)));
  const ENiels zero = Curve::eniels_identity();
  Ext acc = Curve::identity();
  for (int i = 251; i >= 0; i--) {
    const u32 bit = (k[i >> 5] >> (i & 31)) & 1u;
    acc = Curve::dbl(acc);
    acc = Curve::add<true>(acc, Curve::select(zero, pn, 0u - bit));
  }
  u32 w[8];
  Fq::to_words(w, acc.u); st(out160, w); Fq::to_words(w, acc.v); st(out160 + 32, w); Fq::to_words(w, acc.z); st(out160 + 64, w);
  Fq::to_words(w, acc.t1); st(out160 + 96, w); Fq::to_words(w, acc.t2); st(out160 + 128, w);
}
// sum_i (+/-) P_i with affine-Niels operands after additions only (the MSM bucket / fixed-base inner loop), then doublings
extern "C" void emu_signed_sum(int n, const uint8_t* points, const uint8_t* signs, int doublings, uint8_t* out64) {
  Ext acc = Curve::identity();
  for (int i = 0; i < n; i++) acc = Curve::add_signed<true>(acc, Curve::to_niels(load_affine(points + 64 * i)), signs[i] ? ~0u : 0u);
  for (int i = 0; i < doublings; i++) acc = Curve::dbl(acc);
  // fold with itself through the extended + extended path (to_niels(ext) + add<true>), then subtract it again
  const ENiels en = Curve::to_niels<true>(acc);
  Ext twice = Curve::add<true>(Curve::add<true>(Curve::identity(), en), en);
  Ext back = Curve::sub<true>(twice, en);
  store_affine(out64, back);
}
// the arithmetic of k_normalize (jj_kernels.h) for one element: (U, V, Z) = (u s, v s, s) -> affine through the plain-form
// inverse and canon_plain_product; also exercises is_zero_product on Z
extern "C" int emu_normalize(const uint8_t* point, const uint8_t* scale32, uint8_t* out64) {
  const Affine a = load_affine(point);
  u32 ws[8]; ld(ws, scale32);
  const Fe sc = Fq::from_words(ws);
  const Fe U = Fq::mul(a.u, sc), V = Fq::mul(a.v, sc), Z = Fq::mul(Fq::one(), sc);
  const bool zz = Fq::is_zero_product(Z);
  const Fe zinv = Fq::select(Fq::invert(Z), Fq::zero(), zz ? ~0u : 0u);
  const Fe zp = Fq::mul(zinv, Fq::plain_one());
  u32 w[8];
  Fq::pack(w, Fq::canon_plain_product(Fq::mul(U, zp))); st(out64, w);
  Fq::pack(w, Fq::canon_plain_product(Fq::mul(V, zp))); st(out64 + 32, w);
  return (zz ? 1 : 0) | (Fq::is_zero_product(Fq::one()) ? 2 : 0) | (Fq::is_zero_product(Fq::mul(Fq::zero(), sc)) ? 0 : 4) | (Fq::is_zero(Z) != zz ? 8 : 0);
}
extern "C" int emu_predicates(const uint8_t* point) {
  const Affine a = load_affine(point);
  const Ext e = Curve::from_affine(a);
  return (Curve::is_on_curve(a) ? 1 : 0) | (Curve::is_torsion_free(a) ? 2 : 0) | (Curve::is_small_order(e) ? 4 : 0) | (Curve::is_identity(e) ? 8 : 0) |
         (Curve::is_identity(Curve::mul_by_cofactor(e)) ? 16 : 0);
}
