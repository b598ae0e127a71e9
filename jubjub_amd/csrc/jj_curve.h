// Jubjub twisted-Edwards group law on CDNA4 — device code, one point per lane, all coordinates in registers.
//
// Formulas are the reference's (same completed-point structure, so the projective point agrees with the Rust code):
//   double        : reference ExtendedPoint::double           src/lib.rs:739-828  (4S + 3M)
//   add ExtNiels  : reference Add<&ExtendedNielsPoint>        src/lib.rs:883-920  (8M)
//   add AffNiels  : reference Add<&AffineNielsPoint>          src/lib.rs:944-968  (7M)
//   into_extended : reference CompletedPoint::into_extended   src/lib.rs:1052-1060
// written for the signed lazy limbs of jj_field.h: subtraction is a limb-wise v_sub, a carry step is inserted only
// where the next product's 64-bit column bound needs it.  Lazy-reduction bounds of every intermediate are verified by
// tools/bounds_check.py (static intervals) and tests/test_emu_field.py (host emulation with a 128-bit shadow).
#pragma once
#include "jj_field.h"

namespace jj {

struct Affine { Fe u, v; };                 // reference AffinePoint          src/lib.rs:80-84
struct Ext { Fe u, v, z, t1, t2; };         // reference ExtendedPoint        src/lib.rs:138-145 ; u, v, z are products ("N"), t1 and t2 stay lazy
struct ANiels { Fe vpu, vmu, t2d; };        // reference AffineNielsPoint     src/lib.rs:254-259
struct ENiels { Fe vpu, vmu, z2, t2d; };    // reference ExtendedNielsPoint   src/lib.rs:326-332 ; z2 = 2Z (the addition only ever uses 2*Z1*Z2)

template <class FT>
struct CurveT {
  typedef FT F;

  static JJ_DEV Ext identity() { Ext p; p.u = F::zero(); p.v = F::one(); p.z = F::one(); p.t1 = F::zero(); p.t2 = F::zero(); return p; }  // lib.rs:680-688
  static JJ_DEV ANiels aniels_identity() { ANiels n; n.vpu = F::one(); n.vmu = F::one(); n.t2d = F::zero(); return n; }                      // lib.rs:263-269
  static JJ_DEV ENiels eniels_identity() { ENiels n; n.vpu = F::one(); n.vmu = F::one(); n.z2 = F::add(F::one(), F::one()); n.t2d = F::zero(); return n; }   // lib.rs:347-354
  static JJ_DEV Ext from_affine(const Affine& a) { Ext p; p.u = a.u; p.v = a.v; p.z = F::one(); p.t1 = a.u; p.t2 = a.v; return p; }           // lib.rs:640-648

  // completed point (u:z, v:t) -> extended; lib.rs:1052-1060.
  static JJ_DEV Ext into_extended(const Fe& cu, const Fe& cv, const Fe& cz, const Fe& ct) {
    Ext p;
    const Fe ou = F::opaque(cu), ov = F::opaque(cv), oz = F::opaque(cz), ot = F::opaque(ct);   // each enters two products: hidden once
    p.u = F::mul_hidden(ou, ot);
    p.v = F::mul_hidden(ov, oz);
    p.z = F::mul_hidden(oz, ot);
    p.t1 = ou;
    p.t2 = ov;
    return p;
  }

  // lib.rs:739-828 (4S + 3M there).  Here 2UV is one product U * (2V) instead of (U+V)^2 - UU - VV: with lazy signed limbs
  // the square route needs a difference, a square, a subtraction AND a carry step on the completed point (its
  // coordinates would meet as 2^30 x 2^30 limbs), the product route needs none of them -- 3S + 4M and 63 additive
  // instructions per doubling instead of 4S + 3M and 115.  Same completed point (2UV, VV+UU, VV-UU, 2ZZ-(VV-UU)), so
  // T1, T2 and the projective coordinates equal the reference's.
  static JJ_DEV Ext dbl(const Ext& p) {
    const Fe ou = F::opaque(p.u);             // U enters the square and the product: hidden once (see Field::opaque)
    const Fe uu = F::sqr_hidden(ou);
    const Fe vv = F::sqr(p.v);
    const Fe zz2 = F::sqr2(p.z);
    const Fe cu = F::mul_hidden(ou, F::opaque(F::dbl(p.v)));   // 2UV   (= T1)
    const Fe vpu = F::add(vv, uu);            // VV + UU
    const Fe vmu = F::sub(vv, uu);            // VV - UU
    const Fe ct = F::sub(zz2, vmu);           // 2Z^2 - (VV-UU), lazy: limbs in (-2^29, 2^30)
    return into_extended(cu, vpu, vmu, ct);
  }

  // shared tail of the four additions: a, b, c, d (d = 2 Z1 Z2, product or lazy 2 Z1) -> extended
  static JJ_DEV Ext add_tail(const Fe& a, const Fe& b, const Fe& c, const Fe& d) {
    return into_extended(F::sub(b, a), F::add(b, a), F::carry(F::add(d, c)), F::sub(d, c));
  }
  // T1*T2 of an accumulator.  T1_SMALL: t1 is a product (after a doubling) or b - a (after an addition), limbs inside
  // (-2^29, 2^29): it meets the lazy t2 as it is.  Otherwise (t1 of unknown provenance, e.g. negated or reloaded) one
  // carry step first.
  template <bool T1_SMALL>
  static JJ_DEV Fe tt(const Ext& p) { if constexpr (T1_SMALL) return F::mul(p.t1, p.t2); else return F::mul(F::carry(p.t1), p.t2); }

  // lib.rs:883-920
  template <bool T1_SMALL = false>
  static JJ_DEV Ext add(const Ext& p, const ENiels& n) {
    const Fe a = F::mul(F::sub(p.v, p.u), n.vmu);
    const Fe b = F::mul(F::add(p.v, p.u), n.vpu);
    const Fe c = F::mul(tt<T1_SMALL>(p), n.t2d);
    const Fe d = F::mul(p.z, n.z2);
    return add_tail(a, b, c, d);
  }
  // the same with T = T1*T2 of p supplied by the caller: an accumulator that is both added to and handed on as an operand
  // (the running sum of the bucket reduce) forms that product once
  static JJ_DEV Ext add_t(const Ext& p, const Fe& T, const ENiels& n) {
    const Fe a = F::mul(F::sub(p.v, p.u), n.vmu);
    const Fe b = F::mul(F::add(p.v, p.u), n.vpu);
    const Fe c = F::mul(T, n.t2d);
    const Fe d = F::mul(p.z, n.z2);
    return add_tail(a, b, c, d);
  }
  // lib.rs:922-940
  template <bool T1_SMALL = false>
  static JJ_DEV Ext sub(const Ext& p, const ENiels& n) {
    const Fe a = F::mul(F::sub(p.v, p.u), n.vpu);
    const Fe b = F::mul(F::add(p.v, p.u), n.vmu);
    const Fe c = F::neg(F::mul(tt<T1_SMALL>(p), n.t2d));
    const Fe d = F::mul(p.z, n.z2);
    return add_tail(a, b, c, d);
  }
  // lib.rs:944-968
  template <bool T1_SMALL = false>
  static JJ_DEV Ext add(const Ext& p, const ANiels& n) {
    const Fe a = F::mul(F::sub(p.v, p.u), n.vmu);
    const Fe b = F::mul(F::add(p.v, p.u), n.vpu);
    const Fe c = F::mul(tt<T1_SMALL>(p), n.t2d);
    return add_tail(a, b, c, F::add(p.z, p.z));
  }
  // lib.rs:970-988
  template <bool T1_SMALL = false>
  static JJ_DEV Ext sub(const Ext& p, const ANiels& n) {
    const Fe a = F::mul(F::sub(p.v, p.u), n.vpu);
    const Fe b = F::mul(F::add(p.v, p.u), n.vmu);
    const Fe c = F::neg(F::mul(tt<T1_SMALL>(p), n.t2d));
    return add_tail(a, b, c, F::add(p.z, p.z));
  }

  // p + n (negmask = 0) or p - n (negmask = ~0) without negating the operand: the subtraction formulas (lib.rs:922-940,
  // 970-988) swap v+u / v-u (two selects) and flip the sign of c (a conditional negation, two VOP2 per limb).
  template <bool T1_SMALL = false>
  static JJ_DEV Ext add_signed(const Ext& p, const ENiels& n, u32 negmask) {
    const Fe a = F::mul(F::sub(p.v, p.u), F::select(n.vmu, n.vpu, negmask));
    const Fe b = F::mul(F::add(p.v, p.u), F::select(n.vpu, n.vmu, negmask));
    const Fe c = F::cneg(F::mul(tt<T1_SMALL>(p), n.t2d), negmask);
    const Fe d = F::mul(p.z, n.z2);
    return add_tail(a, b, c, d);
  }
  template <bool T1_SMALL = false>
  static JJ_DEV Ext add_signed(const Ext& p, const ANiels& n, u32 negmask) {
    const Fe a = F::mul(F::sub(p.v, p.u), F::select(n.vmu, n.vpu, negmask));
    const Fe b = F::mul(F::add(p.v, p.u), F::select(n.vpu, n.vmu, negmask));
    const Fe c = F::cneg(F::mul(tt<T1_SMALL>(p), n.t2d), negmask);
    return add_tail(a, b, c, F::add(p.z, p.z));
  }

  // lib.rs:652-658 : (v+u, v-u, u*v*2d); vpu carried so that it can meet the lazy (V+U) of an accumulator
  static JJ_DEV ANiels to_niels(const Affine& a) {
    ANiels n;
    n.vpu = F::carry(F::add(a.v, a.u));
    n.vmu = F::sub(a.v, a.u);
    n.t2d = F::mul(F::mul(a.u, a.v), F::konst(FqP::D2));
    return n;
  }
  // lib.rs:728-735
  template <bool T1_SMALL = false>
  static JJ_DEV ENiels to_niels(const Ext& p) {
    ENiels n;
    n.vpu = F::carry(F::add(p.v, p.u));
    n.vmu = F::sub(p.v, p.u);
    n.z2 = F::add(p.z, p.z);
    n.t2d = F::mul(tt<T1_SMALL>(p), F::konst(FqP::D2));
    return n;
  }
  // lib.rs:728-735 with T = T1*T2 supplied by the caller
  static JJ_DEV ENiels to_niels_t(const Ext& p, const Fe& T) {
    ENiels n;
    n.vpu = F::carry(F::add(p.v, p.u));
    n.vmu = F::sub(p.v, p.u);
    n.z2 = F::add(p.z, p.z);
    n.t2d = F::mul(T, F::konst(FqP::D2));
    return n;
  }
  // -(vpu, vmu, t2d) = (vmu, vpu, -t2d)   (negation of the underlying point, lib.rs:92-104)
  static JJ_DEV ENiels neg(const ENiels& n) { ENiels r; r.vpu = F::carry(n.vmu); r.vmu = n.vpu; r.z2 = n.z2; r.t2d = F::neg(n.t2d); return r; }
  static JJ_DEV ANiels neg(const ANiels& n) { ANiels r; r.vpu = F::carry(n.vmu); r.vmu = n.vpu; r.t2d = F::neg(n.t2d); return r; }
  // lib.rs:195-211
  static JJ_DEV Ext neg(const Ext& p) { Ext r; r.u = F::neg(p.u); r.v = p.v; r.z = p.z; r.t1 = F::neg(p.t1); r.t2 = p.t2; return r; }

  // masked select: mask all-ones -> b
  static JJ_DEV ENiels select(const ENiels& a, const ENiels& b, u32 mask) {
    ENiels r; r.vpu = F::select(a.vpu, b.vpu, mask); r.vmu = F::select(a.vmu, b.vmu, mask); r.z2 = F::select(a.z2, b.z2, mask); r.t2d = F::select(a.t2d, b.t2d, mask); return r;
  }
  static JJ_DEV ANiels select(const ANiels& a, const ANiels& b, u32 mask) {
    ANiels r; r.vpu = F::select(a.vpu, b.vpu, mask); r.vmu = F::select(a.vmu, b.vmu, mask); r.t2d = F::select(a.t2d, b.t2d, mask); return r;
  }

  static JJ_DEV Ext mul_by_cofactor(const Ext& p) { return dbl(dbl(dbl(p))); }                       // lib.rs:722-724
  static JJ_DEV bool is_identity(const Ext& p) { return F::is_zero(p.u) && F::eq(p.v, p.z); }        // lib.rs:691-696
  static JJ_DEV bool is_small_order(const Ext& p) { return F::is_zero(dbl(dbl(p)).u); }              // lib.rs:699-705
  // [r]P == O  (reference is_torsion_free, lib.rs:709-711, which runs the full 252-step ladder).
  // E(Fq) is cyclic of order 8r, so the same predicate is "the order-8 Tate pairing of P with a generator T of the
  // 8-torsion is trivial": three Miller doubling steps (constant lines, T is fixed) and one exponentiation by
  // (q-1)/8 -- about 250 squarings + 90 multiplications instead of about 1000 + 1300.  Derivation and constants:
  // tools/gen_constants.py torsion_pairing_constants(); equivalence with the ladder is tested on the CPU
  // (tests/test_torsion_pairing.py) and on the GPU against the ladder kernel.
  // The Miller value is zero or undefined only at points of the 8-torsion generated by T, where the answer is
  // "identity only"; every such point makes z = 0 below (or is the identity, handled first).
  // Input must be on the curve (the reference type guarantees it); off-curve input gives an unspecified answer.
  static JJ_DEV bool is_torsion_free(const Affine& a) {
    if (F::is_zero(a.u) && F::eq(a.v, F::one())) return true;
    const Fe pp = F::add(F::one(), a.v), mm = F::sub(F::one(), a.v);
    const Fe l1 = F::sub(pp, F::mul(a.u, F::add(F::mul(F::konst(FqP::TP_A1), a.v), F::konst(FqP::TP_B1))));
    const Fe l2 = F::carry(F::sub(pp, F::mul(a.u, F::add(F::mul(F::konst(FqP::TP_A2), a.v), F::konst(FqP::TP_B2)))));
    const Fe g = F::mul(l1, F::mul(a.u, a.v));
    const Fe g4 = F::sqr(F::sqr(g));
    const Fe k = F::mul(F::sqr(a.u), F::mul(pp, mm));
    const Fe k2 = F::sqr(k);
    const Fe k7 = F::mul(F::mul(k, k2), F::sqr(k2));
    const Fe z = F::mul(F::mul(F::konst(FqP::TP_C), g4), F::mul(F::sqr(l2), k7));
    // z^((q-1)/8) = (z^t)^(2^29),  z^t = z * (z^((t-1)/2))^2
    const Fe w = F::template pow_const<8, FqP::TM1D2>(z);
    Fe b = F::mul(F::mul(z, w), w);
    #pragma unroll 1
    for (int i = 0; i < FqP::TWO_ADICITY - 3; i++) b = F::sqr(b);
    return F::eq(b, F::one());
  }
  // v^2 - u^2 == 1 + d u^2 v^2   (lib.rs:670-675)
  static JJ_DEV bool is_on_curve(const Affine& a) {
    const Fe u2 = F::sqr(a.u), v2 = F::sqr(a.v);
    const Fe lhs = F::sub(v2, u2);
    const Fe rhs = F::add(F::one(), F::mul(F::konst(FqP::D), F::mul(u2, v2)));
    return F::eq(lhs, rhs);
  }

};
typedef CurveT<Fq> Curve;

}  // namespace jj
