// Jubjub twisted-Edwards group law on CDNA4 — device code, one point per lane, all coordinates in registers.
//
// Formulas are the reference's (same completed-point structure, so projective coordinates agree with the
// Rust code up to the field representation):
//   double        : reference ExtendedPoint::double           src/lib.rs:739-828  (4S + 3M)
//   add ExtNiels  : reference Add<&ExtendedNielsPoint>        src/lib.rs:883-920  (8M)
//   add AffNiels  : reference Add<&AffineNielsPoint>          src/lib.rs:944-968  (7M)
//   into_extended : reference CompletedPoint::into_extended   src/lib.rs:1052-1060
// Lazy-reduction bounds of every intermediate are verified by tools/bounds_check.py.
#pragma once
#include "jj_field.h"

namespace jj {

struct Affine { Fe u, v; };                 // reference AffinePoint          src/lib.rs:80-84
struct Ext { Fe u, v, z, t1, t2; };         // reference ExtendedPoint        src/lib.rs:138-145 ; t1 is kept lazy (limbs < 2^31)
struct ANiels { Fe vpu, vmu, t2d; };        // reference AffineNielsPoint     src/lib.rs:254-259
struct ENiels { Fe vpu, vmu, z, t2d; };     // reference ExtendedNielsPoint   src/lib.rs:326-332

// FT: the field flavour (Field<FqP, PIN>); Curve = CurveT<Fq> everywhere except register-bound kernels.
template <class FT>
struct CurveT {
  typedef FT F;

  static JJ_DEV Ext identity() { Ext p; p.u = F::zero(); p.v = F::one(); p.z = F::one(); p.t1 = F::zero(); p.t2 = F::zero(); return p; }  // lib.rs:680-688
  static JJ_DEV ANiels aniels_identity() { ANiels n; n.vpu = F::one(); n.vmu = F::one(); n.t2d = F::zero(); return n; }                      // lib.rs:263-269
  static JJ_DEV ENiels eniels_identity() { ENiels n; n.vpu = F::one(); n.vmu = F::one(); n.z = F::one(); n.t2d = F::zero(); return n; }       // lib.rs:347-354
  static JJ_DEV Ext from_affine(const Affine& a) { Ext p; p.u = a.u; p.v = a.v; p.z = F::one(); p.t1 = a.u; p.t2 = a.v; return p; }           // lib.rs:640-648

  // completed point (u:z, v:t) -> extended; lib.rs:1052-1060.  cu,ct,cz N-like; cv L.
  static JJ_DEV Ext into_extended(const Fe& cu, const Fe& cv, const Fe& cz, const Fe& ct) {
    Ext p;
    p.u = F::mul(cu, ct);
    p.v = F::mul(cv, cz);
    p.z = F::mul(cz, ct);
    p.t1 = cu;
    p.t2 = cv;
    return p;
  }

  // lib.rs:739-828
  static JJ_DEV Ext dbl(const Ext& p) {
    const Fe uu = F::sqr(p.u);
    const Fe vv = F::sqr(p.v);
    const Fe zz = F::sqr(p.z);
    const Fe uv2 = F::sqr(F::add(p.u, p.v));
    const Fe vpu = F::add(vv, uu);            // VV + UU   (L)
    const Fe vmu = F::sub(vv, uu);            // VV - UU   (N, +3p)
    const Fe cu = F::sub_lazy(uv2, vpu);      // (U+V)^2 - (VV+UU)   (lazy: meets the carried ct, and is T1)
    const Fe ct = F::dbl_sub_wide(zz, vmu);   // 2Z^2 - (VV-UU)
    return into_extended(cu, vpu, vmu, ct);
  }

  // lib.rs:883-920
  static JJ_DEV Ext add(const Ext& p, const ENiels& n) {
    const Fe a = F::mul(F::sub(p.v, p.u), n.vmu);
    const Fe b = F::mul(F::add(p.v, p.u), n.vpu);
    const Fe c = F::mul(F::mul(F::carry(p.t1), p.t2), n.t2d);
    const Fe zz = F::mul(p.z, n.z);
    const Fe d = F::add(zz, zz);
    return into_extended(F::sub_lazy(b, a), F::add(b, a), F::carry(F::add(d, c)), F::sub(d, c));
  }
  // lib.rs:922-940
  static JJ_DEV Ext sub(const Ext& p, const ENiels& n) {
    const Fe a = F::mul(F::sub(p.v, p.u), n.vpu);
    const Fe b = F::mul(F::add(p.v, p.u), n.vmu);
    const Fe c = F::mul(F::mul(F::carry(p.t1), p.t2), n.t2d);
    const Fe zz = F::mul(p.z, n.z);
    const Fe d = F::add(zz, zz);
    return into_extended(F::sub_lazy(b, a), F::add(b, a), F::sub(d, c), F::carry(F::add(d, c)));
  }
  // lib.rs:944-968
  static JJ_DEV Ext add(const Ext& p, const ANiels& n) {
    const Fe a = F::mul(F::sub(p.v, p.u), n.vmu);
    const Fe b = F::mul(F::add(p.v, p.u), n.vpu);
    const Fe c = F::mul(F::mul(F::carry(p.t1), p.t2), n.t2d);
    const Fe d = F::add(p.z, p.z);
    return into_extended(F::sub_lazy(b, a), F::add(b, a), F::carry(F::add(d, c)), F::sub(d, c));
  }
  // lib.rs:970-988
  static JJ_DEV Ext sub(const Ext& p, const ANiels& n) {
    const Fe a = F::mul(F::sub(p.v, p.u), n.vpu);
    const Fe b = F::mul(F::add(p.v, p.u), n.vmu);
    const Fe c = F::mul(F::mul(F::carry(p.t1), p.t2), n.t2d);
    const Fe d = F::add(p.z, p.z);
    return into_extended(F::sub_lazy(b, a), F::add(b, a), F::sub(d, c), F::carry(F::add(d, c)));
  }

  // p + n (negmask = 0) or p - n (negmask = ~0) without negating the operand: the subtraction formulas (lib.rs:922-940,
  // 970-988) swap v+u / v-u and d+c / d-c, so four selects replace a field negation and three selects.
  static JJ_DEV Ext add_signed(const Ext& p, const ENiels& n, u32 negmask) {
    const Fe a = F::mul(F::sub(p.v, p.u), F::select(n.vmu, n.vpu, negmask));
    const Fe b = F::mul(F::add(p.v, p.u), F::select(n.vpu, n.vmu, negmask));
    const Fe c = F::mul(F::mul(F::carry(p.t1), p.t2), n.t2d);
    const Fe zz = F::mul(p.z, n.z);
    const Fe d = F::add(zz, zz);
    const Fe plus = F::carry(F::add(d, c)), minus = F::sub(d, c);
    return into_extended(F::sub_lazy(b, a), F::add(b, a), F::select(plus, minus, negmask), F::select(minus, plus, negmask));
  }
  static JJ_DEV Ext add_signed(const Ext& p, const ANiels& n, u32 negmask) {
    const Fe a = F::mul(F::sub(p.v, p.u), F::select(n.vmu, n.vpu, negmask));
    const Fe b = F::mul(F::add(p.v, p.u), F::select(n.vpu, n.vmu, negmask));
    const Fe c = F::mul(F::mul(F::carry(p.t1), p.t2), n.t2d);
    const Fe d = F::add(p.z, p.z);
    const Fe plus = F::carry(F::add(d, c)), minus = F::sub(d, c);
    return into_extended(F::sub_lazy(b, a), F::add(b, a), F::select(plus, minus, negmask), F::select(minus, plus, negmask));
  }

  // lib.rs:652-658 : (v+u, v-u, u*v*2d), all N
  static JJ_DEV ANiels to_niels(const Affine& a) {
    ANiels n;
    n.vpu = F::carry(F::add(a.v, a.u));
    n.vmu = F::sub(a.v, a.u);
    n.t2d = F::mul(F::mul(a.u, a.v), F::konst(FqP::D2));
    return n;
  }
  // lib.rs:728-735
  static JJ_DEV ENiels to_niels(const Ext& p) {
    ENiels n;
    n.vpu = F::carry(F::add(p.v, p.u));
    n.vmu = F::sub(p.v, p.u);
    n.z = p.z;
    n.t2d = F::mul(F::mul(F::carry(p.t1), p.t2), F::konst(FqP::D2));
    return n;
  }
  // -(vpu, vmu, t2d) = (vmu, vpu, -t2d)   (negation of the underlying point, lib.rs:92-104)
  static JJ_DEV ENiels neg(const ENiels& n) { ENiels r; r.vpu = n.vmu; r.vmu = n.vpu; r.z = n.z; r.t2d = F::neg(n.t2d); return r; }
  static JJ_DEV ANiels neg(const ANiels& n) { ANiels r; r.vpu = n.vmu; r.vmu = n.vpu; r.t2d = F::neg(n.t2d); return r; }
  // lib.rs:195-211
  static JJ_DEV Ext neg(const Ext& p) { Ext r; r.u = F::neg(p.u); r.v = p.v; r.z = p.z; r.t1 = F::neg(F::carry(p.t1)); r.t2 = p.t2; return r; }

  // masked select: mask all-ones -> b
  static JJ_DEV ENiels select(const ENiels& a, const ENiels& b, u32 mask) {
    ENiels r; r.vpu = F::select(a.vpu, b.vpu, mask); r.vmu = F::select(a.vmu, b.vmu, mask); r.z = F::select(a.z, b.z, mask); r.t2d = F::select(a.t2d, b.t2d, mask); return r;
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
    const Fe l2 = F::sub(pp, F::mul(a.u, F::add(F::mul(F::konst(FqP::TP_A2), a.v), F::konst(FqP::TP_B2))));
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
typedef CurveT<Field<FqP, 0>> CurveNP;   // products without association pins (see JJ_MUL_PIN in jj_field.h)

}  // namespace jj
