#!/usr/bin/env python3
"""
Static bound verification for the lazy-reduction field arithmetic in jubjub_amd/csrc/jj_field.h and the
point formulas in jj_curve.h.

Every device value is modelled by (per-limb upper bounds, value upper bound).  The checker replays the
exact operation sequences of the device formulas and asserts that
  * no 64-bit column accumulator of a Montgomery product can overflow,
  * no limb of a biased subtraction can underflow and no 32-bit limb can overflow,
  * values stay inside the range for which the Montgomery output bound holds,
then iterates the ladder body to a fixed point so the invariants are inductive.

Run: python tools/bounds_check.py     (also imported by tests/test_bounds.py)
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_constants import Q, R as RMOD, LB, NL, MASK, MONT, limbs, bias


class FieldModel:
    def __init__(self, p):
        self.p = p
        self.P = limbs(p)
        self.BIAS_N = bias(p, 3, 30)
        self.BIAS_L = bias(p, 5, 31)

    # ---- bound objects: (limb_bounds[9], value_bound) ; all bounds inclusive maxima
    def N(self, val_mult=2.0):
        v = int(self.p * val_mult)
        return ([MASK] * (NL - 1) + [min(MASK, v >> (LB * (NL - 1)))], v)

    def const(self, x):
        return (limbs(x), x)

    def _reduce_check(self, cols, what):
        """cols: max column sums of the product part. Simulate worst-case reduce()."""
        c = list(cols) + [0]
        for k in range(NL):
            m = MASK
            c[k] += m * (self.P[0] if self.P[0] != 1 else 1)
            assert c[k] < (1 << 64), f"{what}: column {k} overflows: 2^{c[k].bit_length()}"
            c[k + 1] += c[k] >> LB
            for j in range(1, NL):
                c[k + j] += m * self.P[j]
        for k in range(NL, 2 * NL):
            assert c[k] < (1 << 64), f"{what}: column {k} overflows: 2^{c[k].bit_length()}"
            if k + 1 < 2 * NL:
                c[k + 1] += c[k] >> LB

    def mul(self, a, b, what="mul"):
        al, av = a
        bl, bv = b
        assert max(al) < (1 << 32) and max(bl) < (1 << 32), what
        cols = [0] * (2 * NL - 1)
        for i in range(NL):
            for j in range(NL):
                cols[i + j] += al[i] * bl[j]
        self._reduce_check(cols, what)
        # value: (a*b + m*p)/R with m < R  ->  < a*b/R + p
        val = (av * bv) // MONT + self.p
        assert val <= 2 * self.p, f"{what}: output value bound {val / self.p:.3f}p exceeds 2p"
        return ([MASK] * (NL - 1) + [min(MASK, val >> (LB * (NL - 1)))], val)

    def sqr(self, a, what="sqr"):
        al, av = a
        assert max(al) < (1 << 31), f"{what}: doubled limb overflows 32 bits"
        return self.mul(a, a, what)

    def add(self, a, b, what="add"):
        l = [x + y for x, y in zip(a[0], b[0])]
        assert max(l) < (1 << 32), what
        return (l, a[1] + b[1])

    def carry(self, a, what="carry"):
        l = a[0]
        out = [MASK] + [MASK + (l[i - 1] >> LB) for i in range(1, NL - 1)] + [l[NL - 1] + (l[NL - 2] >> LB)]
        out[0] = min(MASK, l[0])
        # top limb can also be bounded through the value
        out[NL - 1] = min(out[NL - 1], a[1] >> (LB * (NL - 1)))
        assert max(out) < (1 << 32), what
        return (out, a[1])

    def _sub(self, a, b, B, K, what):
        for i in range(NL):
            assert b[0][i] <= B[i], f"{what}: limb {i} of subtrahend (<= {b[0][i]:#x}) may exceed bias {B[i]:#x}"
        t = [a[0][i] + B[i] for i in range(NL)]
        assert max(t) < (1 << 32), f"{what}: limb overflow"
        return self.carry((t, a[1] + K * self.p), what)

    def sub(self, a, b, what="sub"):
        return self._sub(a, b, self.BIAS_N, 3, what)

    def sub_lazy(self, a, b, what="sub_lazy"):
        B = self.BIAS_N
        for i in range(NL):
            assert b[0][i] <= B[i], f"{what}: limb {i} of subtrahend may exceed bias"
        t = [a[0][i] + B[i] for i in range(NL)]
        assert max(t) < (1 << 32), f"{what}: limb overflow"
        return (t, a[1] + 3 * self.p)

    def sub_wide(self, a, b, what="sub_wide"):
        return self._sub(a, b, self.BIAS_L, 5, what)

    def neg(self, a, what="neg"):
        zero = ([0] * NL, 0)
        return self._sub(zero, a, self.BIAS_N, 3, what)

    def join(self, a, b):
        return ([max(x, y) for x, y in zip(a[0], b[0])], max(a[1], b[1]))

    def leq(self, a, b):
        return all(x <= y for x, y in zip(a[0], b[0])) and a[1] <= b[1]


def check_curve(verbose=True):
    F = FieldModel(Q)
    D2 = F.const(((2 * (-10240 * pow(10241, -1, Q))) % Q * MONT) % Q)

    def into_extended(cu, cv, cz, ct, w):
        return dict(u=F.mul(cu, ct, w + ".U"), v=F.mul(cv, cz, w + ".V"), z=F.mul(cz, ct, w + ".Z"), t1=cu, t2=cv)

    def dbl(p, w="dbl"):
        uu, vv, zz = F.sqr(p["u"], w + ".uu"), F.sqr(p["v"], w + ".vv"), F.sqr(p["z"], w + ".zz")
        uv2 = F.sqr(F.add(p["u"], p["v"]), w + ".uv2")
        vpu = F.add(vv, uu)
        vmu = F.sub(vv, uu, w + ".vmu")
        zz2 = F.add(zz, zz)
        cu = F.sub_lazy(uv2, vpu, w + ".cu")
        ct = F.sub_wide(zz2, vmu, w + ".ct")
        return into_extended(cu, vpu, vmu, ct, w)

    def add_niels(p, n, w="add", affine=False, negate=False):
        vmu, vpu = (n["vpu"], n["vmu"]) if negate else (n["vmu"], n["vpu"])
        a = F.mul(F.sub(p["v"], p["u"], w + ".v-u"), vmu, w + ".a")
        b = F.mul(F.add(p["v"], p["u"]), vpu, w + ".b")
        c = F.mul(F.mul(F.carry(p["t1"]), p["t2"], w + ".tt"), n["t2d"], w + ".c")
        if affine:
            d = F.add(p["z"], p["z"])
        else:
            zz = F.mul(p["z"], n["z"], w + ".zz")
            d = F.add(zz, zz)
        plus, minus = F.carry(F.add(d, c)), F.sub(d, c, w + ".d-c")
        cz, ct = (minus, plus) if negate else (plus, minus)
        return into_extended(F.sub_lazy(b, a, w + ".b-a"), F.add(b, a), cz, ct, w)

    def to_niels_ext(p, w="to_niels"):
        return dict(vpu=F.carry(F.add(p["v"], p["u"])), vmu=F.sub(p["v"], p["u"], w + ".vmu"), z=p["z"],
                    t2d=F.mul(F.mul(F.carry(p["t1"]), p["t2"], w + ".tt"), D2, w + ".t2d"))

    # Inputs: affine points loaded through from_words (mul by R2): N with value < 2p
    N2 = F.N(2.0)
    aff = dict(u=N2, v=N2, z=N2, t1=N2, t2=N2)
    niels_in = dict(vpu=F.carry(F.add(N2, N2)), vmu=F.sub(N2, N2), z=N2, t2d=N2)
    # table-entry negation (signed windows)
    neg_t2d = F.neg(niels_in["t2d"])
    niels_any = dict(vpu=F.join(niels_in["vpu"], niels_in["vmu"]), vmu=F.join(niels_in["vpu"], niels_in["vmu"]),
                     z=N2, t2d=F.join(niels_in["t2d"], neg_t2d))

    # fixed point over the accumulator invariant
    inv = aff
    for it in range(20):
        nxt = dict(inv)
        for cand in (dbl(inv), add_niels(inv, niels_any), add_niels(inv, niels_any, negate=True),
                     add_niels(inv, niels_any, affine=True), add_niels(inv, niels_any, affine=True, negate=True)):
            nxt = {k: F.join(nxt[k], cand[k]) for k in nxt}
        if all(F.leq(nxt[k], inv[k]) for k in inv):
            break
        inv = nxt
    else:
        raise AssertionError("accumulator invariant did not converge")
    # table entries built from accumulator-class points must be valid niels operands
    tn = to_niels_ext(inv)
    for k in ("vpu", "vmu", "z", "t2d"):
        assert F.leq(tn[k], niels_any[k]) or k == "z", (k, tn[k], niels_any[k])
    # z of a table entry is an accumulator z: N with value <= 2p  -> same class as N2
    assert F.leq(tn["z"], N2)
    if verbose:
        for k, v in inv.items():
            print(f"  acc.{k}: max limb 2^{max(v[0]).bit_length()}  value < {v[1] / Q:.3f} q")
    return inv


def check_field_misc(p, name, verbose=True):
    F = FieldModel(p)
    N2 = F.N(2.0)
    # from_words: unpack (limbs < 2^29, value < 2^256) * R2
    unp = ([MASK] * (NL - 1) + [(1 << 24) - 1], (1 << 256) - 1)
    r2 = F.const((MONT * MONT) % p)
    x = F.mul(unp, r2, name + ".from_words")
    wide = F.add(x, F.mul(unp, F.const(((1 << 256) * MONT * MONT) % p), name + ".from_wide_hi"))
    # elementwise kernels take from_words outputs (and sums of two) as inputs
    for a in (x, wide):
        for b in (x, wide):
            F.mul(a, b, name + ".mul")
        F.sqr(a, name + ".sqr")
        F.sub(a, x, name + ".sub")
    # to_words / canon: mul by 1 or ONE of anything up to 8p with limbs < 2^31
    big = ([(1 << 31) - 1] * (NL - 1) + [MASK], 8 * p)
    one = F.const(MONT % p)
    out = F.mul(big, one, name + ".canon")
    assert out[1] < 2 * p
    plain1 = ([1] + [0] * (NL - 1), 1)
    out = F.mul(big, plain1, name + ".to_words")
    assert out[1] <= p, "to_words needs value <= p before the conditional subtract"
    if verbose:
        print(f"  {name}: from_words value < {x[1] / p:.3f} p, wide < {wide[1] / p:.3f} p")


def main():
    print("curve formulas (Fq):")
    check_curve()
    print("field helpers:")
    check_field_misc(Q, "Fq")
    check_field_misc(RMOD, "Fr")
    print("all bounds hold")


if __name__ == "__main__":
    main()
