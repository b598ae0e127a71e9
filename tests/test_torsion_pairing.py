"""
The subgroup test the HIP path uses (Curve::is_torsion_free in jubjub_amd/csrc/jj_curve.h) is not the reference's
algorithm: the reference multiplies by r with the full ladder (src/lib.rs:709-711), the device evaluates the order-8
Tate pairing against a fixed generator of the 8-torsion.  This file checks, on the CPU, that the two predicates agree:
the formula is restated here with Python integers from the constants tools/gen_constants.py derives, and compared with
the pinned oracle's ladder on every coset of the prime-order subgroup, on the eight small-order points of the
reference's EIGHT_TORSION vector (src/lib.rs:1589-1677) and on random points of the full group.
"""
import os
import random
import sys

import pytest

from oracle import jubjub_ref as J

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import gen_constants as G  # noqa: E402

Q = J.Q


@pytest.fixture(scope="module")
def tp():
    return G.torsion_pairing_constants(J.EDWARDS_D)


def pairing_torsion_free(P, c):
    """Same operation sequence as the device function."""
    u, v = P
    if u == 0 and v == 1:
        return True
    pp, mm = (1 + v) % Q, (1 - v) % Q
    l1 = (pp - u * (c["TP_A1"] * v + c["TP_B1"])) % Q
    l2 = (pp - u * (c["TP_A2"] * v + c["TP_B2"])) % Q
    g = l1 * u * v % Q
    k = u * u * pp * mm % Q
    z = c["TP_C"] * pow(g, 4, Q) * l2 * l2 * pow(k, 7, Q) % Q
    t = (Q - 1) >> 32
    w = pow(z, (t - 1) // 2, Q)
    b = z * w * w % Q
    return pow(b, 1 << 29, Q) == 1


def fast_add(p, q):
    return J.affine_add_fast(p, q)


def test_header_constants_match_generator(tp):
    hdr = open(os.path.join(os.path.dirname(G.__file__), "..", "jubjub_amd", "csrc", "jj_constants.h")).read()
    for name, val in tp.items():
        want = ", ".join("0x%08xu" % x for x in G.limbs(val * G.MONT % Q))
        assert "%s[9] = {%s}" % (name, want) in hdr, name


def test_small_order_points(golden, tp):
    pts = [(sum(int(h, 16) << (64 * i) for i, h in enumerate(p["u"])), sum(int(h, 16) << (64 * i) for i, h in enumerate(p["v"])))
           for p in golden["EIGHT_TORSION_raw"]["points"]]
    assert len(pts) == 8
    for p in pts:
        assert pairing_torsion_free(p, tp) == bool(J.ext_is_torsion_free(J.affine_to_extended(p)))
    assert sum(pairing_torsion_free(p, tp) for p in pts) == 1        # the identity only


def test_every_coset_of_the_subgroup(golden, tp):
    rng = random.Random(20260928)
    tors = [(sum(int(h, 16) << (64 * i) for i, h in enumerate(p["u"])), sum(int(h, 16) << (64 * i) for i, h in enumerate(p["v"])))
            for p in golden["EIGHT_TORSION_raw"]["points"]]
    for _ in range(40):
        s = J.scalar_mul_fast(J.GENERATOR, 8 * rng.randrange(1, J.R_MOD))     # prime-order point
        for t in tors:
            p = fast_add(s, t)
            assert pairing_torsion_free(p, tp) == (t == J.AFFINE_IDENTITY)


def test_random_full_group_points_against_ladder(tp):
    rng = random.Random(7)
    hits = 0
    for _ in range(48):
        p = J.scalar_mul_fast(J.GENERATOR, rng.randrange(1, 8 * J.R_MOD))
        want = bool(J.ext_is_torsion_free(J.affine_to_extended(p)))       # the reference's definition
        assert pairing_torsion_free(p, tp) == want
        hits += want
    p = J.scalar_mul_fast(J.GENERATOR, 8 * 12345)
    assert pairing_torsion_free(p, tp) and J.ext_is_torsion_free(J.affine_to_extended(p))
