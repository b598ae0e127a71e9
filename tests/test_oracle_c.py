"""
Pins the C oracle (oracle/jubjub_oracle.c via oracle/c_oracle.py) against the reference's golden
vectors and cross-checks it against the Python big-int oracle on seeded random inputs.
"""
import os
import random

import numpy as np
import pytest

from conftest import limbs
from oracle import c_oracle as O
from oracle import jubjub_ref as J

Q, R = J.Q, J.R_MOD


def b32(x):
    return np.frombuffer(int(x).to_bytes(32, "little"), dtype=np.uint8)


def arr32(xs):
    return np.stack([b32(x) for x in xs]) if len(xs) else np.zeros((0, 32), np.uint8)


def pt64(p):
    return np.concatenate([b32(p[0]), b32(p[1])])


def arr64(ps):
    return np.stack([pt64(p) for p in ps]) if len(ps) else np.zeros((0, 64), np.uint8)


def to_int(row):
    return int.from_bytes(bytes(row), "little")


def to_pt(row):
    return (to_int(row[:32]), to_int(row[32:]))


def rand_points(rng, n, subgroup=False):
    """Random curve points: k*G for random k (G generates the full group of order 8r)."""
    pts = []
    for _ in range(n):
        k = rng.randrange(1, 8 * R)
        p = J.scalar_mul_fast(J.GENERATOR, k)
        if subgroup:
            p = J.ext_to_affine(J.ext_mul_by_cofactor(J.affine_to_extended(p)))
        pts.append(p)
    return pts


def test_selftest():
    assert O.lib().jjo_selftest() == 0


@pytest.mark.parametrize("which,p,F", [(O.FQ, Q, J.FQ), (O.FR, R, J.FR)])
def test_field_ops_vs_python(which, p, F):
    rng = random.Random(1234 + which)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << 255) % p, F.R, F.R2 % p]
    a = edge + [rng.randrange(p) for _ in range(300)]
    b = list(reversed(edge)) + [rng.randrange(p) for _ in range(300)]
    A, B = arr32(a), arr32(b)
    for op, fn in [("add", F.add), ("sub", F.sub), ("mul", F.mul)]:
        out, _ = O.field_op(which, op, A, B)
        assert [to_int(r) for r in out] == [fn(x, y) for x, y in zip(a, b)], op
    for op, fn in [("neg", F.neg), ("square", F.square), ("double", F.double)]:
        out, _ = O.field_op(which, op, A)
        assert [to_int(r) for r in out] == [fn(x) for x in a], op
    out, ok = O.field_op(which, "invert", A)
    for x, r, k in zip(a, out, ok):
        assert (to_int(r), int(k)) == F.invert(x)
    out, ok = O.field_op(which, "sqrt", A)
    sq = J.fq_sqrt if which == O.FQ else J.fr_sqrt
    for x, r, k in zip(a, out, ok):
        s, good = sq(x)
        assert int(k) == good
        if good:
            assert to_int(r) == s
        else:
            assert to_int(r) == 0


def test_mont_limb_vectors(golden):
    t = golden["fr_mul_consistency_mont"]
    a, b, c = ([int(x, 16) for x in t[k]] for k in "abc")
    assert O.mont_mul(O.FR, a, b) == c  # lib.rs:1776
    w = golden["fr"]["from_bytes_wide"]
    out = O.from_bytes_wide(O.FR, np.full((1, 64), 0xFF, np.uint8))
    assert O.to_mont(O.FR, out[0]) == [int(x, 16) for x in w["max_output_mont"]]  # fr.rs:1024-1034
    tb = golden["fr"]["to_bytes"]
    assert O.from_mont(O.FR, J.int_to_limbs(J.FR.R2)) == bytes(tb["R2"])
    assert O.from_mont(O.FR, J.int_to_limbs(J.FR.R)) == bytes(tb["one"])
    for bad in golden["fr"]["from_bytes_invalid"]["cases"]:
        assert O.from_bytes(O.FR, np.array(bad, np.uint8))[1][0] == 0
    assert O.from_bytes(O.FR, np.array(tb["neg_one"], np.uint8))[1][0] == 1


def test_from_bytes_wide_vs_python():
    rng = random.Random(7)
    raw = np.frombuffer(rng.randbytes(64 * 200), dtype=np.uint8).reshape(200, 64)
    for which, F in ((O.FQ, J.FQ), (O.FR, J.FR)):
        out = O.from_bytes_wide(which, raw)
        assert [to_int(r) for r in out] == [F.from_bytes_wide(bytes(r)) for r in raw]


def test_serialization_and_zip216(golden):
    encs = np.array(golden["serialization_16"]["encodings"], np.uint8)
    gen8 = J.ext_to_affine(J.ext_mul_by_cofactor(J.affine_to_extended(J.GENERATOR)))
    pts = [J.scalar_mul_fast(gen8, k) for k in range(1, 17)]
    assert (O.compress(arr64(pts)) == encs).all()
    out, ok = O.decompress(encs)
    assert ok.all() and [to_pt(r) for r in out] == pts
    out, ok = O.batch_from_bytes(encs)
    assert ok.all() and [to_pt(r) for r in out] == pts
    z = np.array(golden["zip216_noncanonical"]["encodings"], np.uint8)
    assert not O.decompress(z)[1].any() and not O.batch_from_bytes(z)[1].any()
    out, ok = O.decompress(z, flags=0)
    assert ok.all()
    re = O.compress(out)
    assert (re != z).any(axis=1).all()
    re[:, 31] |= 0x80
    assert (re == z).all()


def test_eight_torsion_ladder(golden):
    tors = [(limbs(p["u"]), limbs(p["v"])) for p in golden["EIGHT_TORSION_raw"]["points"]]
    g = O.varbase_mul(np.array(golden["FR_MODULUS_BYTES"]["bytes"], np.uint8).reshape(1, 32), arr64([J.GENERATOR]))
    cur = g.copy()
    for t in tors:
        assert to_pt(cur[0]) == t
        cur = O.point_op("add", cur, g)
    T = arr64(tors)
    assert O.predicate("is_small_order", T).all()
    assert O.predicate("is_identity", O.point_op("mul_by_cofactor", T)).all()
    assert O.predicate("is_on_curve", T).all()
    assert list(O.predicate("is_torsion_free", T)) == [0] * 7 + [1]
    assert list(O.predicate("is_prime_order", T)) == [0] * 8
    assert list(O.predicate("is_torsion_free", arr64([J.GENERATOR]))) == [0]


def test_ladders_vs_python():
    rng = random.Random(99)
    pts = rand_points(rng, 6)
    ks = [0, 1, R - 1, R, (1 << 252) - 1] + [rng.randrange(R)]
    S = arr32(ks)
    ext = O.varbase_mul_ext(S, arr64(pts))
    aff = O.varbase_mul(S, arr64(pts))
    for p, k, e, a in zip(pts, ks, ext, aff):
        want = J.ext_multiply(J.affine_to_extended(p), k.to_bytes(32, "little"))
        got = tuple(to_int(e[32 * i : 32 * i + 32]) for i in range(5))
        assert got == want  # projective coordinates match the exact reference ladder
        assert to_pt(a) == J.ext_to_affine(want)
    base = pts[0]
    fb = O.fixedbase_mul(S, pt64(base))
    for k, a in zip(ks, fb):
        want = J.affine_niels_multiply(J.affine_to_niels(base), k.to_bytes(32, "little"))
        assert to_pt(a) == J.ext_to_affine(want)
    # scalars with the top four bits set are read as their low 252 bits (lib.rs:281-288)
    hi = (0xF << 252) | ks[5]
    a = O.varbase_mul(arr32([hi]), arr64([pts[1]]))
    b = O.varbase_mul(arr32([ks[5]]), arr64([pts[1]]))
    assert (a == b).all()


def test_point_ops_vs_python():
    rng = random.Random(5)
    P, Qs = rand_points(rng, 8), rand_points(rng, 8)
    P[0], Qs[1] = J.AFFINE_IDENTITY, J.AFFINE_IDENTITY
    Qs[2] = P[2]
    Qs[3] = J.affine_neg(P[3])
    A, B = arr64(P), arr64(Qs)
    ext = [J.affine_to_extended(p) for p in P]
    assert [to_pt(r) for r in O.point_op("double", A)] == [J.ext_to_affine(J.ext_double(e)) for e in ext]
    assert [to_pt(r) for r in O.point_op("add", A, B)] == [J.ext_to_affine(J.ext_add_affine(e, q)) for e, q in zip(ext, Qs)]
    assert [to_pt(r) for r in O.point_op("sub", A, B)] == [J.ext_to_affine(J.ext_sub_affine(e, q)) for e, q in zip(ext, Qs)]
    assert [to_pt(r) for r in O.point_op("neg", A)] == [J.affine_neg(p) for p in P]
    assert [to_pt(r) for r in O.point_op("mul_by_cofactor", A)] == [J.ext_to_affine(J.ext_mul_by_cofactor(e)) for e in ext]
    n = O.to_niels(A)
    for row, p in zip(n, P):
        assert tuple(to_int(row[32 * i : 32 * i + 32]) for i in range(3)) == J.affine_to_niels(p)


def test_batch_normalize_and_msm():
    rng = random.Random(11)
    pts = rand_points(rng, 5)
    ext = []
    for p in pts:
        e = J.affine_to_extended(p)
        for _ in range(rng.randrange(1, 4)):
            e = J.ext_double(e)
        ext.append(e)
    raw = np.stack([np.concatenate([b32(c) for c in e]) for e in ext])
    out = O.batch_normalize(raw)
    assert [to_pt(r) for r in out] == [J.ext_to_affine(e) for e in ext]
    ks = [rng.randrange(R) for _ in pts]
    got = O.msm(arr32(ks), arr64(pts))
    want = J.ext_to_affine(J.msm([k.to_bytes(32, "little") for k in ks], pts))
    assert to_pt(got) == want
    assert to_pt(O.point_sum(arr64(pts))) == J.ext_to_affine(J.ext_sum([J.affine_to_extended(p) for p in pts]))
    # empty inputs
    assert to_pt(O.msm(np.zeros((0, 32), np.uint8), np.zeros((0, 64), np.uint8))) == J.AFFINE_IDENTITY
    # the bucket method (the CPU baseline of the MSM workload) gives the same group element as the fold of ladders, for every window
    # width, with full-width scalars (top four bits ignored like the ladder), torsion points and the identity among the terms
    import util

    n = 700
    S = util.rand_scalars(77, n, full_width=True)
    P = util.rand_points(78, n)
    P[:8] = util.arr64([J.AFFINE_IDENTITY] * 8)
    S[8:12] = 0
    want = O.msm(S, P)
    for c in (1, 3, 8, 13):
        assert (O.msm_pippenger(S, P, c) == want).all(), c
    assert to_pt(O.msm_pippenger(S[:0], P[:0], 13)) == J.AFFINE_IDENTITY


def test_decompress_flags_vs_python():
    rng = random.Random(3)
    encs = []
    for p in rand_points(rng, 6):
        encs.append(J.affine_to_bytes(p))
    encs += [rng.randbytes(32) for _ in range(40)]
    encs.append((Q).to_bytes(32, "little"))  # v == q rejected
    encs.append((Q - 1).to_bytes(32, "little"))  # (0,-1)
    E = np.frombuffer(b"".join(encs), np.uint8).reshape(-1, 32)
    for flags in (0, 1, 1 | 2, 1 | 4, 1 | 8, 1 | 2 | 4 | 8):
        out, ok = O.decompress(E, flags)
        for e, r, k in zip(encs, out, ok):
            p, good = J.affine_from_bytes(e, zip216=bool(flags & 1))
            if good:
                ep = J.affine_to_extended(p)
                if (flags & 2) and not J.ext_is_torsion_free(ep):
                    good = 0
                if (flags & 4) and J.ext_is_small_order(ep):
                    good = 0
                if good and (flags & 8):
                    p = J.ext_to_affine(J.ext_mul_by_cofactor(ep))
            assert int(k) == good
            assert to_pt(r) == (p if good else (0, 0))
    out, ok = O.batch_from_bytes(E)
    o2, k2 = O.decompress(E, 1)
    assert (out == o2).all() and (ok == k2).all()


def test_committed_oracle_vectors():
    """The committed fixtures (tests/golden/oracle_vectors.json) agree with the C oracle."""
    import json, os
    v = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.json")))
    fx = lambda h: np.frombuffer(bytes.fromhex(h), dtype=np.uint8)
    S = np.stack([fx(c["scalar"]) for c in v["varbase"]])
    P = np.stack([fx(c["point"]) for c in v["varbase"]])
    assert (O.varbase_mul(S, P) == np.stack([fx(c["out"]) for c in v["varbase"]])).all()
    assert (O.varbase_mul_ext(S, P) == np.stack([fx(c["ext"]) for c in v["varbase"]])).all()
    fb = v["fixedbase"]
    S = np.stack([fx(c["scalar"]) for c in fb["cases"]])
    assert (O.fixedbase_mul(S, fx(fb["base"])) == np.stack([fx(c["out"]) for c in fb["cases"]])).all()
    E = np.stack([fx(c["in"]) for c in v["decompress"]])
    for flags in (0, 1, 3, 5, 9, 15):
        out, ok = O.decompress(E, flags)
        assert list(ok) == [c["f%d" % flags]["ok"] for c in v["decompress"]]
        assert (out == np.stack([fx(c["f%d" % flags]["out"]) for c in v["decompress"]])).all()
    for m in v["msm"]:
        got = O.msm(np.stack([fx(s) for s in m["scalars"]]), np.stack([fx(p) for p in m["points"]]))
        assert (got == fx(m["out"])).all()


def test_config0_cpu_script_runs():
    """tests/config1_cpu.py (BASELINE.json configs[0] on the CPU port: 1k Fq multiplications + 1k ExtendedPoint::double, the shape
    of benches/fq_bench.rs:25-33 and point_bench.rs:6-11) runs and prints two positive ns/op figures"""
    import re
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "config1_cpu.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    vals = [float(v) for v in re.findall(r"([0-9.]+) ns/op", r.stdout)]
    assert len(vals) == 2 and all(v > 0 for v in vals), r.stdout
