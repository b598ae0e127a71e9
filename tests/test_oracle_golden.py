"""
Pins the Python oracle (oracle/jubjub_ref.py) against every known-answer vector the
reference's own tests hold (transcribed to tests/golden/reference_vectors.json), by
replaying the reference tests src/lib.rs:1456-1935 and src/fr.rs:787-1244.
"""
import pytest

from conftest import limbs
from oracle import jubjub_ref as J

Q, R = J.Q, J.R_MOD
FQ, FR = J.FQ, J.FR


def raw_point(g):
    return (limbs(g["u"]) % Q, limbs(g["v"]) % Q)


# ---------------------------------------------------------------- constants


def test_moduli_match_readme_and_evidence(golden):
    assert "0x%064x" % R in golden["readme_hex"]["values"]
    assert "0x%064x" % Q in golden["readme_hex"]["values"]
    ev = golden["evidence"]
    assert int(ev["p"]) == Q and int(ev["l"]) == R
    assert int(ev["d"]) == J.EDWARDS_D and int(ev["a"]) == -1
    # both evidence base points are on the curve
    for x, y in (("x0", "y0"), ("x1", "y1")):
        assert J.affine_is_on_curve((int(ev[x]), int(ev[y])))


def test_curve_constants(golden):
    assert limbs(golden["EDWARDS_D_raw"]["limbs"]) == J.EDWARDS_D  # lib.rs:399-404
    assert limbs(golden["EDWARDS_D2_raw"]["limbs"]) == J.EDWARDS_D2  # lib.rs:407-412
    assert raw_point(golden["GENERATOR_raw"]) == J.GENERATOR
    assert raw_point(golden["FULL_GENERATOR_raw"]) == J.GENERATOR
    assert bytes(golden["FR_MODULUS_BYTES"]["bytes"]) == J.FR_MODULUS_BYTES
    assert J.affine_is_on_curve(J.GENERATOR)


def test_fr_constants(golden):
    f = golden["fr"]
    assert limbs(f["MODULUS"]["limbs"]) == R
    m32 = [int(x, 16) for x in f["MODULUS_LIMBS_32"]["limbs"]]
    assert sum(v << (32 * i) for i, v in enumerate(m32)) == R
    assert R.bit_length() == f["MODULUS_BITS"]["value"]
    assert int(f["INV"]["value"], 16) == FR.INV
    assert limbs(f["R_mont"]["limbs"]) == FR.R
    assert limbs(f["R2_mont"]["limbs"]) == FR.R2
    assert limbs(f["R3_mont"]["limbs"]) == FR.R3
    assert FR.from_mont_limbs([int(x, 16) for x in f["TWO_INV_mont"]["limbs"]]) == pow(2, -1, R)
    gen = FR.from_mont_limbs([int(x, 16) for x in f["GENERATOR_mont"]["limbs"]])
    assert gen == 6
    s = f["S"]["value"]
    t = (R - 1) >> s
    assert t & 1 == 1
    rou = FR.from_mont_limbs([int(x, 16) for x in f["ROOT_OF_UNITY_mont"]["limbs"]])
    assert rou == pow(gen, t, R) == R - 1
    delta = FR.from_mont_limbs([int(x, 16) for x in f["DELTA_mont"]["limbs"]])
    assert delta == pow(gen, 1 << s, R)
    assert limbs(f["DELTA_T_EXP"]["limbs"]) == t and pow(delta, t, R) == 1  # fr.rs:801-810
    assert limbs(f["SQRT_EXP"]["limbs"]) == (R + 1) // 4
    assert limbs(f["R_MINUS_2"]["limbs"]) == R - 2
    assert limbs(f["LARGEST_mont"]["limbs"]) == R - 1


def test_fr_inv_derivation():
    # fr.rs:813-826
    inv = 1
    for _ in range(63):
        inv = (inv * inv) % (1 << 64)
        inv = (inv * (R & ((1 << 64) - 1))) % (1 << 64)
    inv = (-inv) % (1 << 64)
    assert inv == FR.INV


# ---------------------------------------------------------------- Fr byte vectors


def test_fr_to_from_bytes(golden):
    tb = golden["fr"]["to_bytes"]
    assert FR.to_bytes(0) == bytes(tb["zero"])
    assert FR.to_bytes(1) == bytes(tb["one"])
    r2_elem = FR.from_mont_limbs(J.int_to_limbs(FR.R2))  # the element whose Montgomery limbs are R2
    assert FR.to_bytes(r2_elem) == bytes(tb["R2"])
    assert FR.to_bytes(FR.neg(1)) == bytes(tb["neg_one"])
    # fr.rs:891-961
    assert FR.from_bytes(bytes(tb["zero"])) == (0, 1)
    assert FR.from_bytes(bytes(tb["one"])) == (1, 1)
    assert FR.from_bytes(bytes(tb["R2"])) == (r2_elem, 1)
    assert FR.from_bytes(bytes(tb["neg_one"]))[1] == 1
    for bad in golden["fr"]["from_bytes_invalid"]["cases"]:
        assert FR.from_bytes(bytes(bad))[1] == 0
    assert golden["fr"]["debug_R2"]["hex"] == "0x%064x" % r2_elem


def test_fr_from_bytes_wide(golden):
    w = golden["fr"]["from_bytes_wide"]
    r2_elem = FR.from_mont_limbs(J.int_to_limbs(FR.R2))
    assert FR.from_bytes_wide(bytes(w["r2_input"])) == r2_elem
    assert FR.from_bytes_wide(bytes(w["neg_one_input"])) == R - 1
    mx = FR.from_bytes_wide(b"\xff" * 64)
    assert FR.to_mont_limbs(mx) == [int(x, 16) for x in w["max_output_mont"]]
    # from_u512 tests fr.rs:963-997
    assert FR.from_bytes_wide(R.to_bytes(32, "little") + bytes(32)) == 0
    assert FR.to_mont_limbs(FR.from_bytes_wide((1).to_bytes(64, "little"))) == J.int_to_limbs(FR.R)
    assert FR.to_mont_limbs(FR.from_bytes_wide((1 << 256).to_bytes(64, "little"))) == J.int_to_limbs(FR.R2)
    r3 = FR.from_mont_limbs(J.int_to_limbs(FR.R3))
    one = FR.from_mont_limbs(J.int_to_limbs(FR.R))
    assert mx == FR.sub(r3, one)


def test_fr_arith_vectors(golden):
    f = golden["fr"]
    M = lambda key: FR.from_mont_limbs([int(x, 16) for x in key])
    largest = M(f["LARGEST_mont"]["limbs"])
    assert FR.add(largest, largest) == M(f["test_addition"]["largest_plus_largest_mont"])
    assert FR.add(largest, M(["0x1", "0x0", "0x0", "0x0"])) == 0
    assert FR.neg(largest) == M(["0x1", "0x0", "0x0", "0x0"])
    assert FR.neg(0) == 0
    # from_raw fr.rs:1230-1244
    assert FR.to_mont_limbs(FR.from_raw([0xFFFFFFFFFFFFFFFF] * 4)) == FR.to_mont_limbs(
        FR.from_raw([int(x, 16) for x in f["test_from_raw"]["expect_mont_of_all_ones"]])
    )
    assert FR.from_raw(J.int_to_limbs(R)) == 0
    assert FR.to_mont_limbs(FR.from_raw([1, 0, 0, 0])) == J.int_to_limbs(FR.R)


def test_fr_mul_consistency_limbs(golden):
    t = golden["fr_mul_consistency_mont"]
    a, b, c = (FR.from_mont_limbs([int(x, 16) for x in t[k]]) for k in "abc")
    assert FR.mul(a, b) == c  # lib.rs:1776


def test_fr_sqrt_count(golden):
    t = golden["fr"]["test_sqrt"]
    sq = FR.from_mont_limbs([int(x, 16) for x in t["start_mont"]])
    one = 1
    none = 0
    for _ in range(t["iters"]):
        s, ok = J.fr_sqrt(sq)
        if not ok:
            none += 1
        else:
            assert FR.mul(s, s) == sq
        sq = FR.sub(sq, one)
    assert none == t["none_count"]


def test_fr_inversion():
    assert FR.invert(0) == (0, 0)
    assert FR.invert(1) == (1, 1)
    assert FR.invert(R - 1) == (R - 1, 1)


# ---------------------------------------------------------------- curve tests


def test_d_is_non_quadratic_residue():
    # lib.rs:1462-1466
    assert J.fq_sqrt(J.EDWARDS_D)[1] == 0
    assert J.fq_sqrt(FQ.neg(J.EDWARDS_D))[1] == 0
    assert J.fq_sqrt(FQ.invert(FQ.neg(J.EDWARDS_D))[0])[1] == 0


def test_niels_identities():
    assert J.affine_to_niels(J.AFFINE_IDENTITY) == J.AFFINE_NIELS_IDENTITY
    assert J.ext_to_niels(J.EXT_IDENTITY) == J.EXT_NIELS_IDENTITY
    assert J.affine_is_on_curve(J.AFFINE_IDENTITY)


def test_assoc(golden):
    p = J.ext_mul_by_cofactor(J.affine_to_extended(raw_point(golden["TEST_POINT_raw"])))
    assert J.ext_is_on_curve(p)
    a, b = golden["test_assoc_scalars"]["a"], golden["test_assoc_scalars"]["b"]
    lhs = J.ext_mul_scalar(J.ext_mul_scalar(p, a), b)
    rhs = J.ext_mul_scalar(p, FR.mul(a, b))
    assert J.ext_eq(lhs, rhs)


def test_batch_normalize(golden):
    p = J.ext_mul_by_cofactor(J.affine_to_extended(raw_point(golden["TEST_POINT_raw"])))
    v = []
    for _ in range(10):
        v.append(p)
        p = J.ext_double(p)
    expected = [J.ext_to_affine(x) for x in v]
    norm, aff = J.batch_normalize(v)
    assert aff == expected
    norm2, aff2 = J.batch_normalize(norm)
    assert aff2 == expected and all(J.ext_is_on_curve(x) for x in norm2)


def test_find_eight_torsion(golden):
    g = J.affine_to_extended(J.GENERATOR)
    assert not J.ext_is_small_order(g)
    g = J.ext_multiply(g, J.FR_MODULUS_BYTES)
    assert J.ext_is_small_order(g)
    cur = g
    for pt in golden["EIGHT_TORSION_raw"]["points"]:
        assert J.ext_to_affine(cur) == raw_point(pt)
        cur = J.ext_add(cur, g)


def test_find_curve_generator():
    trial = bytearray(32)
    for _ in range(255):
        a, ok = J.affine_from_bytes(trial)
        if ok:
            assert J.affine_is_on_curve(a)
            b = J.ext_multiply(J.affine_to_extended(a), J.FR_MODULUS_BYTES)
            assert J.ext_is_small_order(b)
            b = J.ext_double(b)
            assert J.ext_is_small_order(b)
            b = J.ext_double(b)
            assert J.ext_is_small_order(b)
            if not J.ext_is_identity(b):
                b = J.ext_double(b)
                assert J.ext_is_small_order(b) and J.ext_is_identity(b)
                assert a == J.GENERATOR
                assert J.ext_is_torsion_free(J.ext_mul_by_cofactor(J.affine_to_extended(a)))
                return
        trial[0] += 1
    pytest.fail("should have found a generator of the curve")


def test_small_order_and_identity(golden):
    tors = [raw_point(p) for p in golden["EIGHT_TORSION_raw"]["points"]]
    for t in tors:
        assert J.ext_is_small_order(J.affine_to_extended(t))
        assert J.ext_is_identity(J.ext_mul_by_cofactor(J.affine_to_extended(t)))
    # lib.rs:1738-1749: internal projective coordinates
    a = J.ext_mul_by_cofactor(J.affine_to_extended(tors[0]))
    b = J.ext_mul_by_cofactor(J.affine_to_extended(tors[1]))
    assert a[0] == b[0]
    assert a[1] == a[2] and b[1] == b[2]
    assert a[1] != b[1] and a[2] != b[2]


def test_mul_consistency(golden):
    t = golden["fr_mul_consistency_mont"]
    a, b, c = (FR.from_mont_limbs([int(x, 16) for x in t[k]]) for k in "abc")
    p = J.ext_mul_by_cofactor(J.affine_to_extended(raw_point(golden["TEST_POINT_raw"])))
    pc = J.ext_mul_scalar(p, c)
    pab = J.ext_mul_scalar(J.ext_mul_scalar(p, a), b)
    assert J.ext_eq(pc, pab)
    n = J.ext_to_niels(p)
    nm = lambda n_, k: J.ext_niels_multiply(n_, FR.to_bytes(k))
    assert J.ext_eq(pc, J.ext_mul_scalar(nm(n, a), b))
    assert J.ext_eq(nm(n, c), pab)
    assert J.ext_eq(nm(n, c), J.ext_mul_scalar(nm(n, a), b))
    an = J.affine_to_niels(J.ext_to_affine(p))
    am = lambda k: J.affine_niels_multiply(an, FR.to_bytes(k))
    assert J.ext_eq(pc, J.ext_mul_scalar(am(a), b))
    assert J.ext_eq(am(c), pab)
    assert J.ext_eq(am(c), J.ext_mul_scalar(am(a), b))


def test_serialization_consistency(golden):
    gen = J.ext_mul_by_cofactor(J.affine_to_extended(J.GENERATOR))
    p = gen
    encs = [bytes(e) for e in golden["serialization_16"]["encodings"]]
    batched = J.batch_from_bytes(encs)
    for enc, (bpt, bok) in zip(encs, batched):
        assert J.ext_is_on_curve(p)
        aff = J.ext_to_affine(p)
        ser = J.affine_to_bytes(aff)
        de, ok = J.affine_from_bytes(ser)
        assert ok and de == aff
        assert bok and bpt == aff
        assert ser == enc
        p = J.ext_add(p, gen)


def test_zip_216(golden):
    for enc in golden["zip216_noncanonical"]["encodings"]:
        b = bytearray(enc)
        assert J.affine_from_bytes(b)[1] == 0
        assert J.batch_from_bytes([b])[0][1] == 0
        c = bytearray(b)
        c[31] &= 0x7F
        assert J.affine_from_bytes(c)[1] == 1
        parsed, ok = J.affine_from_bytes(b, zip216=False)
        assert ok
        e = bytearray(J.affine_to_bytes(parsed))
        assert bytes(e) != bytes(b)
        e[31] |= 0x80
        assert bytes(e) == bytes(b)


def test_wnaf_recommendations(golden):
    tab = golden["wnaf_recommendations"]["table"]
    for i, r in enumerate(tab):
        assert J.recommended_wnaf_for_num_scalars(r) == 4 + i
        assert J.recommended_wnaf_for_num_scalars(r + 1) == 5 + i
