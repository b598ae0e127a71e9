"""
GPU parity tests: every entry point of the C ABI (through jubjub_amd.Engine) against the pinned oracle on the
same seeded inputs, bit-exact, plus the reference's own known-answer vectors (tests/golden/reference_vectors.json).
Run with:  python -m pytest tests -m gpu
"""
import numpy as np
import pytest

from oracle import c_oracle as O
from oracle import jubjub_ref as J
from util import (EDGE_SCALARS, Q, R, arr32, arr64, b32, pt64, rand_points, rand_scalars, to_int, to_pt, torsion_points)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from jubjub_amd import Engine

    e = Engine(0)
    yield e
    e.close()


def field_inputs(p, seed, n=1000):
    rng = np.random.default_rng(seed)
    edge = [0, 1, 2, p - 1, p - 2, p, p + 1, (p - 1) // 2, (1 << 255) - 1, (1 << 256) - 1, 1 << 255, 3 * p]
    edge = [e for e in edge if e < (1 << 256)]
    a = np.concatenate([arr32(edge), rng.integers(0, 256, size=(n, 32), dtype=np.uint8)])
    b = np.concatenate([arr32(list(reversed(edge))), rng.integers(0, 256, size=(n, 32), dtype=np.uint8)])
    return a, b


@pytest.mark.parametrize("fname,which,p", [("fq", O.FQ, Q), ("fr", O.FR, R)])
def test_field_ops(eng, fname, which, p):
    a, b = field_inputs(p, 11 + which)
    for op in ("add", "sub", "mul"):
        assert (eng.field_binary(fname, op, a, b) == O.field_op(which, op, a, b)[0]).all(), op
    for op in ("neg", "square", "double"):
        assert (eng.field_unary(fname, op, a) == O.field_op(which, op, a)[0]).all(), op
    out, ok = eng.field_unary_ok(fname, "invert", a)
    eo, ek = O.field_op(which, "invert", a)
    assert (ok == ek).all() and (out == eo).all()
    out, ok = eng.field_unary_ok(fname, "from_bytes", a)
    eo, ek = O.from_bytes(which, a)
    assert (ok == ek).all() and (out == eo).all()
    wide = np.concatenate([a, b], axis=1)
    assert (eng.from_bytes_wide(fname, wide) == O.from_bytes_wide(which, wide)).all()
    # empty batch
    assert eng.field_binary(fname, "mul", a[:0], b[:0]).shape == (0, 32)


@pytest.mark.parametrize("fname,which,p", [("fq", O.FQ, Q), ("fr", O.FR, R)])
def test_field_sqrt(eng, fname, which, p):
    a, _ = field_inputs(p, 23 + which, n=300)
    sq = O.field_op(which, "square", a)[0]
    x = np.concatenate([a, sq])
    out, ok = eng.field_unary_ok(fname, "sqrt", x)
    eo, ek = O.field_op(which, "sqrt", x)
    assert (ok == ek).all()
    assert (out == eo).all()
    assert ok[len(a):].all()


@pytest.mark.parametrize("fname,p", [("fq", Q), ("fr", R)])
def test_field_pow(eng, fname, p):
    rng = np.random.default_rng(55)
    a = rng.integers(0, 256, size=(300, 32), dtype=np.uint8)
    e = rng.integers(0, 256, size=(300, 32), dtype=np.uint8)
    e[0] = 0
    e[1] = b32(p - 2)                       # inversion exponent (reference test_invert_is_pow, src/fr.rs:1177-1202)
    e[2] = b32(1)
    a[3] = 0
    got = eng.field_binary(fname, "pow", a, e)
    want = [pow(to_int(x) % p, to_int(y), p) for x, y in zip(a, e)]
    assert [to_int(r) for r in got] == want
    inv, ok = eng.field_unary_ok(fname, "invert", a[1:2])
    assert (inv == got[1:2]).all() and ok.all()


def test_fr_golden_vectors(eng, golden):
    # reference src/fr.rs:856-961 and src/lib.rs:1758-1776 through the GPU path
    tb = golden["fr"]["to_bytes"]
    neg1 = eng.field_unary("fr", "neg", arr32([1]))
    assert bytes(neg1[0]) == bytes(tb["neg_one"])
    bad = np.array(golden["fr"]["from_bytes_invalid"]["cases"], np.uint8)
    assert not eng.field_unary_ok("fr", "from_bytes", bad)[1].any()
    assert eng.field_unary_ok("fr", "from_bytes", np.array([tb["neg_one"]], np.uint8))[1].all()
    t = golden["fr_mul_consistency_mont"]
    a, b, c = (J.FR.from_mont_limbs([int(x, 16) for x in t[k]]) for k in "abc")
    assert to_int(eng.field_binary("fr", "mul", arr32([a]), arr32([b]))[0]) == c
    mx = eng.from_bytes_wide("fr", np.full((1, 64), 0xFF, np.uint8))
    assert J.FR.to_mont_limbs(to_int(mx[0])) == [int(x, 16) for x in golden["fr"]["from_bytes_wide"]["max_output_mont"]]


def test_point_ops(eng, golden):
    P = rand_points(1, 300)
    Qp = rand_points(2, 300)
    tors = torsion_points(golden)
    ident = arr64([J.AFFINE_IDENTITY])
    P = np.concatenate([P, tors, ident, P[:4], P[:4]])
    Qp = np.concatenate([Qp, tors[::-1], ident, P[:4], O.point_op("neg", P[:4])])
    for op in ("double", "neg", "mul_by_cofactor"):
        got = getattr(eng, "point_" + op if op != "mul_by_cofactor" else op)(P)
        assert (got == O.point_op(op, P)).all(), op
    assert (eng.point_add(P, Qp) == O.point_op("add", P, Qp)).all()
    assert (eng.point_sub(P, Qp) == O.point_op("sub", P, Qp)).all()
    assert (eng.to_niels(P) == O.to_niels(P)).all()
    for pred in ("is_identity", "is_small_order", "is_on_curve", "is_torsion_free", "is_prime_order"):
        assert (eng.predicate(pred, P) == O.predicate(pred, P)).all(), pred
    off = P.copy()
    off[:, 0] ^= 1
    assert (eng.predicate("is_on_curve", off) == O.predicate("is_on_curve", off)).all()
    assert (eng.point_sum(P) == O.point_sum(P)).all()
    assert to_pt(eng.point_sum(P[:0])) == J.AFFINE_IDENTITY


def test_torsion_free_pairing_vs_reference_ladder(eng, golden, monkeypatch):
    """The default subgroup test (order-8 Tate pairing) against the reference's definition [r]P == O, computed by the
    C oracle's ladder and by this library's own ladder mode: random points of every coset of the subgroup, the
    small-order points, the identity."""
    from jubjub_amd import Engine

    tors = torsion_points(golden)
    sub = rand_points(71, 512, subgroup=True)
    cosets = np.concatenate([O.point_op("add", sub[64 * j:64 * (j + 1)], np.repeat(tors[j:j + 1], 64, axis=0)) for j in range(8)])
    P = np.concatenate([rand_points(72, 1500), cosets, tors, arr64([J.AFFINE_IDENTITY])])
    want = O.predicate("is_torsion_free", P)
    assert 0 < int(want.sum()) < len(P)
    assert (eng.predicate("is_torsion_free", P) == want).all()
    assert (eng.predicate("is_prime_order", P) == O.predicate("is_prime_order", P)).all()
    opts = {}
    opts['torsion_check_ladder'] = 1
    e2 = Engine(0, options=opts)
    assert (e2.predicate("is_torsion_free", P) == want).all()
    enc = O.compress(P)
    for flags in (1 | 2, 1 | 2 | 4 | 8):
        a, ka = eng.decompress(enc, flags)
        b, kb = e2.decompress(enc, flags)
        assert (ka == kb).all() and (a == b).all()
    e2.close()


def test_varbase_edges(eng, golden):
    pts = np.concatenate([rand_points(3, 8), torsion_points(golden), arr64([J.GENERATOR, J.AFFINE_IDENTITY])])
    S, Pn = [], []
    for k in EDGE_SCALARS:
        for p in pts:
            S.append(b32(k))
            Pn.append(p)
    S, Pn = np.stack(S), np.stack(Pn)
    assert (eng.varbase_mul(S, Pn) == O.varbase_mul(S, Pn)).all()
    assert (eng.varbase_mul_vartime(S, Pn) == O.varbase_mul(S, Pn)).all()          # the table ladder for public scalars


def test_varbase_per_lane_and_per_quad_kernels(golden, monkeypatch):
    """Batches up to JJ_VB_QUAD_MAX run one scalar multiplication per quad of lanes, larger ones one per lane; both
    kernels on the same inputs (edge scalars x special points, random, sizes around the wave/quad granularity)."""
    from jubjub_amd import Engine

    pts = np.concatenate([rand_points(3, 8), torsion_points(golden), arr64([J.GENERATOR, J.AFFINE_IDENTITY])])
    S = np.stack([b32(k) for k in EDGE_SCALARS for _ in pts])
    Pn = np.stack([p for _ in EDGE_SCALARS for p in pts])
    S = np.concatenate([S, rand_scalars(51, 1500, full_width=True)])
    Pn = np.concatenate([Pn, rand_points(52, 1500)])
    want = O.varbase_mul(S, Pn)
    for quad_max in ("0", "1048576"):
        opts = {}
        opts['vb_quad_max'] = int(quad_max)
        e2 = Engine(0, options=opts)
        assert (e2.varbase_mul(S, Pn) == want).all(), quad_max                     # constant-time: k_varbase_ct3 / k_varbase_ct_quad
        assert (e2.varbase_mul_vartime(S, Pn) == want).all(), quad_max             # table ladder: k_varbase / k_varbase_quad
        assert (e2.varbase_mul_vartime_compressed(S, Pn) == O.compress(want)).all(), quad_max
        for m in (1, 3, 15, 16, 17, 63, 64, 65, 255, 257):
            assert (e2.varbase_mul(S[:m], Pn[:m]) == want[:m]).all(), (quad_max, m)
            assert (e2.varbase_mul_vartime(S[:m], Pn[:m]) == want[:m]).all(), (quad_max, m)
        e2.close()


@pytest.mark.parametrize("window,quad_max", [("3", "0"), ("3", "32768"), ("2", "0")])
def test_varbase_constant_time_ladder(golden, monkeypatch, window, quad_max):
    """jj_varbase_mul_ct: no scalar-dependent address or branch, the reference's conditional_select discipline (src/lib.rs:334-343,
    357-379).  Signed 3-bit windows (default; table {P, 2P} in registers, {3P, 4P} in a per-lane LDS slot that is read whole for every
    window, mask selects) and signed 2-bit windows (JJ_VB_CT_WINDOW=2: {P, 2P} in registers alone): every edge scalar on random /
    torsion / generator / identity points, random inputs, ragged and empty batches, digit patterns that put every window value into
    every position class -- the same points as the table ladder and the oracle.  quad_max 32768: the batch runs one scalar
    multiplication per quad of lanes (k_varbase_ct_quad: every lane keeps its own coordinate of {P .. 4P} in registers); 0: one per lane."""
    from jubjub_amd import Engine

    opts = {}
    opts['vb_ct_window'] = int(window)
    opts['vb_quad_max'] = int(quad_max)
    eng = Engine(0, options=opts)
    pts = np.concatenate([rand_points(3, 8), torsion_points(golden), arr64([J.GENERATOR, J.AFFINE_IDENTITY])])
    S = np.stack([b32(k) for k in EDGE_SCALARS for _ in pts])
    Pn = np.stack([p for _ in EDGE_SCALARS for p in pts])
    S = np.concatenate([S, rand_scalars(151, 3000, full_width=True)])
    Pn = np.concatenate([Pn, rand_points(152, 3000)])
    want = O.varbase_mul(S, Pn)
    got = eng.varbase_mul_ct(S, Pn)
    assert (got == want).all()
    assert (got == eng.varbase_mul(S, Pn)).all() and (got == eng.varbase_mul_vartime(S, Pn)).all()
    for m in (0, 1, 63, 64, 65, 257):
        assert (eng.varbase_mul_ct(S[:m], Pn[:m]) == want[:m]).all(), m
    # two-bit patterns: every window value in every position class
    pat = arr32([int(d * 64, 4) & ((1 << 252) - 1) for d in "0123"] + [int("0123" * 16, 4), int("3210" * 16, 4), int("2" * 63, 4) * 4 + 3])
    # three-bit patterns: every octal digit everywhere, rising / falling runs (carries of the signed recoding through whole runs of 3s and 4s)
    pat = np.concatenate([pat, arr32([int(d * 84, 8) & ((1 << 252) - 1) for d in "01234567"] + [int("01234567" * 10 + "0123", 8), int("76543210" * 10 + "7654", 8) & ((1 << 252) - 1),
                                                                                               int("34" * 42, 8), int("43" * 42, 8) & ((1 << 252) - 1)])])
    Pq = rand_points(153, len(pat))
    assert (eng.varbase_mul_ct(pat, Pq) == O.varbase_mul(pat, Pq)).all()
    eng.close()


def test_varbase_random(eng):
    n = 3000   # not a multiple of the block size; exercises the grid-stride tail
    S = rand_scalars(5, n, full_width=True)
    P = rand_points(6, n)
    got = eng.varbase_mul(S, P)
    assert (got == O.varbase_mul(S, P)).all()
    assert eng.varbase_mul(S[:0], P[:0]).shape == (0, 64)
    assert (eng.varbase_mul(S[:1], P[:1]) == got[:1]).all()
    assert (eng.varbase_mul_compressed(S, P) == O.compress(got)).all()
    tab = eng.fixedbase_table(P[0])
    assert (eng.fixedbase_mul_compressed(tab, S) == O.compress(O.fixedbase_mul(S, P[0]))).all()
    tab.close()


def test_varbase_shared_scalar(eng, golden):
    pts = np.concatenate([rand_points(91, 700), torsion_points(golden), arr64([J.GENERATOR, J.AFFINE_IDENTITY])])
    for k in (0, 1, R - 1, (1 << 252) - 1, to_int(rand_scalars(92, 1, full_width=True)[0])):
        S = np.repeat(b32(k)[None, :], len(pts), axis=0)
        assert (eng.varbase_mul_scalar(b32(k), pts) == O.varbase_mul(S, pts)).all(), k
    assert eng.varbase_mul_scalar(b32(5), pts[:0]).shape == (0, 64)


def test_varbase_shared_scalar_kernel_large_ragged_batch(eng, golden, monkeypatch):
    """above JJ_VB_QUAD_MAX the shared-scalar kernel runs (k_varbase<.., SHARED>: the scalar is read through a wave-uniform
    address, its digits live in scalar registers); ragged batch sizes exercise the waves' work cursor"""
    from jubjub_amd import Engine

    opts = {}
    opts['vb_quad_max'] = 0                # every size through the per-lane kernels
    e2 = Engine(0, options=opts)
    pts = np.concatenate([rand_points(93, 40001 - 10), torsion_points(golden), arr64([J.GENERATOR, J.AFFINE_IDENTITY])])
    for k in (to_int(rand_scalars(94, 1, full_width=True)[0]), R - 1):
        S = np.repeat(b32(k)[None, :], len(pts), axis=0)
        want = O.varbase_mul(S, pts)
        assert (e2.varbase_mul_scalar(b32(k), pts) == want).all()
        for m in (1, 63, 65, 4099):
            assert (e2.varbase_mul_scalar(b32(k), pts[:m]) == want[:m]).all(), m
    Sr = rand_scalars(95, len(pts), full_width=True)
    assert (e2.varbase_mul(Sr, pts) == O.varbase_mul(Sr, pts)).all()      # per-unit scalars, same ragged size
    e2.close()


def test_varbase_exact_projective(eng):
    n = 200
    S = np.concatenate([arr32(EDGE_SCALARS), rand_scalars(7, n)])
    P = rand_points(8, len(S))
    got = eng.varbase_mul_exact(S, P)
    assert (got == O.varbase_mul_ext(S, P)).all()      # all five projective coordinates, bit for bit


def test_eight_torsion_through_gpu(eng, golden):
    # reference find_eight_torsion (src/lib.rs:1679-1696)
    g = eng.varbase_mul(np.array([golden["FR_MODULUS_BYTES"]["bytes"]], np.uint8), arr64([J.GENERATOR]))
    tors = torsion_points(golden)
    cur = g.copy()
    for t in tors:
        assert (cur[0] == t).all()
        cur = eng.point_add(cur, g)


def test_fixedbase(eng):
    for base in (J.GENERATOR, to_pt(O.point_op("mul_by_cofactor", arr64([J.GENERATOR]))[0]), to_pt(rand_points(9, 1)[0])):
        S = np.concatenate([arr32(EDGE_SCALARS), rand_scalars(10, 2500, full_width=True)])
        want = O.fixedbase_mul(S, pt64(base))
        for wbits in (0, 7, 6, 8, 10, 12):    # 0 = 7: signed comb in LDS; 6: LDS window table (both constant-time selects); 8..12: L2-resident wide windows
            tab = eng.fixedbase_table(pt64(base), wbits)
            got = eng.fixedbase_mul(tab, S)
            assert (got == want).all(), wbits
            assert eng.fixedbase_mul(tab, S[:0]).shape == (0, 64)
            tab.close()
    with pytest.raises(Exception):
        eng.fixedbase_table(pt64(J.GENERATOR), 5)


def test_fixedbase_signed_comb(golden):
    """k_fixedbase_comb (8 teeth, 8 column blocks, 32 additions + 3 doublings; even scalars take the last entry from T_0 -+ B):
    every parity / sign class of the last column, scalars whose comb columns are all +, all -, alternating, 0, 1, 2, r - 1, r,
    2^252 - 1, top bits set; bases of every kind (generator, prime-order, 8-torsion, order 2, identity); ragged waves; the
    the constant-time shuffle select (the per-lane LDS gather exists in -DJJ_EXPERIMENTS probe builds only: tools/fixedbase_floor.sh); and the
    chained (multi-base) form."""
    from jubjub_amd import Engine

    e2 = Engine(0)
    comb = [0, 1, 2, 3, 4, (1 << 252) - 1, (1 << 252) - 2, 1 << 251, (1 << 251) + 1, sum(1 << (32 * i) for i in range(8)) & ((1 << 252) - 1), sum(1 << (32 * i + 1) for i in range(8)) & ((1 << 252) - 1), sum(0x88888888 << (32 * i) for i in range(8)) & ((1 << 252) - 1),
            int("5" * 63, 16), int("a" * 62, 16), (1 << 224) - 1, 1 << 224, (1 << 224) + 2, (1 << 32) - 1, 1 << 32, (1 << 31) | 1, R - 1, R, R + 1, 8 * R - 1]
    S = np.concatenate([arr32(EDGE_SCALARS), arr32([k & ((1 << 256) - 1) for k in comb]), rand_scalars(171, 1500, full_width=True)])
    tors = torsion_points(golden)
    bases = [pt64(J.GENERATOR), O.point_op("mul_by_cofactor", arr64([J.GENERATOR]))[0], rand_points(172, 1)[0], tors[1], tors[4], pt64(J.AFFINE_IDENTITY)]
    for b in bases:
        tab = e2.fixedbase_table(b, 7)
        want = O.fixedbase_mul(S, b)
        assert (e2.fixedbase_mul(tab, S) == want).all()
        for m in (1, 63, 64, 65, 129):
            assert (e2.fixedbase_mul(tab, S[:m]) == want[:m]).all(), m
        tab.close()
    t1, t2, t3 = e2.fixedbase_table(bases[0], 7), e2.fixedbase_table(bases[2], 7), e2.fixedbase_table(bases[1], 6)
    S3 = np.stack([S, S[::-1].copy(), np.roll(S, 7, axis=0)])
    want = O.point_op("add", O.point_op("add", O.fixedbase_mul(S3[0], bases[0]), O.fixedbase_mul(S3[1], bases[2])), O.fixedbase_mul(S3[2], bases[1]))
    assert (e2.fixedbase_multi_mul([t1, t2, t3], S3) == want).all()
    for t in (t1, t2, t3):
        t.close()
    e2.close()


def test_fixedbase_multi_base_sums(eng):
    """jj_fixedbase_multi_mul: sum_j k_ij * B_j for LDS tables, wide-window tables and a mix, against the oracle's
    AffineNielsPoint ladders folded with point additions."""
    n = 1500
    bases = rand_points(81, 3)
    S = np.stack([rand_scalars(82 + j, n, full_width=(j == 1)) for j in range(3)])
    S[0, :len(EDGE_SCALARS)] = arr32(EDGE_SCALARS)
    want = O.fixedbase_mul(S[0], bases[0])
    for j in (1, 2):
        want = O.point_op("add", want, O.fixedbase_mul(S[j], bases[j]))
    for widths in ((0, 0, 0), (10, 10, 10), (0, 12, 0)):
        tabs = [eng.fixedbase_table(bases[j], widths[j]) for j in range(3)]
        assert (eng.fixedbase_multi_mul(tabs, S) == want).all(), widths
        assert (eng.fixedbase_multi_mul(tabs[:1], S[:1]) == O.fixedbase_mul(S[0], bases[0])).all()
        for t in tabs:
            t.close()
    tab = eng.fixedbase_table(bases[0])
    assert eng.fixedbase_multi_mul([tab, tab], S[:2, :0]).shape == (0, 64)
    tab.close()


@pytest.mark.parametrize("bits", [[64, 64, 64], [124, 124], [40] * 6, [1, 250 - 9], [10] * 21, [250]])
def test_fixedbase_composite_short_scalars_one_pass(eng, bits):
    """jj_fixedbase_composite_*: several bases with short scalars through ONE LDS table set and one accumulator per lane (SURVEY
    8(f)-4), against the oracle's AffineNielsPoint ladders on the masked scalars folded with point additions
    (multiply_bits, reference src/lib.rs:297-301): field-boundary patterns (all ones, top bit only, zero), bits above the field
    ignored, full-group bases."""
    nb, n = len(bits), 700
    bases = rand_points(300 + nb, nb)
    S = np.stack([rand_scalars(310 + b, n, full_width=True) for b in range(nb)])
    for b in range(nb):
        S[b, 0] = 0xFF                                                          # all ones: the field holds 2^bits - 1, the rest is ignored
        S[b, 1] = b32(1 << (bits[b] - 1))
        S[b, 2] = 0
        S[b, 3] = b32((1 << bits[b]) - 1)
    tab = eng.fixedbase_composite_table(bases, bits)
    got = eng.fixedbase_composite_mul(tab, S)
    want = None
    for b in range(nb):
        masked = arr32([to_int(x) & ((1 << bits[b]) - 1) for x in S[b]])
        term = O.fixedbase_mul(masked, bases[b])
        want = term if want is None else O.point_op("add", want, term)
    assert (got == want).all(), bits
    assert eng.fixedbase_composite_mul(tab, S[:, :0]).shape == (0, 64)
    tab.close()
    with pytest.raises(Exception):
        eng.fixedbase_composite_table(rand_points(1, 4), [64, 64, 64, 64])     # 44 windows: does not fit the 42 slots


def test_msm(eng):
    for n in (0, 1, 2, 3, 7, 33, 127, 511, 512, 1000, 2047, 2048, 5000, 32767, 32768, 40000):
        S = rand_scalars(12 + n, n, full_width=True)
        P = rand_points(13 + n, n, subgroup=(n % 2 == 0))
        assert (eng.msm(S, P) == O.msm(S, P)).all(), n


@pytest.mark.parametrize("small_max", ["0", "100000"])
def test_msm_small_batch_path_and_pippenger_on_the_same_inputs(monkeypatch, small_max):
    """JJ_MSM_SMALL_MAX: the two-launch small-batch path (per-term tables, 64 windows of 3-4 bits; default up to 2^14 terms) and
    Pippenger, each forced over every size: ragged sizes around the 128-quad workgroups, edge scalars, identity / torsion points."""
    from jubjub_amd import Engine

    opts = {}
    opts['msm_small_max'] = int(small_max)
    e2 = Engine(0, options=opts)
    for n in (1, 2, 33, 127, 128, 129, 511, 513, 700, 1025, 4097, 40000):
        S = rand_scalars(112 + n, n, full_width=True)
        P = rand_points(113 + n, n, subgroup=(n % 2 == 0))
        assert (e2.msm(S, P) == O.msm(S, P)).all(), (small_max, n)
    m = len(EDGE_SCALARS)
    S = arr32(EDGE_SCALARS)
    g8 = J.scalar_mul_fast(J.GENERATOR, J.R_MOD)
    special = arr64([J.AFFINE_IDENTITY, g8, J.scalar_mul_fast(g8, 4), J.GENERATOR, J.affine_neg(J.GENERATOR)])
    for k in range(len(special)):
        P = np.repeat(special[k:k + 1], m, axis=0)
        assert (e2.msm(S, P) == O.msm(S, P)).all(), (small_max, "special", k)
        for i in range(0, m, 5):                                                  # single terms: every edge scalar on its own
            assert (e2.msm(S[i:i + 1], P[i:i + 1]) == O.msm(S[i:i + 1], P[i:i + 1])).all(), (small_max, k, i)
    e2.close()


@pytest.mark.parametrize("mode", ["segments", "chunks", "chunks-offsets-in-memory"])
def test_msm_both_accumulation_schemes(monkeypatch, mode):
    """Bucket accumulation by length-sorted segments (default from 147 456 terms) and by fixed chunks + fix-up (below; the bucket offsets
    staged in LDS, round 6, or read from memory as in rounds 2-5), forced in turn on the same inputs, including skewed digit distributions
    and short segments."""
    from jubjub_amd import Engine

    opts = {}
    opts['msm_accum'] = {"segments": 1, "chunks": 0, "chunks-offsets-in-memory": 0}[mode]
    if mode == "chunks-offsets-in-memory":
        opts['msm_acc_lds'] = 0
    opts['msm_seg_len'] = 8
    opts['msm_small_max'] = 0
    e2 = Engine(0, options=opts)
    for n in (1, 5, 300, 4099, 70000):
        S = rand_scalars(212 + n, n, full_width=True)
        P = rand_points(213 + n, n)
        assert (e2.msm(S, P) == O.msm(S, P)).all(), (mode, n)
    n = 6000
    S = np.repeat(rand_scalars(31, 1, full_width=True), n, axis=0)      # every term in the same bucket of every window
    P = rand_points(32, n)
    assert (e2.msm(S, P) == O.msm(S, P)).all()
    S2 = rand_scalars(33, n)
    S2[: n // 2] = S2[0]
    assert (e2.msm(S2, P) == O.msm(S2, P)).all()
    e2.close()


def test_msm_skewed_buckets(eng):
    """All-equal scalars / repeated points: every term of a window lands in one bucket (worst case for bucket methods)."""
    n = 6000
    S = np.repeat(rand_scalars(31, 1, full_width=True), n, axis=0)
    P = rand_points(32, n)
    assert (eng.msm(S, P) == O.msm(S, P)).all()
    S2 = rand_scalars(33, n)
    S2[: n // 2] = S2[0]
    P2 = np.repeat(rand_points(34, 3), n // 3, axis=0)
    assert (eng.msm(S2, P2) == O.msm(S2, P2)).all()
    Z = np.zeros((n, 32), np.uint8)                      # all-zero scalars: every digit of the recoded form cancels
    assert to_pt(eng.msm(Z, P)) == J.AFFINE_IDENTITY


def test_msm_around_the_large_input_switch(eng):
    """Default configuration on both sides of the switch from 23 windows + chunks + fix-up to 17 windows + length-sorted segments
    (147 456 terms), ragged sizes, against the oracle."""
    for n in (147455, 147457):
        S = rand_scalars(81 + n, n, full_width=True)
        P = rand_points(82, n)
        assert (eng.msm(S, P) == O.msm(S, P)).all(), n


def test_msm_big_bucket_list_overflow(eng):
    """128 distinct scalars repeated over 2^17 terms: every non-empty bucket holds 1024 entries = 64 chunk heads, and there are about
    128 x 23 of them -- more than the big-bucket work list holds (2048), so the fix-up's pairs of lanes also run their serial
    fallback, next to a full work list for the workgroup-per-bucket kernel."""
    n = 1 << 17
    base = rand_scalars(71, 128, full_width=True)
    S = np.ascontiguousarray(base[np.arange(n) % 128])
    P = rand_points(72, n)
    assert (eng.msm(S, P) == O.msm(S, P)).all()


@pytest.mark.parametrize("window", [16, 17, 19, 20, 21, 23, 28, 32])
@pytest.mark.parametrize("sort", ["2pass", "1pass"])
def test_msm_window_counts_both_sorts(monkeypatch, window, sort):
    """JJ_MSM_WINDOWS: W windows tiling the 253 scalar bits exactly (16: 13 windows of 16 bits + 3 of 15, the default from 2^20
    terms; 21: one of 13 bits + 20 of 12; 23: all 11 bits; ...) forced on small inputs, with the two-pass sort (coarse bin, then
    the low bits in LDS; always taken above 4096 buckets per window) and the single-pass one (JJ_MSM_SORT decides at exactly 4096):
    ragged sizes, a bin far larger than the LDS stage (equal scalars), zero digits, the largest top-window digit."""
    from jubjub_amd import Engine

    opts = {}
    opts['msm_windows'] = window
    opts['msm_sort_two_pass'] = {"2pass": 1, "1pass": 0}[sort]
    opts['msm_small_max'] = 0
    e2 = Engine(0, options=opts)
    for n in (1, 2, 300, 8191, 8193, 30000):
        S = rand_scalars(512 + n + window, n, full_width=True)
        P = rand_points(513 + n, n, subgroup=(n % 2 == 0))
        assert (e2.msm(S, P) == O.msm(S, P)).all(), (window, sort, n)
    if sort == "1pass" and window not in (16, 20, 23, 32):
        e2.close()
        return                                                   # the skewed cases below once per bucket-count class
    n = 20000
    P = rand_points(61, n)
    S = np.repeat(rand_scalars(62, 1, full_width=True), n, axis=0)      # one bucket per window holds every term
    assert (e2.msm(S, P) == O.msm(S, P)).all()
    S2 = rand_scalars(63, n)
    S2[: n // 2] = S2[0]
    S2[n // 2: n // 2 + 500] = 0
    S2[-1] = 0xFF
    S2[-1, 31] = 0x0F                                                     # 2^252 - 1: largest top-window digit after recoding
    assert (e2.msm(S2, P) == O.msm(S2, P)).all()
    e2.close()


@pytest.mark.parametrize("window,rows,l2chunk", [(16, 8, 0), (16, 4, 8), (16, 32, 2), (16, 0, 0), (17, 4, 0), (17, 16, 4), (19, 8, 0), (20, 2, 16), (23, 4, 0), (32, 2, 0)])
def test_msm_two_level_bucket_reduce(monkeypatch, window, rows, l2chunk):
    """JJ_MSM_REDUCE_L1 / JJ_MSM_REDUCE_L2_CHUNK: the two-level bucket reduce (lane-form column sums S_m, T_m over `rows` rows of the
    bucket matrix, then the quad chain over the columns; the default from 16 384 buckets per window) forced over window layouts with
    one and two window widths, every level-2 chunk length class, one workgroup and several per window, and switched off (rows = 0);
    ragged sizes, equal scalars (one bucket per window holds everything), zero digits and the largest top-window digit."""
    from jubjub_amd import Engine

    opts = {}
    opts['msm_windows'] = window
    opts['msm_reduce_l1'] = rows
    if l2chunk:
        opts['msm_reduce_l2_chunk'] = l2chunk
    opts['msm_small_max'] = 0
    e2 = Engine(0, options=opts)
    for n in (1, 2, 301, 30011):
        S = rand_scalars(1512 + n + window, n, full_width=True)
        P = rand_points(1513 + n, n, subgroup=(n % 2 == 0))
        assert (e2.msm(S, P) == O.msm(S, P)).all(), (window, rows, l2chunk, n)
    n = 9000
    P = rand_points(1661, n)
    S = np.repeat(rand_scalars(1662, 1, full_width=True), n, axis=0)
    assert (e2.msm(S, P) == O.msm(S, P)).all()
    S2 = rand_scalars(1663, n)
    S2[: n // 2] = S2[0]
    S2[n // 2: n // 2 + 500] = 0
    S2[-1] = 0xFF
    S2[-1, 31] = 0x0F
    assert (e2.msm(S2, P) == O.msm(S2, P)).all()
    # every bucket of the low windows in use: scalars 1 .. 2^15 and their negatives' neighbours (digits of both signs, all rows and columns)
    n = 1 << 15
    S3 = np.zeros((n, 32), np.uint8)
    v = np.arange(1, n + 1, dtype=np.uint32)
    S3[:, 0] = v & 0xFF; S3[:, 1] = (v >> 8) & 0xFF; S3[:, 2] = (v >> 16) & 0xFF
    P3 = np.repeat(rand_points(1664, 8), n // 8, axis=0)
    assert (e2.msm(S3, P3) == O.msm(S3, P3)).all()
    e2.close()


@pytest.mark.parametrize("mode", ["separate", "fused"])
@pytest.mark.parametrize("window", [16, 17, 18])
def test_msm_two_pass_sort_histogram_modes(monkeypatch, mode, window):
    """JJ_MSM_SORT_HIST: the coarse histogram of the two-pass sort taken inside the conversion kernel, the tiles' runs reserved with global atomics
    (k_msm_convert_hist; the default up to 3 x 2^20 terms), against the separate histogram + plan kernels (round 4; the default above): ragged sizes around
    the 1024-term and 4096-term workgroups and the 8192-term tiles, a window partition (slots != windows), equal scalars (one bin holds every entry), zeros."""
    from jubjub_amd import Engine

    opts = {}
    opts['msm_sort_hist_fused'] = {"separate": 0, "fused": 1}[mode]
    opts['msm_windows'] = window
    opts['msm_small_max'] = 0
    e2 = Engine(0, options=opts)
    for n in (1, 1023, 1025, 4097, 8191, 8193, 50021):
        S = rand_scalars(3512 + n + window, n, full_width=True)
        P = rand_points(3513 + n, n, subgroup=(n % 2 == 0))
        assert (e2.msm(S, P) == O.msm(S, P)).all(), (mode, window, n)
    n = 20000
    S, P = rand_scalars(3600, n, full_width=True), rand_points(3601, n)
    recs = np.stack([e2.msm_partial(S, P, g, 3) for g in range(3)])                 # windows g, g + 3, ...: slot s is window 3 s + g
    assert (e2.msm_combine(recs) == O.msm(S, P)).all()
    S1 = np.repeat(rand_scalars(3602, 1, full_width=True), n, axis=0)
    assert (e2.msm(S1, P) == O.msm(S1, P)).all()
    S2 = rand_scalars(3603, n)
    S2[: n // 2] = 0
    assert (e2.msm(S2, P) == O.msm(S2, P)).all()
    for _ in range(3):                                                                # back to back: the two parities of the totals / cursors
        assert (e2.msm(S, P) == O.msm(S, P)).all()
    e2.close()


def test_msm_back_to_back_sizes(monkeypatch):
    """Pippenger calls of changing sizes back to back on one context: every call reuses (and regrows) the workspaces of the one
    before it, including the LDS-staged conversion's ragged last workgroup (n not a multiple of 64)."""
    from jubjub_amd import Engine

    opts = {}
    opts['msm_small_max'] = 0
    e2 = Engine(0, options=opts)
    for n in (1, 300, 5000, 40000, 2000, 40001, 63, 65):
        S = rand_scalars(712 + n, n, full_width=True)
        P = rand_points(713 + n, n)
        assert (e2.msm(S, P) == O.msm(S, P)).all(), n
    e2.close()


def test_msm_multipass(monkeypatch):
    """Inputs larger than one Pippenger pass are folded pass by pass (pass size shrunk here via the env knob)."""
    from jubjub_amd import Engine

    opts = {}
    opts['msm_pass_log2'] = 12
    e2 = Engine(0, options=opts)
    n = 10000
    S, P = rand_scalars(41, n, full_width=True), rand_points(42, n)
    assert (e2.msm(S, P) == O.msm(S, P)).all()
    n = 4096 * 2 + 77                                       # the last pass is small: two window layouts meet in one host tail
    assert (e2.msm(S[:n], P[:n]) == O.msm(S[:n], P[:n])).all()
    e2.close()


def test_msm_async_jobs_interleaved(eng):
    """jj_msm_begin / jj_msm_finish: several MSMs of different sizes (small-batch path, one-pass and two-pass Pippenger, empty)
    queued back to back on one context, finished out of order, each exactly once; the host tail of one job runs while the
    kernels of the next are in flight.  Device-resident and host inputs."""
    import torch

    sizes = (3000, 0, 1, 40000, 129, 70000, 17000)
    data = [(rand_scalars(900 + n, n, full_width=True), rand_points(901 + n, n)) for n in sizes]
    want = [O.msm(S, P) for S, P in data]
    jobs = [eng.msm_begin(S, P) for S, P in data]                         # host arrays: staged at begin
    for k in (3, 0, 6, 1, 5, 2, 4):
        assert (eng.msm_finish(jobs[k]) == want[k]).all(), ("host", sizes[k])
    with pytest.raises(Exception):
        eng.msm_finish(jobs[0])                                           # a job is finished once
    dev = torch.device("cuda", 0)
    dd = [(torch.from_numpy(S).to(dev), torch.from_numpy(P).to(dev)) for S, P in data]
    for depth in (2, 4):                                                  # a sliding window of jobs in flight, as bench.py runs them
        pending, got = [], []
        for S, P in dd * 2:
            pending.append(eng.msm_begin(S, P))
            if len(pending) == depth:
                got.append(eng.msm_finish(pending.pop(0)))
        got += [eng.msm_finish(j) for j in pending]
        for k, g in enumerate(got):
            assert (g == want[k % len(sizes)]).all(), ("device", depth, k)


@pytest.mark.parametrize("lanes", ["1", "3", "4"])
def test_msm_jobs_over_several_lanes(monkeypatch, lanes):
    """JJ_MSM_LANES: device-pointer jobs of jj_msm_begin alternate over the context's MSM lanes (own streams and workspaces; default 2),
    so the dependent chains of one MSM overlap the sort / accumulation of the next: mixed sizes (small-batch path, one-pass and
    two-pass Pippenger, multi-pass), more jobs in flight than lanes, workspaces that grow while other lanes are busy, results
    written to device memory, and the synchronous jj_msm in between (lane 0)."""
    import torch

    from jubjub_amd import Engine

    opts = {}
    opts['msm_lanes'] = int(lanes)
    opts['msm_pass_log2'] = 16
    e2 = Engine(0, options=opts)
    dev = torch.device("cuda", 0)
    sizes = (700, 40000, 5, 20000, 70000, 300, 100000, 17000, 33000)
    data = [(rand_scalars(1200 + n, n, full_width=True), rand_points(1201 + n, n)) for n in sizes]
    want = [O.msm(S, P) for S, P in data]
    dd = [(torch.from_numpy(S).to(dev), torch.from_numpy(P).to(dev)) for S, P in data]
    pending, got = [], []
    for k, (S, P) in enumerate(dd + dd):
        pending.append(e2.msm_begin(S, P))
        if k % 5 == 2:
            assert (e2.msm(*dd[k % len(dd)]).cpu().numpy() == want[k % len(dd)]).all()      # a synchronous MSM between the jobs
        if len(pending) == 5:
            got.append(e2.msm_finish(pending.pop(0)))
    got += [e2.msm_finish(j) for j in pending]
    for k, g in enumerate(got):
        assert (g == want[k % len(sizes)]).all(), (lanes, k)
    e2.close()


@pytest.mark.parametrize("G", [2, 3, 8])
def test_msm_partial_records_term_and_window_partition(eng, G):
    """jj_msm_partial + jj_msm_combine: the MSM cut G ways by terms (every part: all windows of its own terms) and by windows (every
    part: windows g, g + G, ... of all terms), records gathered and combined in one host tail; small-batch and Pippenger sizes,
    uneven shards (parts of different window layouts), an empty part."""
    from jubjub_amd import _lib
    from jubjub_amd.dist import shard_bounds

    for n in (5, 1000, 20000, 70000):
        S, P = rand_scalars(950 + n, n, full_width=True), rand_points(951 + n, n)
        want = O.msm(S, P)
        recs = np.stack([eng.msm_partial(S, P, g, G) for g in range(G)])
        assert recs.shape == (G, _lib.MSM_PARTIAL_BYTES)
        assert (eng.msm_combine(recs) == want).all(), ("windows", n, G)
        parts = []
        for g in range(G):
            lo, hi = shard_bounds(n, g, G)
            parts.append(eng.msm_partial(S[lo:hi], P[lo:hi]))
        assert (eng.msm_combine(np.stack(parts)) == want).all(), ("terms", n, G)
    # uneven cut: a Pippenger part, a small-batch part and an empty one
    n = 40000
    S, P = rand_scalars(977, n), rand_points(978, n)
    parts = [eng.msm_partial(S[:30000], P[:30000]), eng.msm_partial(S[30000:], P[30000:]), eng.msm_partial(S[:0], P[:0])]
    assert (eng.msm_combine(np.stack(parts)) == O.msm(S, P)).all()
    assert to_pt(eng.msm_combine(np.zeros((0, _lib.MSM_PARTIAL_BYTES), np.uint8))) == J.AFFINE_IDENTITY
    bad = parts[0].copy()
    bad[0] ^= 1                                                            # damaged magic
    with pytest.raises(Exception):
        eng.msm_combine(bad[None, :])
    # device-resident: the record is a CUDA tensor (what all_gather moves), combined after one copy to the host
    import torch

    dev = torch.device("cuda", 0)
    Sd, Pd = torch.from_numpy(S).to(dev), torch.from_numpy(P).to(dev)
    recs = torch.stack([eng.msm_partial(Sd, Pd, g, G) for g in range(G)])
    assert recs.is_cuda and (eng.msm_combine(recs) == O.msm(S, P)).all()


@pytest.mark.parametrize("fold", ["device", "host"])
def test_msm_gathered_records_folded_on_the_device(monkeypatch, fold):
    """jj_msm_combine_dev (what jj_msm_allgather runs after ncclAllGather): G device-resident records are added window by window on the
    device into ONE record (k_msm_fold_records) before the host tail -- both partitions, G = 2, 3, 8 and 70 (more records than the
    fold's 64 quads), an empty shard between the others, and records of different window layouts (host fallback).  JJ_MSM_FOLD=host
    keeps round 4's path (every record copied, the host adds them): both must give the oracle's point."""
    import torch

    from jubjub_amd import Engine, _lib
    from jubjub_amd.dist import shard_bounds

    opts = {}
    opts['msm_fold_dev'] = {"host": 0, "device": 1}[fold]
    opts['msm_fold_min'] = 2                       # (the default folds on the device from 8 records)
    e2 = Engine(0, options=opts)
    dev = torch.device("cuda", 0)
    n = 70000
    S, P = rand_scalars(2950, n, full_width=True), rand_points(2951, n)
    want = O.msm(S, P)
    Sd, Pd = torch.from_numpy(S).to(dev), torch.from_numpy(P).to(dev)
    for G in (2, 3, 8, 70):
        parts = []
        for g in range(G):
            lo, hi = shard_bounds(n, g, G)
            parts.append(e2.msm_partial(Sd[lo:hi], Pd[lo:hi]))
        assert (e2.msm_combine(torch.stack(parts)) == want).all(), ("terms", G)
        if G <= 8:
            recs = torch.stack([e2.msm_partial(Sd, Pd, g, G) for g in range(G)])
            assert (e2.msm_combine(recs) == want).all(), ("windows", G)
    # Pippenger shards of one layout with an empty shard in the middle; then a small-batch shard among them (64 windows against 23)
    a, b = e2.msm_partial(Sd[:30000], Pd[:30000]), e2.msm_partial(Sd[30000:60000], Pd[30000:60000])
    empty = e2.msm_partial(Sd[:0], Pd[:0])
    assert (e2.msm_combine(torch.stack([a, empty, b])) == O.msm(S[:60000], P[:60000])).all()
    small = e2.msm_partial(Sd[60000:], Pd[60000:])
    assert (e2.msm_combine(torch.stack([a, small, b, empty])) == want).all()
    assert to_pt(e2.msm_combine(torch.zeros((0, _lib.MSM_PARTIAL_BYTES), dtype=torch.uint8, device=dev))) == J.AFFINE_IDENTITY
    assert (e2.msm_combine(a[None, :]) == O.msm(S[:30000], P[:30000])).all()            # one record: no fold
    bad = torch.stack([a, b]).clone()
    bad[1, 0] ^= 1                                                                       # damaged magic in a gathered record
    with pytest.raises(Exception):
        e2.msm_combine(bad)
    e2.close()


@pytest.mark.parametrize("G, fold_min", [(8, 8), (2, 8), (3, 2)])
def test_msm_allgather_of_G_ranks_played_on_one_gpu(monkeypatch, G, fold_min):
    """jj_msm_allgather and jj_msm_allgather_begin / jj_msm_finish with G > 1 ranks, which a 1-GPU box cannot hold: the context gets an
    all-gather that plays the other ranks (tests/util.py LoopbackComm: their records, computed beforehand, land in the other slots of the
    receive buffer, stream-ordered like RCCL's kernel).  Every rank's call must return the oracle's sum of ALL terms -- term and
    window partition, ragged shards, device and host inputs, G records folded on the device (G >= fold_min) or added by the host,
    three jobs with different terms in flight finished out of order, and shards of different window layouts (a 100-term rank among
    Pippenger ranks: the fold kernel declines, the finish copies all records out of the job's own buffer)."""
    import torch

    from jubjub_amd import Engine
    from jubjub_amd.dist import shard_bounds
    from util import LoopbackComm

    opts = {}
    opts['msm_fold_min'] = fold_min
    dev = torch.device("cuda", 0)
    e2 = Engine(0, options=opts)
    n = 9000 * G + 5
    batches = []
    for k in range(3):
        S, P = rand_scalars(4100 + k, n, full_width=True), rand_points(4200 + k, n)
        batches.append((S, P, torch.from_numpy(S).to(dev), torch.from_numpy(P).to(dev), O.msm(S, P).reshape(64)))
    for partition in ("terms", "window"):
        def mine(b, g):
            if partition == "window":
                return b[2], b[3]
            lo, hi = shard_bounds(n, g, G)
            return b[2][lo:hi], b[3][lo:hi]

        recs = [torch.stack([e2.msm_partial(*mine(b, g)) if partition == "terms" else e2.msm_partial(b[2], b[3], g, G) for g in range(G)]) for b in batches]
        for rank in sorted({0, G // 2, G - 1}):
            comm = LoopbackComm(rank, G)
            for r in recs:
                comm.add_round(r)
            e2.set_comm(comm)
            for k, b in enumerate(batches):                           # synchronous calls, one per prepared round
                ds, dp = mine(b, rank)
                got = e2.msm_allgather(ds, dp, partition) if k != 1 else e2.msm_allgather(ds.cpu().numpy(), dp.cpu().numpy(), partition)
                assert (got == b[4]).all(), (partition, rank, k)
            jobs = [e2.msm_allgather_begin(*mine(b, rank), partition) for b in batches]        # rounds 3, 4, 5 = the same three again
            for k in (2, 0, 1):
                assert (e2.msm_finish(jobs[k]) == batches[k][4]).all(), (partition, rank, "job", k)
            assert comm.calls() == 6
            e2.set_comm(None)
            comm.close()
    # ranks whose shards have different window layouts: rank 1 holds 100 terms (small-batch path, 64 windows), the others thousands
    S, P, Sd, Pd, want = batches[0]
    cuts = [0, 9000, 9100] + [9100 + (n - 9100) * g // (G - 2) for g in range(1, G - 1)] if G > 2 else [0, n - 100, n]
    cuts = cuts[:G] + [n]
    recs = torch.stack([e2.msm_partial(Sd[cuts[g]:cuts[g + 1]], Pd[cuts[g]:cuts[g + 1]]) for g in range(G)])
    for rank in (0, 1):
        comm = LoopbackComm(rank, G)
        comm.add_round(recs)
        e2.set_comm(comm)
        lo, hi = cuts[rank], cuts[rank + 1]
        assert (e2.msm_allgather(Sd[lo:hi], Pd[lo:hi]) == want).all(), ("mixed layouts", rank)
        j = e2.msm_allgather_begin(Sd[lo:hi], Pd[lo:hi])
        assert (e2.msm_finish(j) == want).all(), ("mixed layouts, job", rank)
        e2.set_comm(None)
        comm.close()
    e2.close()


def test_serialization_golden(eng, golden):
    encs = np.array(golden["serialization_16"]["encodings"], np.uint8)
    gen8 = eng.mul_by_cofactor(arr64([J.GENERATOR]))
    pts = eng.varbase_mul(arr32(list(range(1, 17))), np.repeat(gen8, 16, axis=0))
    assert (eng.compress(pts) == encs).all()                          # reference src/lib.rs:1811-1887
    out, ok = eng.decompress(encs)
    assert ok.all() and (out == pts).all()
    z = np.array(golden["zip216_noncanonical"]["encodings"], np.uint8)  # reference src/lib.rs:1893-1935
    assert not eng.decompress(z)[1].any()
    o2, k2 = eng.decompress(z, flags=0)
    assert k2.all()
    re = eng.compress(o2)
    re[:, 31] |= 0x80
    assert (re == z).all()


def test_decompress_flags(eng, golden):
    rng = np.random.default_rng(77)
    valid = O.compress(np.concatenate([rand_points(14, 200), rand_points(15, 100, subgroup=True), torsion_points(golden)]))
    junk = rng.integers(0, 256, size=(300, 32), dtype=np.uint8)
    special = np.stack([b32(Q), b32(Q - 1), b32(Q + 1), b32((1 << 255) | 1), b32(0), b32((1 << 256) - 1)])
    E = np.concatenate([valid, junk, special, np.array(golden["zip216_noncanonical"]["encodings"], np.uint8)])
    for flags in (0, 1, 1 | 2, 1 | 4, 1 | 8, 1 | 2 | 4 | 8, 2 | 8):
        out, ok = eng.decompress(E, flags)
        eo, ek = O.decompress(E, flags)
        assert (ok == ek).all(), flags
        assert (out == eo).all(), flags
    assert (eng.compress(O.decompress(valid, 0)[0]) == valid).all()


def test_batch_normalize(eng):
    S = rand_scalars(16, 700)
    P = rand_points(17, 700)
    ext = O.varbase_mul_ext(S, P)
    assert (eng.batch_normalize(ext) == O.batch_normalize(ext)).all()
    assert (eng.batch_normalize(ext[:5]) == O.batch_normalize(ext[:5])).all()


def test_torch_device_tensors(eng):
    torch = pytest.importorskip("torch")
    S = rand_scalars(18, 4096)
    P = rand_points(19, 4096)
    dS, dP = torch.from_numpy(S).cuda(), torch.from_numpy(P).cuda()
    out = eng.varbase_mul(dS, dP)
    assert out.is_cuda and out.shape == (4096, 64)
    assert (out.cpu().numpy() == O.varbase_mul(S, P)).all()
    a = eng.field_binary("fq", "mul", dS, dS)
    assert (a.cpu().numpy() == O.field_op(O.FQ, "mul", S, S)[0]).all()
    with pytest.raises(Exception):
        eng.varbase_mul(dS.flatten()[1:1 + 32 * 8].reshape(8, 32), dP[:8])   # misaligned device pointer is rejected


def test_committed_oracle_vectors_gpu(eng):
    """GPU path against the committed fixtures tests/golden/oracle_vectors.json (no oracle call needed)."""
    import json, os
    v = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.json")))
    fx = lambda h: np.frombuffer(bytes.fromhex(h), dtype=np.uint8)
    S = np.stack([fx(c["scalar"]) for c in v["varbase"]])
    P = np.stack([fx(c["point"]) for c in v["varbase"]])
    assert (eng.varbase_mul(S, P) == np.stack([fx(c["out"]) for c in v["varbase"]])).all()
    assert (eng.varbase_mul_exact(S, P) == np.stack([fx(c["ext"]) for c in v["varbase"]])).all()
    fb = v["fixedbase"]
    tab = eng.fixedbase_table(fx(fb["base"]))
    S = np.stack([fx(c["scalar"]) for c in fb["cases"]])
    assert (eng.fixedbase_mul(tab, S) == np.stack([fx(c["out"]) for c in fb["cases"]])).all()
    E = np.stack([fx(c["in"]) for c in v["decompress"]])
    for flags in (0, 1, 3, 5, 9, 15):
        out, ok = eng.decompress(E, flags)
        assert list(ok) == [c["f%d" % flags]["ok"] for c in v["decompress"]], flags
        assert (out == np.stack([fx(c["f%d" % flags]["out"]) for c in v["decompress"]])).all(), flags
    for m in v["msm"]:
        got = eng.msm(np.stack([fx(s) for s in m["scalars"]]), np.stack([fx(p) for p in m["points"]]))
        assert (got == fx(m["out"])).all()


def test_reference_style_api(eng, golden):
    """The reference's test_mul_consistency / test_assoc / test_zip_216 written against jubjub_amd.group
    (reference src/lib.rs:1504-1527, 1756-1804, 1892-1935)."""
    from jubjub_amd.group import FixedBase, Fr, Points, msm

    tp = golden["TEST_POINT_raw"]
    from conftest import limbs
    p = Points(eng, arr64([(limbs(tp["u"]) % Q, limbs(tp["v"]) % Q)])).mul_by_cofactor()
    assert p.is_on_curve().all()
    k1, k2 = Fr.from_u64(eng, [1000]), Fr.from_u64(eng, [3938])
    assert (p * k1) * k2 == p * (k1 * k2)
    t = golden["fr_mul_consistency_mont"]
    a, b, c = (Fr(eng, arr32([J.FR.from_mont_limbs([int(x, 16) for x in t[k]])])) for k in "abc")
    assert a * b == c
    assert p * c == (p * a) * b
    fb = FixedBase(eng, p.data[0])
    assert fb * c == (p * a) * b and p * c == (fb * a) * b
    assert msm(Points(eng, np.concatenate([p.data, p.data])), Fr(eng, np.concatenate([a.data, b.data]))) == p * (a + b)
    for enc in golden["zip216_noncanonical"]["encodings"]:
        e = np.array([enc], np.uint8)
        assert not Points.from_bytes(eng, e)[1].any()
        pts, ok = Points.from_bytes(eng, e, zip216=False)
        assert ok.all()
        re = pts.to_bytes()
        re[:, 31] |= 0x80
        assert (re == e).all()
    with pytest.raises(ValueError):
        Points.generator(eng, 2) * Fr.from_u64(eng, [1])
    g = Points.generator(eng)
    assert not g.is_torsion_free().any() and g.clear_cofactor().is_prime_order().all()
    sub, ok = Points.from_bytes(eng, g.to_bytes(), subgroup=True)
    assert not ok.any() and (sub.data == 0).all()


def test_host_buffer_pipeline(monkeypatch):
    """Large batches handed over as HOST buffers go through the chunked, double-buffered copy/compute pipeline; the
    result must equal the oracle (small chunks, many slots reuses, ragged tail) for every pipelined entry point."""
    from jubjub_amd import Engine

    opts = {}
    opts['pipe_chunk_log2'] = 10
    e2 = Engine(0, options=opts)
    for n in (2048, 2049, 5000, 7 * 1024 + 1):
        S, P = rand_scalars(60 + n, n, full_width=True), rand_points(61 + n, n)
        want = O.varbase_mul(S, P)
        assert (e2.varbase_mul(S, P) == want).all(), n
        assert (e2.varbase_mul_compressed(S, P) == O.compress(want)).all(), n
        for wbits in (0, 10):
            tab = e2.fixedbase_table(P[0], wbits)
            fw = O.fixedbase_mul(S, P[0])
            assert (e2.fixedbase_mul(tab, S) == fw).all(), (n, wbits)
            assert (e2.fixedbase_mul_compressed(tab, S) == O.compress(fw)).all(), (n, wbits)
            tab.close()
    e2.close()


def test_host_buffer_pipeline_default_chunks(eng):
    torch = pytest.importorskip("torch")
    n = (1 << 19) + 12345                                   # three default-size chunks with a ragged tail
    S, P = rand_scalars(70, n), rand_points(71, n)
    got = eng.varbase_mul(S, P)                             # host path (pipelined)
    dev = eng.varbase_mul(torch.from_numpy(S).cuda(), torch.from_numpy(P).cuda()).cpu().numpy()   # device path
    assert (got == dev).all()
    idx = np.arange(0, n, 257)
    assert (got[idx] == O.varbase_mul(S[idx], P[idx])).all()


def test_synthetic_input_generators(eng):
    """jj_synth_scalars / jj_random_points (Group::random semantics, reference src/lib.rs:1244-1267, 1290-1298) against the
    oracle's restatement of the same counter-based streams; any index is reproducible on any device."""
    n, first = 300, 12345
    s = eng.synth_scalars(n, J.SEED, first)
    assert (s == arr32([J.synth_scalar(first + i) for i in range(n)])).all()
    assert all(to_int(r) < R for r in s)
    pts = eng.random_points(n, J.SEED, first)
    want = [J.synth_point(first + i) for i in range(n)]
    assert (pts == arr64([p for p, _ in want])).all()
    assert max(t for _, t in want) >= 5                      # the rejection loop is exercised well past the first draw
    assert eng.predicate("is_on_curve", pts).all() and not eng.predicate("is_identity", pts).any()
    sub = eng.random_points(64, J.SEED ^ 0x55, 7, subgroup=True)
    assert (sub == arr64([J.synth_point(7 + i, seed=J.SEED ^ 0x55, subgroup=True)[0] for i in range(64)])).all()
    assert eng.predicate("is_prime_order", sub).all()
    # a later window of the same stream, produced independently, equals the corresponding slice
    assert (eng.random_points(50, J.SEED, first + 100) == pts[100:150]).all()
    assert eng.synth_scalars(0, 1).shape == (0, 32) and eng.random_points(0, 1).shape == (0, 64)
    raw = eng.synth_bytes32(n, 99, first)
    assert raw.tobytes() == b"".join(J.synth_bytes32(first + i, 99) for i in range(n))
    import torch

    dev = torch.device("cuda", 0)
    assert (eng.random_points(n, J.SEED, first, device=dev).cpu().numpy() == pts).all()
    assert (eng.synth_scalars(n, J.SEED, first, device=dev).cpu().numpy() == s).all()


@pytest.mark.parametrize("fname,p", [("fq", Q), ("fr", R)])
def test_to_le_bits(eng, fname, p):
    """PrimeFieldBits::to_le_bits / char_le_bits (reference src/fr.rs:746-785): bits of the canonical integer, LSB first"""
    a, _ = field_inputs(p, 91, n=200)
    bits = eng.to_le_bits(fname, a)
    assert bits.shape == (len(a), 256)
    want = np.array([[(to_int(x) % p >> b) & 1 for b in range(256)] for x in a], dtype=np.uint8)
    assert (bits == want).all()
    assert eng.to_le_bits(fname, a[:0]).shape == (0, 256)
    assert (eng.char_le_bits() == np.array([(R >> b) & 1 for b in range(256)], dtype=np.uint8)).all()


def test_empty_batches_with_validity_outputs(eng):
    """n = 0 is accepted by the entry points that also return a validity array"""
    e32 = np.zeros((0, 32), np.uint8)
    out, ok = eng.decompress(e32, 1)
    assert out.shape == (0, 64) and ok.shape == (0,)
    for f in ("fq", "fr"):
        for op in ("invert", "sqrt", "from_bytes"):
            o, k = eng.field_unary_ok(f, op, e32)
            assert o.shape == (0, 32) and k.shape == (0,)


@pytest.mark.parametrize("wbits", [13, 14, 16])
def test_fixedbase_wide_windows(eng, wbits):
    """the wide-window tables behind the published fixed_base_wide_window figure (16-bit: 64 MB table)"""
    s = np.concatenate([arr32(EDGE_SCALARS), rand_scalars(300 + wbits, 2000), rand_scalars(400 + wbits, 48, full_width=True)])
    base = pt64(J.GENERATOR)
    tab = eng.fixedbase_table(base, wbits)
    assert (eng.fixedbase_mul(tab, s) == O.fixedbase_mul(s, base)).all()
    tab.close()


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_multi_device_c_abi(devices):
    """jj_multi_*: one context + host thread per listed device (here the same GPU listed several times), contiguous shards,
    MSM partial points folded on the host -- against the oracle and against ragged / empty / fewer-units-than-devices batches"""
    from jubjub_amd import MultiEngine

    me = MultiEngine(devices)
    assert me.device_count == len(devices)
    base = pt64(J.GENERATOR)
    tab = me.fixedbase_table(base)
    for n in (0, 1, 2, 1001):
        s, p = rand_scalars(50 + n, n), rand_points(60 + n, n)
        assert (me.varbase_mul(s, p) == O.varbase_mul(s, p)).all()
        assert (me.fixedbase_mul(tab, s) == O.fixedbase_mul(s, base)).all()
        assert (me.msm(s, p) == O.msm(s, p)).all()
        enc = O.compress(p) if n else np.zeros((0, 32), np.uint8)
        if n:
            enc[::5] ^= 0x40
        out, ok = me.decompress(enc, 1 | 4 | 8)
        eo, ek = O.decompress(enc, 1 | 4 | 8)
        assert (ok == ek).all() and (out == eo).all()
    import torch

    with pytest.raises(Exception):                       # device pointers are refused: the batch is cut on the host
        me._check(me._lib.jj_multi_varbase_mul(me._h, 4, torch.zeros(128, dtype=torch.uint8, device="cuda").data_ptr(), None, None))
    me.close()


def test_context_is_thread_safe_and_stream_switches_are_ordered(eng):
    """two host threads share one context (entry points serialise on the context lock); an asynchronous call on a torch
    stream followed by a call on the context's own stream must not corrupt the first one's workspaces"""
    import threading

    import torch

    s, p = rand_scalars(71, 20000), rand_points(72, 20000)
    want = O.varbase_mul(s, p)
    res = [None, None]

    def work(k):
        res[k] = eng.varbase_mul(s, p) if k == 0 else eng.fixedbase_mul(tab, s)

    tab = eng.fixedbase_table(pt64(J.GENERATOR))
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert (res[0] == want).all() and (res[1] == O.fixedbase_mul(s, pt64(J.GENERATOR))).all()
    dev = torch.device("cuda", 0)
    st, pt = torch.from_numpy(s).to(dev), torch.from_numpy(p).to(dev)
    side = torch.cuda.Stream(dev)
    for _ in range(3):
        with torch.cuda.stream(side):
            a = eng.varbase_mul(st, pt)                    # asynchronous on `side`
        b = eng.varbase_mul(s[:3000], p[:3000])            # numpy: the context's own stream, same workspaces
        side.synchronize()
        assert (a.cpu().numpy() == want).all() and (b == want[:3000]).all()
    tab.close()


def test_group_mirror_random_and_bits(eng):
    """jubjub_amd.group: Group::random / Field::random / to_le_bits under the crate's names"""
    from jubjub_amd import group as G

    k = G.Fr.random(eng, 40, 7, first_index=3)
    lo_seed, hi_seed = G.random_stream_seeds(7)
    wide = lambda i: int.from_bytes(J.synth_bytes32(3 + i, lo_seed) + J.synth_bytes32(3 + i, hi_seed), "little")   # 64 PRNG bytes, from_bytes_wide
    assert (k.data == arr32([wide(i) % R for i in range(40)])).all()
    bits = k.to_le_bits()
    assert all(int("".join(str(b) for b in row[::-1]), 2) == to_int(x) for row, x in zip(bits, k.data))
    p = G.Points.random(eng, 40, 9, subgroup=True)
    assert p.is_prime_order().all() and p.is_on_curve().all()
    q = G.Fq.random(eng, 8, 5)
    lo_seed, hi_seed = G.random_stream_seeds(5)
    assert (q.data == arr32([int.from_bytes(J.synth_bytes32(i, lo_seed) + J.synth_bytes32(i, hi_seed), "little") % Q for i in range(8)])).all()
    assert lo_seed != hi_seed and G.random_stream_seeds(6) != (lo_seed, hi_seed)


def test_config0_shape_on_the_gpu(eng):
    """BASELINE.json configs[0] (the reference's own CPU benches, benches/fq_bench.rs:25-33 and point_bench.rs:6-11) as a parity case:
    1024 random Fq pairs (64 PRNG bytes each through from_bytes_wide, as tests/common.rs:15-21) multiplied, and 1024 random
    extended points doubled, GPU vs the oracle; the CPU timing of the same shape is tests/config1_cpu.py (profiles/r3_config1_cpu.txt)."""
    rng = np.random.default_rng(1024)
    wide_a, wide_b = rng.integers(0, 256, size=(1024, 64), dtype=np.uint8), rng.integers(0, 256, size=(1024, 64), dtype=np.uint8)
    a, b = eng.from_bytes_wide("fq", wide_a), eng.from_bytes_wide("fq", wide_b)
    assert (a == O.from_bytes_wide(O.FQ, wide_a)).all() and (b == O.from_bytes_wide(O.FQ, wide_b)).all()
    assert (eng.field_binary("fq", "mul", a, b) == O.field_op(O.FQ, "mul", a, b)[0]).all()
    pts = rand_points(1025, 1024)
    assert (eng.point_double(pts) == O.point_op("double", pts)).all()


@pytest.mark.parametrize("fold_min", [2, 8])
def test_msm_allgather_failing_rank_posts_a_poison_record(fold_min):
    """A rank that fails BEFORE its all-gather (here: more terms than one pass takes) must not leave the others waiting in theirs: it still
    gathers -- an all-zero record -- and returns its error; every healthy rank's fold / host tail rejects the set ("damaged"), so every
    rank of the collective reports a failure instead of hanging or summing without the lost terms.  The communicator stays usable.
    (Ranks played on one GPU by tests/util.py LoopbackComm, which counts the gathers.)"""
    import torch

    from jubjub_amd import Engine
    from jubjub_amd.engine import JubjubError
    from util import LoopbackComm

    G, n = 4, 3000
    dev = torch.device("cuda", 0)
    S, P = rand_scalars(6100, G * n, full_width=True), rand_points(6200, G * n)
    Sd, Pd = torch.from_numpy(S).to(dev), torch.from_numpy(P).to(dev)
    good = Engine(0, options={"msm_fold_min": fold_min})
    recs = torch.stack([good.msm_partial(Sd[g * n:(g + 1) * n], Pd[g * n:(g + 1) * n]) for g in range(G)])
    # the failing rank (1): its pass limit is below its term count
    bad = Engine(0, options={"msm_pass_log2": 10})
    comm = LoopbackComm(1, G)
    comm.add_round(recs)
    bad.set_comm(comm)
    with pytest.raises(JubjubError, match="at most one pass"):
        bad.msm_allgather(Sd[n:2 * n], Pd[n:2 * n])
    assert comm.calls() == 1                                       # it took part in the collective all the same
    with pytest.raises(JubjubError, match="at most one pass"):
        bad.msm_allgather_begin(Sd[n:2 * n], Pd[n:2 * n])
    assert comm.calls() == 2
    bad.set_comm(None)
    comm.close()
    bad.close()
    # a healthy rank (0) of the same collective: rank 1's slot holds the poison record
    poisoned = recs.clone()
    poisoned[1].zero_()
    comm = LoopbackComm(0, G)
    comm.add_round(poisoned)
    comm.add_round(poisoned)
    comm.add_round(recs)
    good.set_comm(comm)
    assert good.get_option("msm_lanes") == 1                       # several ranks: all gathers of the communicator on one stream
    with pytest.raises(JubjubError, match="damaged"):
        good.msm_allgather(Sd[:n], Pd[:n])
    job = good.msm_allgather_begin(Sd[:n], Pd[:n])
    with pytest.raises(JubjubError, match="damaged"):
        good.msm_finish(job)
    assert (good.msm_allgather(Sd[:n], Pd[:n]) == O.msm(S, P).reshape(64)).all()      # the next collective of the same communicator is fine
    assert comm.calls() == 3
    good.set_comm(None)
    good.set_option("msm_lanes", 3)                                # a caller that has checked multi-stream gathers on its node may go back
    assert good.get_option("msm_lanes") == 3
    comm.close()
    good.close()


def test_options_by_key_and_no_environment(monkeypatch):
    """jj_ctx_set_option / jj_ctx_get_option: every documented key round-trips, ranges are enforced, and the variables rounds 2-5 read from the
    environment -- among them the two that switched the TIMING DISCIPLINE of an entry point -- change nothing: the context created under them
    has the default options and its constant-time entry points give the oracle's results."""
    from jubjub_amd import Engine
    from jubjub_amd.engine import JubjubError

    for k, v in {"JJ_VARBASE_DEFAULT": "vartime", "JJ_FIXEDBASE_SELECT": "gather", "JJ_MSM_LANES": "4", "JJ_MSM_WINDOWS": "30", "JJ_MSM_SMALL_MAX": "0",
                 "JJ_TORSION_CHECK": "ladder", "JJ_MSM_FOLD_MIN": "2", "JJ_VB_QUAD_MAX": "0"}.items():
        monkeypatch.setenv(k, v)
    e = Engine(0)
    defaults = {"msm_lanes": 3, "msm_windows": 0, "msm_small_max": 1 << 14, "torsion_check_ladder": 0, "msm_fold_min": 8, "vb_quad_max": 32768, "msm_front1": 1, "msm_acc_lds": 1,
                "msm_chunk_waves": 2, "fixedbase_default": 7, "pipe_pageable_register": 0, "msm_fold_dev": 1, "msm_host_split": 1, "msm_pass_log2": 24}
    for k, v in defaults.items():
        assert e.get_option(k) == v, k
    for k, v in {"msm_lanes": 4, "msm_fold_min": 2, "msm_windows": 23, "result_pool_mb": 16, "pipe_chunk_log2": 12, "msm_reduce_l1": 8, "msm_accum": -1}.items():
        e.set_option(k, v)
        assert e.get_option(k) == v, k
    for k, v in {"msm_lanes": 5, "msm_fold_min": 1, "msm_reduce_l1": 3, "dec_c_mid": 12, "msm_chunk": 4, "pipe_chunk_log2": 5}.items():
        with pytest.raises(JubjubError, match="out of range"):
            e.set_option(k, v)
    for k in ("varbase_default", "fixedbase_select", "JJ_MSM_LANES", ""):
        with pytest.raises(JubjubError, match="unknown key"):
            e.set_option(k, 1)
    e.close()
    e = Engine(0)
    S, P = rand_scalars(6300, 700, full_width=True), rand_points(6301, 700)
    assert (e.varbase_mul(S, P) == O.varbase_mul(S, P)).all()
    tab = e.fixedbase_table(pt64(J.GENERATOR), 0)
    assert (e.fixedbase_mul(tab, S) == O.fixedbase_mul(S, pt64(J.GENERATOR))).all()
    tab.close()
    assert (e.msm(S, P) == O.msm(S, P).reshape(64)).all()
    e.close()
