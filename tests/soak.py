#!/usr/bin/env python3
"""Randomised differential soak: every C-ABI entry point against the C oracle on fresh seeds for a wall-clock
budget.  Usage: python tests/soak.py [seconds]   (needs an MI355X).  Prints one summary line per round."""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from jubjub_amd import Engine  # noqa: E402
from oracle import c_oracle as O  # noqa: E402
from oracle import jubjub_ref as J  # noqa: E402
from util import pt64  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
SEED0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
eng = Engine(0)
# second context: the large-input MSM accumulation scheme forced on small inputs, and the per-lane var-base kernel instead of the per-quad one
eng_alt = Engine(0, options={"msm_small_max": 0, "msm_accum": 1, "vb_quad_max": 0})
# Pippenger with 16 / 19 / 23 windows (two-pass and one-pass sort) forced on every size
eng_wide = [Engine(0, options={"msm_small_max": 0, "msm_windows": nwin_msm}) for nwin_msm in (16, 19, 23)]
base = pt64(J.GENERATOR)
G8 = J.scalar_mul_fast(J.GENERATOR, J.R_MOD)             # order-8 component of the generator
TORS = np.stack([pt64(J.scalar_mul_fast(G8, j) if j else J.AFFINE_IDENTITY) for j in range(8)])
t_end = time.time() + budget
rnd = 0
checked = 0
while time.time() < t_end:
    rng = np.random.default_rng(SEED0 + rnd)
    n = int(rng.integers(1, 20000))
    S = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    K = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    K[:, 31] &= 0x0F
    P = O.fixedbase_mul(K, base)
    if rnd % 3 == 0:
        P = O.point_op("mul_by_cofactor", P)
    want_vb = O.varbase_mul(S, P)
    assert (eng.varbase_mul(S, P) == want_vb).all(), ("varbase", rnd)
    assert (eng_alt.varbase_mul(S, P) == want_vb).all(), ("varbase per-lane", rnd)
    assert (eng.varbase_mul_ct(S, P) == want_vb).all(), ("varbase constant-time", rnd)
    bp = P[int(rng.integers(0, n))]
    wbits = [0, 8, 10, 12, 13, 14, 16, 6][rnd % 8]        # comb (0), LDS window table (6) and every wide-window class (16 bits: the 64 MB table)
    tab = eng.fixedbase_table(bp, wbits)
    assert (eng.fixedbase_mul(tab, S) == O.fixedbase_mul(S, bp)).all(), ("fixedbase", rnd, wbits)
    if rnd % 2 == 0:                                          # sums over two fixed bases; one scalar with many bases
        bp2 = P[int(rng.integers(0, n))]
        tab2 = eng.fixedbase_table(bp2, [0, 12][rnd % 4 // 2])
        want2 = O.point_op("add", O.fixedbase_mul(S, bp), O.fixedbase_mul(K, bp2))
        assert (eng.fixedbase_multi_mul([tab, tab2], np.stack([S, K])) == want2).all(), ("multi-base", rnd)
        tab2.close()
        assert (eng.varbase_mul_scalar(S[0], P) == O.varbase_mul(np.repeat(S[:1], n, axis=0), P)).all(), ("shared scalar", rnd)
    tab.close()
    m = n
    want_msm = O.msm(S[:m], P[:m])
    assert (eng.msm(S[:m], P[:m]) == want_msm).all(), ("msm", rnd)
    assert (eng_alt.msm(S[:m], P[:m]) == want_msm).all(), ("msm segments", rnd)
    assert (eng_wide[rnd % 3].msm(S[:m], P[:m]) == want_msm).all(), ("msm wide windows", rnd)
    G = 2 + rnd % 5                                           # the MSM in parts: by windows, by terms, and as overlapping asynchronous jobs
    assert (eng.msm_combine(np.stack([eng.msm_partial(S[:m], P[:m], g, G) for g in range(G)])) == want_msm).all(), ("msm window partition", rnd)
    cut = [m * g // G for g in range(G + 1)]
    assert (eng_alt.msm_combine(np.stack([eng_alt.msm_partial(S[cut[g]:cut[g + 1]], P[cut[g]:cut[g + 1]]) for g in range(G)])) == want_msm).all(), ("msm term partition", rnd)
    jobs = [eng.msm_begin(S[:m >> k], P[:m >> k]) for k in range(3)]
    for k in (1, 0, 2):
        assert (eng.msm_finish(jobs[k]) == O.msm(S[:m >> k], P[:m >> k])).all(), ("msm async", rnd, k)
    if rnd % 2 == 1:                                          # several short-scalar bases through one composite table
        bits = [[64, 64, 64], [124, 124], [40] * 6, [10] * 21][rnd // 2 % 4]
        nbs = len(bits)
        ct = eng.fixedbase_composite_table(P[:nbs], bits)
        q3 = min(n, 2000)
        Sc = np.stack([np.roll(S[:q3], b, axis=0) for b in range(nbs)])
        wantc = None
        for b in range(nbs):
            msk = np.frombuffer(((1 << bits[b]) - 1).to_bytes(32, "little"), dtype=np.uint8)
            term = O.fixedbase_mul(Sc[b] & msk, P[b])
            wantc = term if wantc is None else O.point_op("add", wantc, term)
        assert (eng.fixedbase_composite_mul(ct, Sc) == wantc).all(), ("composite", rnd)
        ct.close()
    enc = O.compress(P)
    bad = rng.integers(0, n, size=max(1, n // 10))
    enc[bad] = rng.integers(0, 256, size=(len(bad), 32), dtype=np.uint8)
    flags = int(rng.choice([0, 1, 3, 5, 9, 13, 15]))
    mm = n
    o1, k1 = eng.decompress(enc[:mm], flags)
    o2, k2 = O.decompress(enc[:mm], flags)
    assert (k1 == k2).all() and (o1 == o2).all(), ("decompress", rnd, flags)
    for f, w in (("fq", O.FQ), ("fr", O.FR)):
        for op in ("add", "sub", "mul"):
            assert (eng.field_binary(f, op, S, K) == O.field_op(w, op, S, K)[0]).all(), (f, op, rnd)
        for op in ("neg", "square", "double"):
            assert (eng.field_unary(f, op, S) == O.field_op(w, op, S)[0]).all(), (f, op, rnd)
        q = min(n, 2000)
        for op in ("invert", "sqrt"):
            a1, b1 = eng.field_unary_ok(f, op, S[:q])
            a2, b2 = O.field_op(w, op, S[:q])
            assert (b1 == b2).all() and (a1 == a2).all(), (f, op, rnd)
    q2 = min(n, 3000)                                         # generators, bit decomposition, several contexts on one device
    canon = O.field_op(O.FR, "add", S[:q2], np.zeros((q2, 32), np.uint8))[0]          # S mod r, canonical bytes
    assert (eng.to_le_bits("fr", S[:q2]) == np.unpackbits(canon, axis=1, bitorder="little")).all(), ("to_le_bits", rnd)
    first = int(rng.integers(0, 1 << 40))
    gp = eng.random_points(64, SEED0 + rnd, first, subgroup=bool(rnd & 1))
    assert (gp == np.stack([pt64(J.synth_point(first + i, SEED0 + rnd, subgroup=bool(rnd & 1))[0]) for i in range(64)])).all(), ("random_points", rnd)
    Qp = O.fixedbase_mul(S, base)
    assert (eng.point_add(P, Qp) == O.point_op("add", P, Qp)).all() and (eng.point_sub(P, Qp) == O.point_op("sub", P, Qp)).all()
    assert (eng.point_double(P) == O.point_op("double", P)).all()
    Pm = P if rnd % 3 else O.point_op("add", P, np.repeat(TORS[rnd % 8:rnd % 8 + 1], n, axis=0))   # shift whole batch into another coset
    for pred in ("is_torsion_free", "is_prime_order", "is_small_order"):
        assert (eng.predicate(pred, Pm) == O.predicate(pred, Pm)).all(), (pred, rnd)
    checked += n
    rnd += 1
    print("round %d ok: n=%d flags=%d window_bits=%d  (%d units so far, %.0f s left)" % (rnd, n, flags, wbits, checked, t_end - time.time()), flush=True)
print("SOAK PASSED: %d rounds, %d units per op family, all bit-exact vs the C oracle" % (rnd, checked))
