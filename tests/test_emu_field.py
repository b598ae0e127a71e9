"""Host emulation of the DEVICE arithmetic (jj_field.h / jj_curve.h compiled for the CPU with -DJJ_HOST_EMU) against the
oracle: the signed 9 x 29-bit Montgomery products, the lazy additions and the point formulas the HIP kernels run, with a
128-bit shadow of every 64-bit column accumulator (any overflow the static checker tools/bounds_check.py missed is
counted).  Test infrastructure only: the product never loads this library."""
import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import jubjub_ref as J
from tests.util import EDGE_SCALARS, Q, R, b32, pt64, to_int, to_pt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "emu_field.cpp")
OUT = os.path.join(ROOT, "tests", "cpp", "libjj_emu.so")
DEPS = [SRC] + [os.path.join(ROOT, "jubjub_amd", "csrc", f) for f in ("jj_field.h", "jj_curve.h", "jj_constants.h")]


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wno-unknown-pragmas", "-shared", "-fPIC", "-o", OUT, SRC])
    lib = ctypes.CDLL(OUT)
    lib.emu_overflow_reset()
    yield lib
    assert lib.emu_overflow_count() == 0, "a 64-bit column accumulator (or a top limb) overflowed in the emulated device arithmetic"


def _buf(n):
    return (ctypes.c_uint8 * n)()


def _in(x):
    return (ctypes.c_uint8 * len(x)).from_buffer_copy(bytes(x))


FIELD_EDGE = lambda p: [0, 1, 2, p - 1, p - 2, p, p + 1, (1 << 256) - 1, (1 << 255), (1 << 255) - 1, 2 * p, 2 * p + 1, (p - 1) // 2, (p + 1) // 2,
                        (1 << 29) - 1, 1 << 29, (1 << 232) - 1, 1 << 232, int("1fffffff" * 8, 16) & ((1 << 256) - 1)]


@pytest.mark.parametrize("name", ["fq", "fr"])
def test_field_ops_emulated(emu, name):
    p = Q if name == "fq" else R
    fn = emu.emu_fq_op if name == "fq" else emu.emu_fr_op
    rng = random.Random(7)
    vals = FIELD_EDGE(p) + [rng.getrandbits(256) for _ in range(200)]
    out, ok = _buf(32), _buf(1)
    ops = {0: lambda a, b: (a + b) % p, 1: lambda a, b: (a - b) % p, 2: lambda a, b: a * b % p, 3: lambda a, b: -a % p,
           4: lambda a, b: a * a % p, 5: lambda a, b: 2 * a % p, 6: lambda a, b: pow(a, -1, p) if a % p else 0,
           7: lambda a, b: 2 * a * a % p, 8: lambda a, b: (2 * a - b) % p}
    for i, a in enumerate(vals):
        b = vals[(7 * i + 3) % len(vals)]
        for op, f in ops.items():
            fn(op, _in(b32(a)), _in(b32(b)), out, ok)          # inputs follow from_raw semantics: reduced mod p
            assert to_int(bytes(out)) == f(a % p, b % p), (name, op, hex(a), hex(b))
            if op == 6:
                assert ok[0] == (1 if a % p else 0)
        fn(9, _in(b32(a)), _in(b32(a % p)), out, ok)
        assert ok[0] == 1
        fn(9, _in(b32(a)), _in(b32((a + 1) % (1 << 256))), out, ok)
        assert ok[0] == (1 if (a % p) == ((a + 1) % (1 << 256)) % p else 0)


@pytest.mark.parametrize("fr", [0, 1])
def test_from_bytes_emulated(emu, fr):
    p = R if fr else Q
    rng = random.Random(11)
    out, ok = _buf(32), _buf(1)
    for a in FIELD_EDGE(p) + [rng.getrandbits(256) for _ in range(50)] + [rng.randrange(p) for _ in range(50)]:
        emu.emu_from_bytes(fr, _in(b32(a)), out, ok)
        assert ok[0] == (1 if a < p else 0)
        assert to_int(bytes(out)) == (a if a < p else 0)
    for _ in range(50):
        w = rng.getrandbits(512)
        emu.emu_from_wide(fr, _in(w.to_bytes(64, "little")), out)
        assert to_int(bytes(out)) == w % p
    emu.emu_from_wide(fr, _in(b"\xff" * 64), out)
    assert to_int(bytes(out)) == ((1 << 512) - 1) % p


def _points(n, seed):
    rng = random.Random(seed)
    pts = []
    while len(pts) < n:
        pts.append(J.ext_to_affine(J.ext_multiply(J.affine_to_extended(J.GENERATOR), b32(rng.getrandbits(252)).tobytes())))
    return pts


def test_varbase_ladder_emulated(emu, golden):
    rng = random.Random(3)
    pts = _points(6, 5)
    tors = [(sum(int(h, 16) << (64 * i) for i, h in enumerate(p["u"])), sum(int(h, 16) << (64 * i) for i, h in enumerate(p["v"])))
            for p in golden["EIGHT_TORSION_raw"]["points"]]
    cases = [(k, pts[i % len(pts)]) for i, k in enumerate(EDGE_SCALARS)] + [(rng.getrandbits(256), pts[i % len(pts)]) for i in range(12)]
    cases += [(rng.getrandbits(252), t) for t in tors] + [(5, (0, 1)), (R, J.GENERATOR)]
    out = _buf(64)
    for k, P in cases:
        emu.emu_varbase(_in(b32(k)), _in(pt64(P)), out)
        want = J.ext_to_affine(J.ext_multiply(J.affine_to_extended(P), b32(k).tobytes()))   # the reference ladder: bits 251..0 of the integer, no reduction mod r
        assert to_pt(np.frombuffer(bytes(out), dtype=np.uint8)) == want, (hex(k), P)


def test_exact_ladder_projective_coordinates_emulated(emu):
    rng = random.Random(9)
    out = _buf(160)
    for P in _points(3, 21):
        k = rng.getrandbits(256)
        emu.emu_varbase_exact(_in(b32(k)), _in(pt64(P)), out)
        want = J.ext_multiply(J.affine_to_extended(P), b32(k).tobytes())
        got = tuple(to_int(bytes(out)[32 * i:32 * i + 32]) for i in range(5))
        assert got == tuple(int(c) for c in want)


def test_signed_sums_and_ext_add_emulated(emu):
    rng = random.Random(13)
    out = _buf(64)
    for n, dbls in ((0, 0), (1, 0), (5, 3), (40, 1), (17, 7)):
        pts = _points(n, 100 + n) if n else []
        signs = bytes(rng.randrange(2) for _ in range(n))
        emu.emu_signed_sum(n, _in(b"".join(pt64(p).tobytes() for p in pts) or b"\0"), _in(signs or b"\0"), dbls, out)
        acc = J.EXT_IDENTITY
        for p, s in zip(pts, signs):
            acc = J.ext_sub_affine(acc, p) if s else J.ext_add_affine(acc, p)
        for _ in range(dbls):
            acc = J.ext_double(acc)
        assert to_pt(np.frombuffer(bytes(out), dtype=np.uint8)) == J.ext_to_affine(acc)


def test_predicates_emulated(emu, golden):
    tors = [(sum(int(h, 16) << (64 * i) for i, h in enumerate(p["u"])), sum(int(h, 16) << (64 * i) for i, h in enumerate(p["v"])))
            for p in golden["EIGHT_TORSION_raw"]["points"]]
    sub = [J.ext_to_affine(J.ext_mul_by_cofactor(J.affine_to_extended(p))) for p in _points(4, 77)]
    mixed = [J.ext_to_affine(J.ext_add_affine(J.affine_to_extended(s), t)) for s, t in zip(sub, tors[1:5])]
    for P in tors + sub + mixed + _points(4, 78):
        e = J.affine_to_extended(P)
        want = (1 if J.affine_is_on_curve(P) else 0) | (2 if J.ext_is_torsion_free(e) else 0) | (4 if J.ext_is_small_order(e) else 0) | \
               (8 if J.ext_is_identity(e) else 0) | (16 if J.ext_is_identity(J.ext_mul_by_cofactor(e)) else 0)
        assert emu.emu_predicates(_in(pt64(P))) == want, P
    assert emu.emu_predicates(_in(pt64((5, 7)))) & 1 == 0          # off the curve


def test_normalise_tail_emulated(emu):
    """k_normalize's per-element arithmetic: plain-form inverse, canon_plain_product, is_zero_product"""
    rng = random.Random(17)
    out = _buf(64)
    pts = _points(5, 31) + [(0, 1), (0, Q - 1)]
    for P in pts:
        for sc in (1, 2, Q - 1, rng.randrange(1, Q), rng.getrandbits(256)):
            flags = emu.emu_normalize(_in(pt64(P)), _in(b32(sc)), out)
            assert flags == (1 if sc % Q == 0 else 0)
            assert to_pt(np.frombuffer(bytes(out), dtype=np.uint8)) == ((P[0] % Q, P[1] % Q) if sc % Q else (0, 0))
        flags = emu.emu_normalize(_in(pt64(P)), _in(b32(Q)), out)          # Z = 0 is skipped like ff's BatchInverter: output zeroed
        assert flags == 1 and to_pt(np.frombuffer(bytes(out), dtype=np.uint8)) == (0, 0)
