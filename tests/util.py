"""Shared helpers for the parity tests (oracle-side data generation; test infrastructure only)."""
import random

import numpy as np

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


def rand_scalars(seed, n, full_width=False):
    """n x 32 bytes.  full_width: arbitrary 256-bit patterns (top bits set), else uniform below 2^252."""
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    if not full_width:
        s[:, 31] &= 0x0F
    return s


def rand_points(seed, n, subgroup=False):
    """n random curve points (full group of order 8r unless subgroup) built with the C oracle's ladder."""
    k = rand_scalars(seed ^ 0x5EED, n)
    base = pt64(J.GENERATOR)
    pts = O.fixedbase_mul(k, base)
    if subgroup:
        pts = O.point_op("mul_by_cofactor", pts)
    return pts


def torsion_points(golden):
    return arr64([(sum(int(h, 16) << (64 * i) for i, h in enumerate(p["u"])),
                   sum(int(h, 16) << (64 * i) for i, h in enumerate(p["v"]))) for p in golden["EIGHT_TORSION_raw"]["points"]])


EDGE_SCALARS = [0, 1, 2, 7, 8, 9, 15, 16, 17, R - 1, R, R + 1, (1 << 252) - 1, (1 << 252) - 2, 1 << 251, (1 << 251) - 1,
                int("8" * 63, 16), int("7" * 63, 16), int("f" * 63, 16), (1 << 255) | 5, (1 << 256) - 1, (0xF << 252) | 12345]
