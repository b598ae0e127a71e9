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


# ---- MSM records (include/jubjub_hip.h jj_msm_partial / jj_msm_combine), built by the oracle: lets the CPU tests drive the
# host-only jj_msm_combine and the distributed logic without a GPU
MSM_PARTIAL_BYTES = 8256
MSM_REC_MAGIC = 0x504D4A4A


def msm_window_layout(W):
    """(start, width) of the W windows that tile the 253 bits of a recoded scalar: 253 = W c + r, the r low windows are c + 1 bits"""
    c, r = divmod(253, W)
    out, bit = [], 0
    for w in range(W):
        width = c + (1 if w < r else 0)
        out.append((bit, width))
        bit += width
    assert bit == 253
    return out


def msm_signed_digits(k, W):
    """digits d_w with k mod 2^252 = sum_w d_w 2^(start_w): signed for w < W - 1, the top one unsigned"""
    lay = msm_window_layout(W)
    kp = (k & ((1 << 252) - 1)) + sum(1 << (s + wd - 1) for s, wd in lay[:-1])
    ds = []
    for w, (s, wd) in enumerate(lay):
        raw = (kp >> s) & ((1 << wd) - 1)
        ds.append(raw if w == W - 1 else raw - (1 << (wd - 1)))
    assert sum(d << lay[w][0] for w, d in enumerate(ds)) == k & ((1 << 252) - 1)
    return ds


def oracle_msm_record(S, P, part_index=0, part_count=1, W=64):
    """the record jj_msm_partial would leave for these terms: window sums S_w = sum_i d_{i,w} P_i of the windows this part owns, each
    as (U, V, Z, T) = (u, v, 1, u v) in the host tail's Montgomery form (value * 2^256 mod q, 32 little-endian bytes)"""
    n = len(S)
    digs = [msm_signed_digits(to_int(S[i]), W) for i in range(n)]
    pts = [to_pt(P[i]) for i in range(n)]
    rec = np.zeros(MSM_PARTIAL_BYTES, np.uint8)
    mask = 0
    for w in range(part_index, W, part_count):
        mask |= 1 << w
        acc = J.AFFINE_IDENTITY
        for i in range(n):
            d = digs[i][w]
            if d:
                t = J.scalar_mul_fast(pts[i], abs(d))
                acc = J.affine_add_fast(acc, J.affine_neg(t) if d < 0 else t)
        coords = (acc[0], acc[1], 1, acc[0] * acc[1] % Q)
        for k, x in enumerate(coords):
            rec[64 + w * 128 + 32 * k: 64 + w * 128 + 32 * k + 32] = b32((x << 256) % Q)
    hdr = np.array([MSM_REC_MAGIC, 2, W, 1, mask & 0xFFFFFFFF, mask >> 32, n & 0xFFFFFFFF, n >> 32], dtype="<u4")
    rec[:32] = np.frombuffer(hdr.tobytes(), np.uint8)
    return rec


class LoopbackComm:
    """Plays the other ranks of an RCCL communicator on ONE GPU (tools/loopback_comm.cpp) -- what Engine.set_comm takes in place of a
    jubjub_amd.dist.RcclComm: the all-gather puts what THIS rank sends into slot `rank` and, into the other slots, the records the test
    prepared for the other ranks (add_round: a (world, MSM_PARTIAL_BYTES) CUDA tensor per call, used in turn)."""

    def __init__(self, rank, world):
        import ctypes as C
        import os
        import subprocess

        tools = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools")
        so, src = os.path.join(tools, "libloopback_comm.so"), os.path.join(tools, "loopback_comm.cpp")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", "-o", so, src])
        self._lib = C.CDLL(so)
        self._lib.jj_loopback_create.restype = C.c_void_p
        self._lib.jj_loopback_create.argtypes = [C.c_int, C.c_int]
        self._lib.jj_loopback_set.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
        self._lib.jj_loopback_rewind.argtypes = [C.c_void_p]
        self._lib.jj_loopback_calls.argtypes = [C.c_void_p]
        self._lib.jj_loopback_calls.restype = C.c_uint
        self._lib.jj_loopback_destroy.argtypes = [C.c_void_p]
        self.rank, self.world = int(rank), int(world)
        self.handle = self._lib.jj_loopback_create(self.rank, self.world)
        self.all_gather_addr = C.cast(self._lib.jj_loopback_all_gather, C.c_void_p).value
        self._keep = []

    def add_round(self, records):
        assert records.is_cuda and records.is_contiguous() and records.shape[0] == self.world
        assert self._lib.jj_loopback_set(self.handle, len(self._keep), records.data_ptr()) == 0
        self._keep.append(records)

    def calls(self):
        return int(self._lib.jj_loopback_calls(self.handle))

    def close(self):
        if self.handle:
            self._lib.jj_loopback_destroy(self.handle)
            self.handle = None
