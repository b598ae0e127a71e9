"""
ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/libjj_oracle.so (the plain-C CPU
restatement in oracle/jubjub_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libjj_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "jubjub_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libjj_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        rc = _lib.jjo_selftest()
        if rc != 0:
            raise RuntimeError("oracle constants self-test failed: %d" % rc)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(a, width=None):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if width is not None:
        a = a.reshape(-1, width)
    return a


FQ, FR = 0, 1
OPS = {"add": 0, "sub": 1, "mul": 2, "neg": 3, "square": 4, "double": 5, "invert": 6, "sqrt": 7}


def field_op(which, op, a, b=None):
    a = _u8(a, 32)
    n = a.shape[0]
    b = _u8(b, 32) if b is not None else None
    out = np.zeros((n, 32), np.uint8)
    ok = np.zeros(n, np.uint8)
    lib().jjo_field_op(which, OPS[op], C.c_size_t(n), _p(a), _p(b), _p(out), _p(ok))
    return out, ok


def from_bytes(which, a):
    a = _u8(a, 32)
    out = np.zeros_like(a)
    ok = np.zeros(a.shape[0], np.uint8)
    lib().jjo_from_bytes(which, C.c_size_t(a.shape[0]), _p(a), _p(out), _p(ok))
    return out, ok


def from_bytes_wide(which, a):
    a = _u8(a, 64)
    out = np.zeros((a.shape[0], 32), np.uint8)
    lib().jjo_from_bytes_wide(which, C.c_size_t(a.shape[0]), _p(a), _p(out))
    return out


def mont_mul(which, a_limbs, b_limbs):
    a = np.array(a_limbs, dtype=np.uint64)
    b = np.array(b_limbs, dtype=np.uint64)
    o = np.zeros(4, np.uint64)
    lib().jjo_mont_mul(which, _p(a), _p(b), _p(o))
    return [int(x) for x in o]


def to_mont(which, b32):
    a = _u8(b32)
    o = np.zeros(4, np.uint64)
    lib().jjo_to_mont(which, _p(a), _p(o))
    return [int(x) for x in o]


def from_mont(which, limbs):
    a = np.array(limbs, dtype=np.uint64)
    o = np.zeros(32, np.uint8)
    lib().jjo_from_mont(which, _p(a), _p(o))
    return bytes(o)


def varbase_mul(scalars, points):
    s, p = _u8(scalars, 32), _u8(points, 64)
    out = np.zeros((s.shape[0], 64), np.uint8)
    lib().jjo_varbase_mul(C.c_size_t(s.shape[0]), _p(s), _p(p), _p(out))
    return out


def varbase_mul_ext(scalars, points):
    s, p = _u8(scalars, 32), _u8(points, 64)
    out = np.zeros((s.shape[0], 160), np.uint8)
    lib().jjo_varbase_mul_ext(C.c_size_t(s.shape[0]), _p(s), _p(p), _p(out))
    return out


def fixedbase_mul(scalars, base):
    s, b = _u8(scalars, 32), _u8(base, 64)
    out = np.zeros((s.shape[0], 64), np.uint8)
    lib().jjo_fixedbase_mul(C.c_size_t(s.shape[0]), _p(s), _p(b), _p(out))
    return out


POINT_OPS = {"double": 0, "add": 1, "sub": 2, "neg": 3, "mul_by_cofactor": 4}


def point_op(op, pa, pb=None):
    pa = _u8(pa, 64)
    pb = _u8(pb, 64) if pb is not None else None
    out = np.zeros_like(pa)
    lib().jjo_point_op(POINT_OPS[op], C.c_size_t(pa.shape[0]), _p(pa), _p(pb), _p(out))
    return out


def to_niels(pa):
    pa = _u8(pa, 64)
    out = np.zeros((pa.shape[0], 96), np.uint8)
    lib().jjo_to_niels(C.c_size_t(pa.shape[0]), _p(pa), _p(out))
    return out


PREDICATES = {"is_identity": 0, "is_small_order": 1, "is_torsion_free": 2, "is_prime_order": 3, "is_on_curve": 4}


def predicate(what, pa):
    pa = _u8(pa, 64)
    out = np.zeros(pa.shape[0], np.uint8)
    lib().jjo_predicate(PREDICATES[what], C.c_size_t(pa.shape[0]), _p(pa), _p(out))
    return out


def compress(pa):
    pa = _u8(pa, 64)
    out = np.zeros((pa.shape[0], 32), np.uint8)
    lib().jjo_compress(C.c_size_t(pa.shape[0]), _p(pa), _p(out))
    return out


FLAG_ZIP216, FLAG_TORSION_FREE, FLAG_NOT_SMALL_ORDER, FLAG_CLEAR_COFACTOR = 1, 2, 4, 8


def decompress(enc, flags=FLAG_ZIP216):
    enc = _u8(enc, 32)
    out = np.zeros((enc.shape[0], 64), np.uint8)
    ok = np.zeros(enc.shape[0], np.uint8)
    lib().jjo_decompress(C.c_size_t(enc.shape[0]), _p(enc), int(flags), _p(out), _p(ok))
    return out, ok


def batch_from_bytes(enc):
    enc = _u8(enc, 32)
    n = enc.shape[0]
    out = np.zeros((n, 64), np.uint8)
    ok = np.zeros(n, np.uint8)
    scratch = np.zeros(max(n, 1) * 64, np.uint8)
    lib().jjo_batch_from_bytes(C.c_size_t(n), _p(enc), _p(out), _p(ok), _p(scratch))
    return out, ok


def batch_normalize(ext160):
    e = _u8(ext160, 160)
    n = e.shape[0]
    out = np.zeros((n, 64), np.uint8)
    scratch = np.zeros(max(n, 1) * 64, np.uint8)
    lib().jjo_batch_normalize(C.c_size_t(n), _p(e), _p(out), _p(scratch))
    return out


def msm(scalars, points):
    s, p = _u8(scalars, 32), _u8(points, 64)
    out = np.zeros(64, np.uint8)
    lib().jjo_msm(C.c_size_t(s.shape[0]), _p(s), _p(p), _p(out))
    return out


def msm_pippenger(scalars, points, window_bits=13):
    """the same sum by the bucket method (CPU baseline of the MSM workload; jjo_msm_pippenger)"""
    s, p = _u8(scalars, 32), _u8(points, 64)
    out = np.zeros(64, np.uint8)
    rc = lib().jjo_msm_pippenger(C.c_size_t(s.shape[0]), _p(s), _p(p), C.c_int(window_bits), _p(out))
    if rc:
        raise RuntimeError("jjo_msm_pippenger failed: %d" % rc)
    return out


def point_sum(points):
    p = _u8(points, 64)
    out = np.zeros(64, np.uint8)
    lib().jjo_sum(C.c_size_t(p.shape[0]), _p(p), _p(out))
    return out
