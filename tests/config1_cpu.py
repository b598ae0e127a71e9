#!/usr/bin/env python3
"""BASELINE.json configs[0] ("plumbing, no GPU"): 1k random Fq Montgomery multiplications + 1k ExtendedPoint::double
on the CPU port of the reference algorithm (oracle/jubjub_oracle.c; analogue of benches/fq_bench.rs:25-33 and
benches/point_bench.rs:6-11, on random instead of constant inputs).  Prints ns/op, single thread."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import c_oracle as O  # noqa: E402
from oracle import jubjub_ref as J  # noqa: E402

lib = O.lib()
rng = np.random.default_rng(1)
n, reps = 1024, 2000
a = np.ascontiguousarray(rng.integers(0, 256, size=(n, 32), dtype=np.uint8))
b = np.ascontiguousarray(rng.integers(0, 256, size=(n, 32), dtype=np.uint8))
out = np.zeros((n, 32), np.uint8)
p = lambda x: x.ctypes.data_as(C.c_void_p)
lib.jjo_bench_fq_mul(C.c_size_t(n), p(a), p(b), p(out), 10)
t0 = time.perf_counter()
lib.jjo_bench_fq_mul(C.c_size_t(n), p(a), p(b), p(out), reps)
t = time.perf_counter() - t0
print("Fq mul (4x64 Montgomery, u128 mac):   %7.1f ns/op   (%d x %d dependent multiplications, 1 thread)" % (t / (n * reps) * 1e9, n, reps))
base = np.frombuffer(J.GENERATOR[0].to_bytes(32, "little") + J.GENERATOR[1].to_bytes(32, "little"), dtype=np.uint8)
s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
s[:, 31] &= 0x0F
pts = np.ascontiguousarray(O.fixedbase_mul(s, base))
out64 = np.zeros((n, 64), np.uint8)
lib.jjo_bench_double(C.c_size_t(n), p(pts), p(out64), 10)
t0 = time.perf_counter()
lib.jjo_bench_double(C.c_size_t(n), p(pts), p(out64), 200)
t = time.perf_counter() - t0
print("ExtendedPoint::double (4S + 3M):       %7.1f ns/op   (%d x 200 dependent doublings + one to_affine each, 1 thread)" % (t / (n * 200) * 1e9, n))
