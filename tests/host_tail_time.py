#!/usr/bin/env python3
"""Time of the MSM host tail (jj_msm_combine: window sums of `records` records added window by window, Horner over the windows, one inversion)
on this host's CPU, per window layout W (16: 2^20 terms and more, 17, 23: Pippenger below, 64: small batches) -- no GPU involved.
`scalar` as the first argument forces the scalar 4 x 64-bit chain (jj_ctx_set_option(NULL, "host_tail_scalar", 1)); the default takes the AVX-512 IFMA
chain when the CPU has it.
Usage: python tests/host_tail_time.py [scalar] [records ...]"""
import ctypes
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")   # (under tests/: it checks every timed result against the oracle)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from jubjub_amd import _lib  # noqa: E402
from oracle import c_oracle as O  # noqa: E402
from util import oracle_msm_record, rand_points, rand_scalars  # noqa: E402

lib = _lib.load()
args = sys.argv[1:]
scalar = bool(args) and args[0] == "scalar"
if scalar:
    args = args[1:]
    assert lib.jj_ctx_set_option(None, b"host_tail_scalar", 1) == 0
counts = [int(a) for a in args] or [1, 8]
flags = [w for w in open("/proc/cpuinfo").read().split("flags", 1)[1].split("\n", 1)[0].split() if w in ("avx512ifma", "avx512vl", "adx", "bmi2")]
print("# cpu: %s | %s | host tail: %s" % ([ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0], " ".join(sorted(flags)), "scalar chain (forced)" if scalar else "default"))
for G in counts:
    for W in (16, 17, 23, 64):
        n = 24 * G
        s, p = rand_scalars(1, n), rand_points(2, n)
        cut = [n * g // G for g in range(G + 1)]
        recs = np.ascontiguousarray(np.stack([oracle_msm_record(s[cut[g]:cut[g + 1]], p[cut[g]:cut[g + 1]], 0, 1, W) for g in range(G)]))
        out = np.zeros(64, np.uint8)
        want = O.msm(s, p).reshape(64)
        call = lambda: lib.jj_msm_combine(ctypes.c_size_t(G), recs.ctypes.data, out.ctypes.data)  # noqa: E731
        for _ in range(100):
            call()
        best = 1e9
        for _ in range(7):
            t = time.perf_counter()
            for _ in range(400):
                call()
            best = min(best, (time.perf_counter() - t) / 400)
        print("records %d  W=%2d: %6.1f us per jj_msm_combine (best of 7 x 400), equal to the oracle: %s" % (G, W, best * 1e6, bool((out == want).all())))
