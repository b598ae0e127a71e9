#!/usr/bin/env python3
"""Where the time of ONE synchronous 2^k-term MSM call goes on the host: enqueue (jj_msm_begin returns), wait + host tail (jj_msm_finish),
against the HIP-event span of the kernels.  python experiments/misc/msm_host_timing.py [log2n] [key=value options ...]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from jubjub_amd import Engine  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 17
opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[2:]}
n = 1 << lg
eng = Engine(0, options=dict(opts, msm_lanes=1))
dev = torch.device("cuda", 0)
S = eng.synth_scalars(n, 5, 0, device=dev)
P = eng.random_points(n, 5, 0, subgroup=False, device=dev)
eng.sync()
lib, ctx = eng._lib, eng._ctx
out = np.empty(64, np.uint8)
h = C.c_void_p()
sp, pp = C.c_void_p(S.data_ptr()), C.c_void_p(P.data_ptr())
for _ in range(20):
    assert lib.jj_msm_begin(ctx, C.c_size_t(n), sp, pp, C.byref(h)) == 0
    assert lib.jj_msm_finish(h, out.ctypes.data) == 0
tb, tf, tc = [], [], []
for _ in range(200):
    t0 = time.perf_counter()
    lib.jj_msm_begin(ctx, C.c_size_t(n), sp, pp, C.byref(h))
    t1 = time.perf_counter()
    lib.jj_msm_finish(h, out.ctypes.data)
    t2 = time.perf_counter()
    tb.append(t1 - t0); tf.append(t2 - t1); tc.append(t2 - t0)
med = lambda v: sorted(v)[len(v) // 2] * 1e6  # noqa: E731
print("2^%d terms %s: jj_msm_begin returns after %.1f us, jj_msm_finish %.1f us, whole call %.1f us (medians of 200)" % (lg, opts, med(tb), med(tf), med(tc)))
t0 = time.perf_counter()
for _ in range(200):
    lib.jj_msm(ctx, C.c_size_t(n), sp, pp, out.ctypes.data)
print("jj_msm back to back: %.1f us per call" % ((time.perf_counter() - t0) / 200 * 1e6))
