#!/usr/bin/env python3
"""ONE process that runs the library's roofline probe (k_peak_mad, 6 launches) and the constant-time ladder (k_varbase_ct3, 2^20 units, 3 launches),
for tools/peak_clock.sh to count under rocprofv3: do the two run at the same clock?"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from jubjub_amd import Engine  # noqa: E402

eng = Engine(0)
dev = torch.device("cuda", 0)
n = 1 << 20
S = eng.synth_scalars(n, 3, 0, device=dev)
P = eng.random_points(n, 3, 0, subgroup=False, device=dev)
out = torch.empty((n, 64), dtype=torch.uint8, device=dev)
v = (C.c_double * 5)()
for rnd in range(2):
    assert eng._lib.jj_peak_imad32_samples(eng._ctx, 5, v) == 0
    print("k_peak_mad (HIP events): " + " ".join("%.2f" % (x / 1e12) for x in v) + " T mads/s")
    for _ in range(3):
        eng.varbase_mul(S, P, out=out)
    eng.sync()
