#!/usr/bin/env python3
"""Wall time of individual MSM calls (device-resident inputs) at a given size: python experiments/misc/msm_each.py [log2n]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from jubjub_amd import Engine  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
eng = Engine(0)
n = 1 << log2n
S = eng.synth_scalars(n, 7, 0, device="cuda:0")
P = eng.random_points(n, 7, 0, subgroup=False, device="cuda:0")
for _ in range(3):
    eng.msm(S, P)
torch.cuda.synchronize()
ts = []
for _ in range(24):
    t0 = time.perf_counter()
    eng.msm(S, P)
    ts.append((time.perf_counter() - t0) * 1e3)
print("2^%d terms, ms per call:" % log2n, " ".join("%.3f" % t for t in ts))
m = n - 77777
ts = []
for _ in range(8):
    t0 = time.perf_counter()
    eng.msm(S[:m], P[:m])
    ts.append((time.perf_counter() - t0) * 1e3)
print("%d terms, ms per call:" % m, " ".join("%.3f" % t for t in ts))
