#!/usr/bin/env python3
"""Where does the large-input MSM configuration (17 windows, length-sorted segments) overtake the mid-size one (23 windows, chunks +
fix-up)?  Wall time per call for n between 2^16.5 and 2^18, both configurations forced in turn.   python experiments/misc/msm_crossover.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from jubjub_amd import Engine  # noqa: E402

CFG = {"mid (W=23, chunks)": {"JJ_MSM_WINDOWS": "23", "JJ_MSM_ACCUM": "chunks"},
       "large (W=17, segments)": {"JJ_MSM_WINDOWS": "17", "JJ_MSM_ACCUM": "segments"},
       "W=20 segments": {"JJ_MSM_WINDOWS": "20", "JJ_MSM_ACCUM": "segments"},
       "default": {}}
base = Engine(0)
for n in (92000, 110000, 131072, 150000, 165000, 185000, 210000, 235000, 262144):
    S = base.synth_scalars(n, 7, 0, device="cuda:0")
    P = base.random_points(n, 7, 0, subgroup=False, device="cuda:0")
    want = base.msm(S, P).cpu()
    row = []
    for name, env in CFG.items():
        for k in ("JJ_MSM_WINDOWS", "JJ_MSM_ACCUM"):
            os.environ.pop(k, None)
        os.environ.update(env)
        eng = Engine(0)
        for _ in range(3):
            got = eng.msm(S, P)
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            eng.msm(S, P)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        assert bool((got.cpu() == want).all()), (n, name)
        row.append("%s %.3f" % (name, ts[len(ts) // 2]))
        eng.close()
    print("n = %7d: " % n + " | ".join(row) + "   (median ms)")
