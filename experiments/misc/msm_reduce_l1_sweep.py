#!/usr/bin/env python3
"""Two-level bucket reduce (k_msm_reduce_l1 / _l2, round 5): wall time per MSM call for window counts x level-1 rows x level-2 chunk
lengths, each forced in turn on the same device-resident inputs and checked against the default configuration's result.
   python experiments/misc/msm_reduce_l1_sweep.py [log2n ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from jubjub_amd import Engine  # noqa: E402

# (round 6: context options instead of environment variables)
CFG = [("W16 one-level", {"msm_windows": 16, "msm_reduce_l1": 0}),
       ("W17 one-level", {"msm_windows": 17, "msm_reduce_l1": 0}),
       ("default", {})]
for W in (16, 17):
    for R in (4, 8):
        CFG.append(("W%d R%d" % (W, R), {"msm_windows": W, "msm_reduce_l1": R}))
sizes = [int(a) for a in sys.argv[1:]] or [18, 19, 20, 21, 22]
base = Engine(0)
for lg in sizes:
    n = 1 << lg
    S = base.synth_scalars(n, 7, 0, device="cuda:0")
    P = base.random_points(n, 7, 0, subgroup=False, device="cuda:0")
    want = base.msm(S, P).cpu()
    for name, env in CFG:
        eng = Engine(0, options=env)
        for _ in range(3):
            got = eng.msm(S, P)
        torch.cuda.synchronize()
        ts = []
        for _ in range(25):
            t0 = time.perf_counter()
            eng.msm(S, P)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        ok = bool((got.cpu() == want).all())
        print("2^%d  %-16s median %.4f ms  min %.4f  %s" % (lg, name, ts[len(ts) // 2], ts[0], "ok" if ok else "MISMATCH"), flush=True)
        eng.close()
