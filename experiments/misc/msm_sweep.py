#!/usr/bin/env python3
"""MSM tuning sweep in one process: wall time per call for (log2n, env overrides) combinations.  The knobs are read at context
creation, so every configuration gets its own Engine.
  python experiments/misc/msm_sweep.py 17 JJ_MSM_WINDOW=11,12,13,14 JJ_MSM_ACCUM=chunks,segments [JJ_MSM_CHUNK=8,16]"""
import itertools
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from jubjub_amd import Engine  # noqa: E402

log2n = int(sys.argv[1])
axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[2:]]
n = 1 << log2n
base = Engine(0)
S = base.synth_scalars(n, 7, 0, device="cuda:0")
P = base.random_points(n, 7, 0, subgroup=False, device="cuda:0")
want = base.msm(S, P).cpu()
for combo in itertools.product(*[v for _, v in axes]):
    for (k, _), v in zip(axes, combo):
        if v == "-":
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    eng = Engine(0)
    for _ in range(3):
        got = eng.msm(S, P)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        eng.msm(S, P)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("2^%d %s: median %.3f ms  min %.3f ms  %s" % (log2n, " ".join("%s=%s" % (k, v) for (k, _), v in zip(axes, combo)), ts[len(ts) // 2], ts[0],
                                                      "ok" if bool((got.cpu() == want).all()) else "MISMATCH"))
    sys.stdout.flush()
    eng.close()
