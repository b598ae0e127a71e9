#!/usr/bin/env python3
"""Mid-size MSMs (2^15 ... 2^19 terms: config 4's per-rank shares): wall time per synchronous jj_msm call for every window count x level-1 rows of the
bucket reduce x accumulation form, each forced in turn on the same device-resident inputs and checked against the default configuration's point.
The planner's choice (jj_msm.hip msm_windows_for: 23 windows below 147 456 terms, 17 from there, 16 from 2^18 (2^20 until round 6); two-level reduce from 16 384
buckets per window) dates from round 3, before the two-level reduce existed.
   python experiments/misc/msm_mid_sweep.py [log2n ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from jubjub_amd import Engine  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [16, 17, 18]
base = Engine(0)                       # (round 6: the planner overrides are context options, include/jubjub_hip.h; the library reads no environment variable)


def timed(eng, S, P, reps=25):
    for _ in range(3):
        got = eng.msm(S, P)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        eng.msm(S, P)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], got


for lg in sizes:
    n = 1 << lg
    S = base.synth_scalars(n, 7, 0, device="cuda:0")
    P = base.random_points(n, 7, 0, subgroup=False, device="cuda:0")
    t_def, want = timed(base, S, P)
    want = want.cpu()
    print("2^%d  default                          %.4f ms" % (lg, t_def), flush=True)
    rows = []
    for W in range(16, 24):
        for R in ("0", "2", "4", "8"):
            for acc in ("chunks", "segments"):
                try:
                    eng = Engine(0, options={"msm_windows": W, "msm_reduce_l1": int(R), "msm_accum": 1 if acc == "segments" else 0})
                    t, got = timed(eng, S, P)
                    ok = bool((got.cpu() == want).all())
                    eng.close()
                except Exception as ex:                      # a combination the planner refuses
                    t, ok = float("inf"), "refused: %s" % str(ex)[:60]
                rows.append((t, W, R, acc, ok))
    rows.sort(key=lambda r: r[0])
    for t, W, R, acc, ok in rows[:12]:
        print("2^%d  W%-2d L1=%s %-8s              %.4f ms  (%.3f of default)  %s" % (lg, W, R, acc, t, t / t_def, "ok" if ok is True else ok), flush=True)
    bad = [r for r in rows if r[4] is not True and r[0] != float("inf")]
    print("2^%d  mismatches: %d of %d" % (lg, len(bad), len(rows)), flush=True)
