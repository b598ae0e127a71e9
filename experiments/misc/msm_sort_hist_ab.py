#!/usr/bin/env python3
"""Two-pass sort of the MSM: coarse histogram inside the conversion kernel + atomic run reservation (round 5 default) against the separate histogram and plan
kernels of round 4 (option msm_sort_hist_fused = 0): wall time per call, alternating, same inputs, results compared.   python experiments/misc/msm_sort_hist_ab.py [log2n ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from jubjub_amd import Engine  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [18, 19, 20, 21, 22]
fused = Engine(0)
sep = Engine(0, options={"msm_sort_hist_fused": 0})        # round 6: a context option (was JJ_MSM_SORT_HIST=separate)
for lg in sizes:
    n = 1 << lg
    S = fused.synth_scalars(n, 7, 0, device="cuda:0")
    P = fused.random_points(n, 7, 0, subgroup=False, device="cuda:0")
    res = {}
    for rep in range(3):
        for name, e in (("fused", fused), ("separate", sep)):
            for _ in range(3):
                got = e.msm(S, P)
            torch.cuda.synchronize()
            ts = []
            for _ in range(25):
                t0 = time.perf_counter()
                e.msm(S, P)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            res.setdefault(name, []).append(ts[len(ts) // 2])
            res[name + "_pt"] = got.cpu()
    ok = bool((res["fused_pt"] == res["separate_pt"]).all())
    print("2^%d  fused %s  separate %s  (median ms of 25 calls, three alternating rounds)  same point: %s" % (
        lg, " ".join("%.4f" % x for x in res["fused"]), " ".join("%.4f" % x for x in res["separate"]), ok), flush=True)
