#!/usr/bin/env python3
"""One rank's share of a 2^log2n-term MSM cut G ways, on ONE GPU: by terms (all windows of n / G terms) and by
windows (windows g, g + G, ... of all n terms), device work only (jj_msm_partial, record left on the device) and with the host tail
of one record; plus the host tail over G records (what every rank runs after the all_gather).
  python experiments/misc/msm_partition_cost.py [log2n] [G]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from jubjub_amd import Engine  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 1 << log2n
eng = Engine(0)
dev = torch.device("cuda", 0)
S = eng.synth_scalars(n, 7, 0, device=dev)
P = eng.random_points(n, 7, 0, subgroup=False, device=dev)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


m = n // G
print("2^%d terms cut %d ways, one rank's share on one MI355X (median / min ms over 30 calls, device-resident inputs):" % (log2n, G))
print("  whole MSM on one GPU (jj_msm)                        : %.3f / %.3f" % timed(lambda: eng.msm(S, P)))
for g in (0, G - 1):
    print("  term partition, rank %d: all windows of %d terms      : %.3f / %.3f  (record left on the device)" % ((g, m) + timed(lambda: eng.msm_partial(S[g * m:(g + 1) * m], P[g * m:(g + 1) * m]))))
for g in (0, G - 1):
    print("  window partition, rank %d: windows %d, %d.. of all terms : %.3f / %.3f  (record left on the device)" % ((g, g, g + G) + timed(lambda: eng.msm_partial(S, P, g, G))))
recs_t = torch.stack([eng.msm_partial(S[g * m:(g + 1) * m], P[g * m:(g + 1) * m]) for g in range(G)])
recs_w = torch.stack([eng.msm_partial(S, P, g, G) for g in range(G)])
want = eng.msm(S, P).cpu().numpy()
assert (eng.msm_combine(recs_t) == want).all() and (eng.msm_combine(recs_w) == want).all()
print("  copy of %d gathered records to the host + ONE host tail : terms %.3f / %.3f   windows %.3f / %.3f" % ((G,) + timed(lambda: eng.msm_combine(recs_t)) + timed(lambda: eng.msm_combine(recs_w))))
ht, hw = recs_t.cpu().numpy(), recs_w.cpu().numpy()
print("  host tail alone (records already on the host)          : terms %.3f / %.3f   windows %.3f / %.3f" % (timed(lambda: eng.msm_combine(ht)) + timed(lambda: eng.msm_combine(hw))))
print("  both partitions give the point of the one-GPU MSM: ok")
