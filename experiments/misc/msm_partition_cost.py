#!/usr/bin/env python3
"""One rank's share of a 2^log2n-term MSM cut G ways, on ONE GPU: by terms (all windows of n / G terms) and by
windows (windows g, g + G, ... of all n terms), device work only (jj_msm_partial, record left on the device) and with the host tail
of one record; plus what every rank runs after the all_gather: the G records folded on the device + one 8 KB copy + the host tail of
one record (jj_msm_combine_dev, round 5) against round 4's copy of all G records + the host's additions (option msm_fold_dev = 0); and the
HYBRID partitions (terms / a  x  windows / b with a b = G: rank (i, j) reduces windows j, j + b, ... of term slice i).
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
# hybrid partitions: a term slices x b window classes
for a in (2, 4):
    b = G // a
    if a * b != G or b < 2:
        continue
    ma = n // a
    for (i, j) in ((0, 0), (a - 1, b - 1)):
        print("  hybrid %d term slices x %d window classes, rank (%d, %d): windows %d, %d.. of %d terms : %.3f / %.3f" % ((a, b, i, j, j, j + b, ma) + timed(lambda: eng.msm_partial(S[i * ma:(i + 1) * ma], P[i * ma:(i + 1) * ma], j, b))))
    recs_h = torch.stack([eng.msm_partial(S[i * ma:(i + 1) * ma], P[i * ma:(i + 1) * ma], j, b) for i in range(a) for j in range(b)])
    assert (eng.msm_combine(recs_h) == eng.msm(S, P).cpu().numpy()).all()
    print("    its %d gathered records folded on the device + one host tail : %.3f / %.3f" % ((G,) + timed(lambda: eng.msm_combine(recs_h))))
recs_t = torch.stack([eng.msm_partial(S[g * m:(g + 1) * m], P[g * m:(g + 1) * m]) for g in range(G)])
recs_w = torch.stack([eng.msm_partial(S, P, g, G) for g in range(G)])
want = eng.msm(S, P).cpu().numpy()
assert (eng.msm_combine(recs_t) == want).all() and (eng.msm_combine(recs_w) == want).all()
print("  %d gathered records folded on the device + 8 KB copy + host tail of ONE record : terms %.3f / %.3f   windows %.3f / %.3f" % ((G,) + timed(lambda: eng.msm_combine(recs_t)) + timed(lambda: eng.msm_combine(recs_w))))
eng_h = Engine(0, options={"msm_fold_dev": 0})
assert (eng_h.msm_combine(recs_t) == want).all()
print("  round 4: copy of all %d records to the host + the host adds them (option msm_fold_dev = 0)     : terms %.3f / %.3f   windows %.3f / %.3f" % ((G,) + timed(lambda: eng_h.msm_combine(recs_t)) + timed(lambda: eng_h.msm_combine(recs_w))))
ht, hw = recs_t.cpu().numpy(), recs_w.cpu().numpy()
print("  host tail alone (records already on the host)          : terms %.3f / %.3f   windows %.3f / %.3f" % (timed(lambda: eng.msm_combine(ht)) + timed(lambda: eng.msm_combine(hw))))
print("  both partitions give the point of the one-GPU MSM: ok")
