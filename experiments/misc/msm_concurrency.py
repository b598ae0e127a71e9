#!/usr/bin/env python3
"""K contexts (each with its own stream and workspaces) running MSMs concurrently on one GPU, one host thread each: aggregate time
per MSM.  The latency-bound tails of one MSM (fix-up, bucket reduce, host tail) leave most of the GPU idle; another context's
sort / accumulation can run there.      python experiments/misc/msm_concurrency.py [log2n] [iters]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from jubjub_amd import Engine  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
n = 1 << log2n
KMAX = 4
engs = [Engine(0) for _ in range(KMAX)]
data = []
for i, e in enumerate(engs):
    data.append((e.synth_scalars(n, 7 + i, 0, device=dev), e.random_points(n, 9 + i, 0, device=dev)))
torch.cuda.synchronize()
want = [engs[0].msm(s, p).cpu() for s, p in data]


def loop(i, mode, out):
    e, (s, p) = engs[i], data[i]
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        r = None
        if mode == "sync":
            for _ in range(iters):
                r = e.msm(s, p)
        else:                                            # two jobs in flight per context
            pend = []
            for _ in range(iters):
                pend.append(e.msm_begin(s, p))
                if len(pend) == 2:
                    r = e.msm_finish(pend.pop(0))
            for j in pend:
                r = e.msm_finish(j)
    st.synchronize()
    out[i] = r


print("2^%d-term MSMs on one MI355X, K contexts in K host threads (aggregate ms per MSM; every result checked)" % log2n)
for mode in ("sync", "async2"):
    for k in range(1, KMAX + 1):
        best = None
        for rep in range(3):
            out = [None] * k
            th = [threading.Thread(target=loop, args=(i, mode, out)) for i in range(k)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            [t.start() for t in th]
            [t.join() for t in th]
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (iters * k) * 1e3
            best = dt if best is None else min(best, dt)
            for i in range(k):
                r = out[i].cpu() if hasattr(out[i], "cpu") else torch.from_numpy(out[i])
                assert bool((r.reshape(64) == want[i].reshape(64)).all()), (mode, k, i)
        print("  %-6s K = %d: %.4f ms per MSM  (%.1f M terms/s)" % (mode, k, best, n / best / 1e3))
