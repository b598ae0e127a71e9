#!/usr/bin/env python3
"""jj_msm_dev (device-side finish: Horner + inversion on one quad of lanes, result left in HBM) against jj_msm (host tail) -> profiles/
r4_msm_dev_finish.txt.   python tools/msm_dev_finish.py   (needs an MI355X)
Per size: median wall time of one synchronous call (the caller waits for the point), and the HOST time of a call that does not wait
(jj_msm_dev only queues work: this is what a pipeline that feeds the sum to the next kernel pays on its host thread)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jubjub_amd import Engine  # noqa: E402

SEED = 0x4A55424A5542


def med(fn, reps=15):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


def main():
    eng = Engine(0)
    dev = torch.device("cuda", 0)
    out = torch.zeros(64, dtype=torch.uint8, device=dev)
    print("%10s %22s %22s %26s %10s" % ("terms", "jj_msm (host tail) ms", "jj_msm_dev + sync ms", "jj_msm_dev host-side ms", "equal"))
    for lg in (0, 10, 14, 17, 20):
        n = 1 << lg
        s = eng.synth_scalars(n, SEED, 0, device=dev)
        p = eng.random_points(n, SEED ^ 1, 0, subgroup=False, device=dev)
        a = eng.msm(s, p)
        b = eng.msm_dev(s, p, out=out)
        torch.cuda.synchronize(dev)
        same = bool(torch.equal(a.to(dev), b))

        def host_tail():
            eng.msm(s, p)

        def dev_sync():
            eng.msm_dev(s, p, out=out); torch.cuda.synchronize(dev)

        def dev_async():
            eng.msm_dev(s, p, out=out)

        t_host, t_dev = med(host_tail), med(dev_sync)
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(15):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter(); dev_async(); ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize(dev)
        print("%10d %22.3f %22.3f %26.3f %10s" % (n, t_host, t_dev, sorted(ts)[7] * 1e3, same), flush=True)


if __name__ == "__main__":
    main()
