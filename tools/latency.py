#!/usr/bin/env python3
"""Small-batch latency of the C ABI (device-resident inputs): median wall time per call for n = 1 .. 2^16.
Usage: python tools/latency.py   (needs an MI355X)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from jubjub_amd import Engine  # noqa: E402

GEN_U = 0x62EDCBB8BF3787C88B0F03DDD60A8187CAF55D1B29BF81AFE4B3D35DF1A7ADFE
base = np.frombuffer(GEN_U.to_bytes(32, "little") + (11).to_bytes(32, "little"), dtype=np.uint8)


def main():
    eng = Engine(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    tab = eng.fixedbase_table(torch.from_numpy(base.copy()).to(dev))
    print("%8s %14s %14s %14s %14s" % ("n", "varbase ms", "fixedbase ms", "msm ms", "decompress ms"))
    for lg in (0, 4, 8, 10, 12, 14, 16):
        n = 1 << lg
        s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
        s[:, 31] &= 0x0F
        pts = eng.fixedbase_mul(tab, s)
        enc = eng.compress(pts)
        row = []
        for fn in (lambda: eng.varbase_mul(s, pts), lambda: eng.fixedbase_mul(tab, s), lambda: eng.msm(s, pts), lambda: eng.decompress(enc, 1)):
            ts = []
            for _ in range(12):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize(dev)
                ts.append(time.perf_counter() - t0)
            row.append(sorted(ts)[len(ts) // 2] * 1e3)
        print("%8d %14.3f %14.3f %14.3f %14.3f" % (n, *row))


if __name__ == "__main__":
    main()
