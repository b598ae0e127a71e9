#!/usr/bin/env python3
"""Sums over several fixed bases with SHORT scalars: one pass over a composite LDS table (jj_fixedbase_composite_mul)
against one pass per base (jj_fixedbase_multi_mul), e.g. 3 bases x 64-bit scalars (value-commitment-like v*G_v terms).
  python tools/composite_bench.py [log2n]      (needs an MI355X)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from jubjub_amd import Engine  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << log2n
eng = Engine(0)
dev = torch.device("cuda", 0)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("%-34s %10s %16s %22s" % ("configuration (2^%d units)" % log2n, "ms / pass", "M units / s", "M short scalar-muls / s"))
for bits in ([64, 64, 64], [124, 124], [40] * 6, [10] * 21):
    nb = len(bits)
    bases = eng.random_points(nb, 99, 0, subgroup=True, device=dev)
    S = eng.synth_bytes32(nb * n, 5, 0, device=dev).reshape(nb, n, 32)
    mask = torch.zeros(nb, 32, dtype=torch.uint8, device=dev)
    for b in range(nb):
        mask[b] = torch.from_numpy(np.frombuffer(((1 << bits[b]) - 1).to_bytes(32, "little"), dtype=np.uint8).copy()).to(dev)
    Sm = S & mask[:, None, :]                                   # the same short scalars for both paths
    ct = eng.fixedbase_composite_table(bases, bits)
    tabs = [eng.fixedbase_table(bases[b:b + 1].reshape(64), 0) for b in range(nb)]
    a = eng.fixedbase_composite_mul(ct, S)
    b_ = eng.fixedbase_multi_mul(tabs, Sm)
    assert bool(torch.equal(a, b_)), bits
    t1 = timed(lambda: eng.fixedbase_composite_mul(ct, S))
    t2 = timed(lambda: eng.fixedbase_multi_mul(tabs, Sm))
    name = "%d x %d bits" % (nb, bits[0])
    print("%-34s %10.3f %16.1f %22.1f" % (name + ": composite table, one pass", t1, n / t1 / 1e3, nb * n / t1 / 1e3))
    print("%-34s %10.3f %16.1f %22.1f" % (name + ": one comb pass per base", t2, n / t2 / 1e3, nb * n / t2 / 1e3))
    ct.close()
    for t in tabs:
        t.close()
print("both paths agree on every unit")
