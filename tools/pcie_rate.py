#!/usr/bin/env python3
"""PCIe-inclusive throughput of the C ABI when it is handed HOST buffers (numpy): the library stages H2D, runs the
kernels and copies the result back.  Reported in DESIGN.md; it is never bench.py's `value`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jubjub_amd import Engine  # noqa: E402

eng = Engine(0)
n = 1 << 20
rng = np.random.default_rng(1)
S = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
S[:, 31] &= 0x0F
GEN_U = 0x62EDCBB8BF3787C88B0F03DDD60A8187CAF55D1B29BF81AFE4B3D35DF1A7ADFE        # generator, reference src/lib.rs:1380-1396
base = np.frombuffer(GEN_U.to_bytes(32, "little") + (11).to_bytes(32, "little"), dtype=np.uint8)
tab = eng.fixedbase_table(base)
P = eng.fixedbase_mul(tab, S[::-1].copy())
for name, fn, units in (("varbase 2^20 (96 MB in, 64 MB out)", lambda: eng.varbase_mul(S, P), n),
                        ("fixedbase 2^20 (32 MB in, 64 MB out)", lambda: eng.fixedbase_mul(tab, S), n),
                        ("decompress 2^20 (32 MB in, 65 MB out)", lambda: eng.decompress(eng.compress(P) if False else ENC, 1), n)):
    if name.startswith("decompress"):
        ENC = eng.compress(P)
    fn()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    dt = (time.perf_counter() - t0) / 5
    print("%-40s %8.2f ms  %8.1f M units/s (host pointers, pageable memory)" % (name, dt * 1e3, units / dt / 1e6))
