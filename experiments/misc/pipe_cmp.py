import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from jubjub_amd import Engine
from oracle import jubjub_ref as J
eng = Engine(0)
n = 1 << 22
rng = np.random.default_rng(1)
S = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); S[:, 31] &= 0x0F
base = np.frombuffer(J.GENERATOR[0].to_bytes(32, "little") + J.GENERATOR[1].to_bytes(32, "little"), dtype=np.uint8)
tab = eng.fixedbase_table(base)
P = eng.fixedbase_mul(tab, S[::-1].copy())
for name, fn in (("varbase", lambda: eng.varbase_mul(S, P)), ("fixedbase", lambda: eng.fixedbase_mul(tab, S))):
    fn()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    dt = (time.perf_counter() - t0) / 3
    print("%s 2^22 host buffers, chunk_log2=%s: %.1f ms  %.1f M/s" % (name, os.environ.get("JJ_PIPE_CHUNK_LOG2", "18"), dt * 1e3, n / dt / 1e6))
