#!/usr/bin/env python3
"""Randomised soak of the host-pointer path (include/jubjub_hip.h "host buffers"): random batch sizes around the pipeline's chunk
boundaries, every array independently page-locked (jj_host_alloc), pageable, or -- results -- a buffer of the library's pool (jj_result_acquire), random chunk lengths, bounce / in-place page-locking,
uniform / ramped chunk schedules, the MSM of the same host arrays in one or several passes -- against the device-resident entry points (bit-exact, all units) and an oracle sample.
Usage: python tests/soak_host.py [seconds] [seed]   (needs an MI355X)
SOAK_FORCE_REGISTER=1: every round in JJ_PIPE_PAGEABLE=register mode; SOAK_TRACE=1: one line before every entry point (where a crash happened)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from jubjub_amd import Engine  # noqa: E402
from oracle import c_oracle as O  # noqa: E402
from oracle import jubjub_ref as J  # noqa: E402
from util import pt64  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
SEED0 = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
base = pt64(J.GENERATOR)
small = O.fixedbase_mul(np.random.default_rng(1).integers(0, 256, size=(4096, 32), dtype=np.uint8) & np.array([255] * 31 + [15], np.uint8), base)
ref = Engine(0)
t_end, rnd, units = time.time() + budget, 0, 0
while time.time() < t_end:
    rng = np.random.default_rng(SEED0 + rnd)
    env = {}
    if rng.integers(0, 2):
        env["pipe_chunk_log2"] = int(rng.integers(14, 20))
    if rng.integers(0, 3) == 0 or os.environ.get("SOAK_FORCE_REGISTER"):
        env["pipe_pageable_register"] = 1
    if rng.integers(0, 3) == 0:
        env["pipe_ramp"] = 0
    if rng.integers(0, 2):
        env["pipe_copy_threads"] = int(rng.integers(1, 9))
    if rng.integers(0, 4) == 0:
        env["msm_host_split"] = 0
    eng = Engine(0, options=env)                     # round 6: context options (jj_ctx_set_option); the library reads no environment variable
    n = int(rng.choice([(1 << 18) + int(rng.integers(-3, 4)), int(rng.integers(1 << 16, 1 << 21)), (1 << 20) + int(rng.integers(-70000, 70000))]))

    pooled = []                                      # result buffers from the library's pool (jj_result_acquire), released at the end of the round

    def host(a, result=False):
        k = int(rng.integers(0, 3 if result else 2))
        if k == 1:
            h = eng.host_alloc(a.shape); h[...] = a
            return h
        if k == 2:
            h = eng.result_acquire(a.shape)          # contents are whatever the previous user left: a result buffer is written, never read
            pooled.append(h)
            return h
        return np.array(a, copy=True)

    S = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    P = np.ascontiguousarray(small[rng.integers(0, 4096, size=n)])
    idx = np.concatenate([rng.integers(0, n, size=300), [0, n - 1]])
    dS, dP = torch.from_numpy(S).cuda(), torch.from_numpy(P).cuda()
    hS, hP = host(S), host(P)
    out = host(np.zeros((n, 64), np.uint8), result=True) if rng.integers(0, 2) else None            # None: a fresh pageable result array
    trace = (lambda what: print("  [%d] %s" % (rnd, what), flush=True)) if os.environ.get("SOAK_TRACE") else (lambda what: None)
    trace("varbase n=%d %s" % (n, env))
    got = eng.varbase_mul(hS, hP, out=out)
    assert (got == ref.varbase_mul(dS, dP).cpu().numpy()).all(), ("varbase", rnd, env, n)
    assert (got[idx] == O.varbase_mul(S[idx], P[idx])).all(), ("varbase oracle", rnd)
    tab, rtab = eng.fixedbase_table(base), ref.fixedbase_table(base)
    out32 = host(np.zeros((n, 32), np.uint8), result=True) if rng.integers(0, 2) else None
    trace("fixedbase")
    got = eng.fixedbase_mul_compressed(tab, hS, out=out32)
    assert (got == ref.fixedbase_mul_compressed(rtab, dS).cpu().numpy()).all(), ("fixedbase compressed", rnd, env, n)
    assert (got[idx] == O.compress(O.fixedbase_mul(S[idx], base))).all(), ("fixedbase oracle", rnd)
    enc = O.compress(P)
    bad = rng.integers(0, n, size=n // 20)
    enc[bad] = rng.integers(0, 256, size=(len(bad), 32), dtype=np.uint8)
    flags = int(rng.choice([1, 5, 13, 15]))
    trace("decompress")
    o1, k1 = eng.decompress(host(enc), flags, out=(host(np.zeros((n, 64), np.uint8), result=True), host(np.zeros((n,), np.uint8), result=True)) if rng.integers(0, 2) else None)
    o2, k2 = ref.decompress(torch.from_numpy(enc).cuda(), flags)
    assert (k1 == k2.cpu().numpy()).all() and (o1 == o2.cpu().numpy()).all(), ("decompress", rnd, env, n, flags)
    eo, ek = O.decompress(enc[idx], flags)
    assert (k1[idx] == ek).all() and (o1[idx] == eo).all(), ("decompress oracle", rnd)
    trace("msm")
    want = ref.msm(dS, dP).cpu().numpy()
    assert (eng.msm(hS, hP) == want).all(), ("msm from host arrays", rnd, env, n)         # 2^19 terms and more: 2..8 passes, copies beside the kernels
    if rnd % 4 == 0:
        assert (want == O.msm_pippenger(S, P).reshape(64)).all(), ("msm oracle", rnd, n)
    trace("close")
    if pooled and rng.integers(0, 2):
        for h in pooled:
            eng.result_release(h)                    # (otherwise the context frees them with itself)
        assert eng.result_pool_stats()["in_use"] == 0
    tab.close(); rtab.close(); eng.close()
    units += n; rnd += 1
    print("round %d ok: n=%d %s (%d units so far, %.0f s left)" % (rnd, n, env, units, t_end - time.time()), flush=True)
print("HOST SOAK PASSED: %d rounds, %d units per entry point, all bit-exact vs the device-resident path and the oracle sample" % (rnd, units))
