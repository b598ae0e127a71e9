#!/usr/bin/env python3
"""Randomised soak of the MSM jobs in flight (jj_msm_begin / jj_msm_allgather_begin + jj_msm_finish): random sizes from the small-batch path to
2^21 terms, 1-6 jobs in flight over 1-4 lanes, finished in random order, interleaved with synchronous calls on the context's stream -- every
point against the synchronous jj_msm of the same terms (another lane, another stream) and, for the smaller ones, the oracle; the distributed halves with an
all-gather that plays 2-8 ranks (tests/util.py LoopbackComm).
Usage: python tests/soak_jobs.py [seconds] [seed]   (needs an MI355X)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from jubjub_amd import Engine  # noqa: E402
from jubjub_amd.dist import shard_bounds  # noqa: E402
from oracle import c_oracle as O  # noqa: E402
from util import LoopbackComm  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
SEED0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
dev = torch.device("cuda", 0)
ref = Engine(0)
NMAX = 1 << 21
S_all = ref.synth_scalars(NMAX, 11, 0, device=dev)
P_all = ref.random_points(NMAX, 11, 0, subgroup=False, device=dev)
t_end, rnd, jobs_done = time.time() + budget, 0, 0
while time.time() < t_end:
    rng = np.random.default_rng(SEED0 + rnd)
    lanes = int(rng.integers(1, 5))
    eng = Engine(0, options={"msm_lanes": lanes})
    depth = int(rng.integers(1, 7))
    G = int(rng.choice([0, 0, 2, 3, 8]))                        # 0: one-GPU jobs; else the distributed halves with G ranks played on this GPU
    njobs = int(rng.integers(3, 10))
    specs = []
    for _ in range(njobs):
        n = int(rng.choice([int(rng.integers(1, 3000)), int(rng.integers(3000, 200000)), int(rng.integers(200000, NMAX // (4 if G else 1)))]))
        lo = int(rng.integers(0, NMAX - n + 1))
        specs.append((lo, n))
    want = [ref.msm(S_all[lo:lo + n], P_all[lo:lo + n]).cpu().numpy() for lo, n in specs]
    comm = None
    if G:
        comm = LoopbackComm(int(rng.integers(0, G)), G)
        for lo, n in specs:
            comm.add_round(torch.stack([ref.msm_partial(S_all[lo + a:lo + b], P_all[lo + a:lo + b]) for a, b in (shard_bounds(n, g, G) for g in range(G))]))
        eng.set_comm(comm)
    pend, got = [], {}
    for i, (lo, n) in enumerate(specs):
        if G:
            a, b = shard_bounds(n, comm.rank, G)
            pend.append((i, eng.msm_allgather_begin(S_all[lo + a:lo + b], P_all[lo + a:lo + b])))
        else:
            pend.append((i, eng.msm_begin(S_all[lo:lo + n], P_all[lo:lo + n])))
        if rng.integers(0, 3) == 0:                            # a synchronous call on the context's stream between the jobs
            k = int(rng.integers(0, len(specs)))
            assert (eng.msm(S_all[specs[k][0]:specs[k][0] + specs[k][1]], P_all[specs[k][0]:specs[k][0] + specs[k][1]]).cpu().numpy() == want[k]).all(), ("sync between jobs", rnd, k)
        while len(pend) >= depth:
            j = pend.pop(int(rng.integers(0, len(pend))))     # any order
            got[j[0]] = eng.msm_finish(j[1])
    while pend:
        j = pend.pop(int(rng.integers(0, len(pend))))
        got[j[0]] = eng.msm_finish(j[1])
    for i, (lo, n) in enumerate(specs):
        assert (got[i] == want[i]).all(), ("job", rnd, i, lo, n, lanes, depth, G)
        if n <= 40000 and rnd % 3 == 0:
            assert (want[i] == O.msm(S_all[lo:lo + n].cpu().numpy(), P_all[lo:lo + n].cpu().numpy()).reshape(64)).all(), ("oracle", rnd, i)
    if comm is not None:
        eng.set_comm(None)
        comm.close()
    eng.close()
    jobs_done += njobs
    rnd += 1
    print("round %d ok: %d jobs, lanes %d, depth %d, G %d (%d jobs so far, %.0f s left)" % (rnd, njobs, lanes, depth, G, jobs_done, t_end - time.time()), flush=True)
print("JOB SOAK PASSED: %d rounds, %d jobs, every point equal to the synchronous call's (and the oracle's for the smaller ones)" % (rnd, jobs_done))
