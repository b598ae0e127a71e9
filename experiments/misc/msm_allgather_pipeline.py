#!/usr/bin/env python3
"""What one rank of a G-rank MSM runs per MSM when MANY MSMs are summed (a prover's commitments), on ONE GPU: the rank's share of a
2^log2n-term MSM (term partition: n / G terms, all windows) through jj_msm_allgather -- synchronous calls -- and through
jj_msm_allgather_begin / jj_msm_finish with 2, 3, 4 jobs in flight, where the all-gather is played by tools/loopback_comm.cpp (the other
ranks' records, prepared beforehand, copied into the receive buffer on the job's stream: one launch, like ncclAllGather's kernel, but
without the xGMI latency of a real gather -- add ~0.02-0.04 ms to a SYNCHRONOUS call for that; with jobs in flight the gather's latency
is hidden like the rest of the exchange).  Against the whole MSM on one GPU, synchronous and with four jobs in flight: the ratio is the
speed-up G ranks can have on a stream of MSMs.
  python experiments/misc/msm_allgather_pipeline.py [log2n] [G]"""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from jubjub_amd import Engine  # noqa: E402
from util import LoopbackComm  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 1 << log2n
m = n // G
eng = Engine(0)
dev = torch.device("cuda", 0)
S = eng.synth_scalars(n, 7, 0, device=dev)
P = eng.random_points(n, 7, 0, subgroup=False, device=dev)
want = eng.msm(S, P).cpu().numpy()
recs = torch.stack([eng.msm_partial(S[g * m:(g + 1) * m], P[g * m:(g + 1) * m]) for g in range(G)])
COUNT = 64


def series(begin, finish, depth, count=COUNT):
    """ms per MSM over `count` MSMs with `depth` jobs in flight (depth 1: synchronous calls); median of 7 series"""
    ts = []
    out = None
    for rep in range(9):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend = []
        for _ in range(count):
            pend.append(begin())
            if len(pend) == depth:
                out = finish(pend.pop(0))
        for j in pend:
            out = finish(j)
        torch.cuda.synchronize()
        if rep >= 2:
            ts.append((time.perf_counter() - t0) * 1e3 / count)
    ts.sort()
    return ts[len(ts) // 2], ts[0], out


print("2^%d terms, %d ranks, term partition; ms per MSM, median / fastest of 7 series of %d MSMs, device-resident inputs" % (log2n, G, COUNT))
one = {}
for depth in (1, 2, 4):
    if depth == 1:
        med, lo, out = series(lambda: eng.msm(S, P), lambda o: o, 1)
        out = out.cpu().numpy() if hasattr(out, "cpu") else out
    else:
        med, lo, out = series(lambda: eng.msm_begin(S, P), eng.msm_finish, depth)
    assert (out == want).all()
    one[depth] = med
    print("  one GPU, whole MSM, %d in flight                         : %.3f / %.3f" % (depth, med, lo))
for rank in (0, G - 1):
    comm = LoopbackComm(rank, G)
    comm.add_round(recs)
    s, p = S[rank * m:(rank + 1) * m], P[rank * m:(rank + 1) * m]
    # round 6: a context with several ranks starts with ONE lane (every gather of the communicator on one stream); a caller that has checked
    # multi-stream gathers on its node sets msm_lanes back after jj_ctx_set_comm -- both are measured
    for lanes in (1, 3):
        eng.set_comm(comm)
        eng.set_option("msm_lanes", lanes)
        for depth in ((1, 2, 4) if lanes == 1 else (2, 3, 4, 6)):
            if depth == 1:
                med, lo, out = series(lambda: eng.msm_allgather(s, p), lambda o: o, 1)
            else:
                med, lo, out = series(lambda: eng.msm_allgather_begin(s, p), eng.msm_finish, depth)
            assert (out == want).all()
            print("  rank %d of %d, msm_lanes %d: 2^%d terms + gather of %d records + fold + host tail, %d in flight : %.3f / %.3f   (x%.2f of one GPU synchronous, x%.2f of one GPU with 4 in flight)" % (
                rank, G, lanes, log2n - (G.bit_length() - 1), G, depth, med, lo, one[1] / med, one[4] / med))
        eng.set_comm(None)
    comm.close()
eng.close()
