"""
world_size-2 gloo tests (CPU) of the multi-GPU host logic in jubjub_amd/dist.py.  The GPU engine is replaced by
an oracle-backed stand-in — allowed in tests only; the product path has no CPU fallback.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleEngine:
    """Engine stand-in backed by the C oracle (tests only)."""

    def __init__(self):
        from oracle import c_oracle as O

        self.O = O

    def varbase_mul(self, s, p):
        return self.O.varbase_mul(np.asarray(s), np.asarray(p))

    def fixedbase_mul(self, table, s):
        return self.O.fixedbase_mul(np.asarray(s), table)

    def decompress(self, e, flags=1):
        return self.O.decompress(np.asarray(e), flags)

    def msm(self, s, p):
        return self.O.msm(np.asarray(s), np.asarray(p))

    def point_sum(self, p):
        return self.O.point_sum(np.asarray(p))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from jubjub_amd import dist as jd
        from util import rand_points, rand_scalars

        eng = OracleEngine()
        S, P = rand_scalars(100, n), rand_points(101, n)
        lo, hi, out = jd.varbase_mul_sharded(eng, S, P)
        total = jd.msm_distributed(eng, S, P)
        pre = jd.msm_distributed(eng, S[lo:hi], P[lo:hi], presharded=True)
        empty = jd.msm_distributed(eng, S[:1], P[:1])          # rank 1 gets an empty shard
        q.put((rank, lo, hi, out.tobytes(), bytes(total), bytes(pre), bytes(empty)))
    finally:
        dist.destroy_process_group()


def test_shard_bounds():
    from jubjub_amd.dist import shard_bounds

    for n in (0, 1, 7, 8, 1 << 20, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_world2_gloo_sharding_and_msm():
    from oracle import c_oracle as O
    from util import rand_points, rand_scalars

    n, world = 37, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    S, P = rand_scalars(100, n), rand_points(101, n)
    want = O.varbase_mul(S, P)
    want_msm = bytes(O.msm(S, P))
    cat = b"".join(r[3] for r in res)
    assert cat == want.tobytes()                       # shards concatenate to the single-process result
    assert res[0][1:3] == (0, 19) and res[1][1:3] == (19, 37)
    for r in res:
        assert r[4] == want_msm and r[5] == want_msm    # every rank holds the full MSM
        assert r[6] == bytes(O.msm(S[:1], P[:1]))
