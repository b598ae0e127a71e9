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

    def msm_partial(self, s, p, part_index=0, part_count=1):
        """the record of window sums jj_msm_partial leaves, built by the oracle (tests/util.py)"""
        from util import oracle_msm_record

        return oracle_msm_record(np.asarray(s), np.asarray(p), part_index, part_count)

    def msm_combine(self, records):
        """the product's host-only jj_msm_combine (no GPU involved): the real second half of the distributed MSM"""
        import ctypes

        from jubjub_amd import _lib

        recs = np.ascontiguousarray(np.asarray(records, dtype=np.uint8).reshape(-1, _lib.MSM_PARTIAL_BYTES))
        out = np.empty(64, np.uint8)
        rc = _lib.load().jj_msm_combine(ctypes.c_size_t(len(recs)), recs.ctypes.data if len(recs) else None, out.ctypes.data)
        assert rc == 0, rc
        return out


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
        bywin = jd.msm_distributed(eng, S, P, partition="window")     # every rank: all terms, windows rank, rank + world, ...
        q.put((rank, lo, hi, out.tobytes(), bytes(total), bytes(pre), bytes(empty), bytes(bywin)))
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
        assert r[7] == want_msm                         # window partition: same point


def test_msm_combine_scalar_and_ifma_chains_agree():
    """The host tail has two implementations of its Horner chain and of the window-by-window sums (jj_host_tail.h: scalar 4 x 64-bit;
    jj_host_tail_ifma.h: AVX-512 IFMA, taken when the CPU has it): the test above in a process that forces the scalar one, and random records
    (1..9 records, layouts 16 / 17 / 23 / 64, equal points in several records = doublings through the addition, P and -P = the identity in the
    middle of a chain, the identity and 8-torsion points as window sums) through both."""
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from jubjub_amd import _lib\nassert _lib.load().jj_ctx_set_option(None, b'host_tail_scalar', int(sys.argv[1] == 'scalar')) == 0\n"
            "import test_dist_cpu as T\nT.test_msm_combine_host_only()\nT._combine_random_records()\nprint('COMBINE OK')\n") % (
                os.path.dirname(os.path.abspath(__file__)), os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    for mode in ("scalar", "auto"):
        r = subprocess.run([sys.executable, "-c", code, mode], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "COMBINE OK" in r.stdout, (mode, r.stdout[-2000:], r.stderr[-2000:])


def _combine_random_records():
    import ctypes

    from jubjub_amd import _lib
    from oracle import c_oracle as O
    from oracle import jubjub_ref as J
    from util import arr64, oracle_msm_record, rand_points, rand_scalars

    lib = _lib.load()
    rng = np.random.default_rng(77)
    g8 = J.scalar_mul_fast(J.GENERATOR, J.R_MOD)
    for trial in range(12):
        G = int(rng.integers(1, 10))
        W = int(rng.choice([16, 17, 23, 64]))
        n = int(rng.integers(G, 4 * G + 2))
        S, P = rand_scalars(500 + trial, n, full_width=True), rand_points(600 + trial, n)
        if n >= 4:
            P[1] = P[0]; S[1] = S[0]                                  # the same term twice: equal window sums meet (a doubling through the addition)
            P[3] = arr64([J.affine_neg(tuple(int.from_bytes(bytes(P[2][k:k + 32]), "little") for k in (0, 32)))])[0]; S[3] = S[2]     # P and -P
        if n >= 6:
            P[4] = arr64([g8])[0]; P[5] = arr64([J.AFFINE_IDENTITY])[0]
        order = rng.permutation(n)
        cut = [n * g // G for g in range(G + 1)]
        recs = np.ascontiguousarray(np.stack([oracle_msm_record(S[order[cut[g]:cut[g + 1]]], P[order[cut[g]:cut[g + 1]]], W=W) for g in range(G)]))
        out = np.empty(64, np.uint8)
        assert lib.jj_msm_combine(ctypes.c_size_t(G), recs.ctypes.data, out.ctypes.data) == 0
        assert (out == O.msm(S, P).reshape(64)).all(), (trial, G, W, n)


def test_msm_combine_host_only():
    """jj_msm_combine against oracle-built records: one record, a term partition into records of the same layout, a window
    partition (disjoint window masks), mixed window layouts in one call, edge scalars on special points, damaged records."""
    import ctypes

    from jubjub_amd import _lib
    from oracle import c_oracle as O
    from oracle import jubjub_ref as J
    from util import EDGE_SCALARS, arr32, arr64, oracle_msm_record, rand_points, rand_scalars

    lib = _lib.load()

    def combine(recs):
        recs = np.ascontiguousarray(np.stack(recs)) if len(recs) else np.zeros((0, _lib.MSM_PARTIAL_BYTES), np.uint8)
        out = np.empty(64, np.uint8)
        rc = lib.jj_msm_combine(ctypes.c_size_t(len(recs)), recs.ctypes.data if len(recs) else None, out.ctypes.data)
        return rc, out

    n = 21
    S, P = rand_scalars(300, n, full_width=True), rand_points(301, n)
    want = O.msm(S, P)
    for W in (64, 23, 16):
        rc, out = combine([oracle_msm_record(S, P, W=W)])
        assert rc == 0 and (out == want).all(), W
        rc, out = combine([oracle_msm_record(S, P, g, 3, W=W) for g in range(3)])           # window partition
        assert rc == 0 and (out == want).all(), W
        rc, out = combine([oracle_msm_record(S[:8], P[:8], W=W), oracle_msm_record(S[8:], P[8:], W=W)])   # term partition
        assert rc == 0 and (out == want).all(), W
    rc, out = combine([oracle_msm_record(S[:8], P[:8], W=64), oracle_msm_record(S[8:15], P[8:15], W=23), oracle_msm_record(S[15:], P[15:], W=16)])
    assert rc == 0 and (out == want).all()                                                # three layouts in one call
    g8 = J.scalar_mul_fast(J.GENERATOR, J.R_MOD)
    Se = arr32(EDGE_SCALARS)
    Pe = arr64(([J.AFFINE_IDENTITY, g8, J.scalar_mul_fast(g8, 4), J.GENERATOR, J.affine_neg(J.GENERATOR)] * 5)[: len(EDGE_SCALARS)])
    rc, out = combine([oracle_msm_record(Se, Pe, W=23)])
    assert rc == 0 and (out == O.msm(Se, Pe)).all()
    rc, out = combine([])
    assert rc == 0 and bytes(out[:32]) == bytes(32) and out[32] == 1 and not out[33:].any()           # the identity
    bad = oracle_msm_record(S, P)
    bad[8] = 65                                                                            # W out of range
    assert combine([bad])[0] != 0
    bad = oracle_msm_record(S, P)
    bad[64 + 31] = 0xFF                                                                    # a coordinate >= q
    assert combine([bad])[0] != 0


def test_rccl_comm_setup_fails_on_every_rank_before_any_collective():
    """RcclComm: a rank that cannot load RCCL (or get the id) must fail the set-up on ALL ranks before the collective ncclCommInitRank
    could leave the healthy ones waiting -- the `agree` step carries every rank's error text, the id broadcast is entered only when
    nobody failed (bench.py then falls back to the torch.distributed exchange)"""
    from jubjub_amd.dist import RcclComm

    calls = []
    with pytest.raises(RuntimeError, match="rank 0: OSError: boom"):          # a healthy rank 1 learns of rank 0's failure
        RcclComm(1, 2, agree=lambda e: (calls.append(e), ["OSError: boom", e])[1], broadcast=lambda raw: calls.append("broadcast"))
    assert calls == [None]
    calls.clear()
    with pytest.raises(RuntimeError, match="rank 0: OSError"):                # the broken rank itself still takes part in the agreement
        RcclComm(0, 2, lib_path="/nonexistent/librccl.so", agree=lambda e: (calls.append(e), [e, None])[1], broadcast=lambda raw: calls.append("broadcast"))
    assert len(calls) == 1 and calls[0].startswith("OSError")
    with pytest.raises(RuntimeError, match="RCCL set-up failed"):
        RcclComm(0, 1, lib_path="/nonexistent/librccl.so")
