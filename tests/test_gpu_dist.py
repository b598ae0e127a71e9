"""
Multi-rank path of bench.py on real hardware (one GPU is enough): the ranks bench.py spawns itself, the torch.distributed
exchange of the MSM's 64-byte partial points (gloo with two ranks stacked on one GPU; RCCL with one rank), strong and weak
scaling shards built from the global unit indices, and the folded point against the oracle.
"""
import functools
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import c_oracle as O
from oracle import jubjub_ref as J

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_bench(args, env_extra, timeout=400):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@functools.lru_cache(maxsize=None)      # five tests ask for the same 2^16-term point: ~28 s of Python big-integer input synthesis each time
def oracle_msm(total):
    import bench

    b32 = lambda k: np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint8)
    s = np.stack([b32(J.synth_scalar(i, bench.SEED)) for i in range(total)])
    p = np.stack([np.concatenate([b32(c) for c in J.synth_point(i, bench.POINT_SEED)[0]]) for i in range(total)])
    return bytes(O.msm(s, p).tolist()).hex()


def test_self_spawned_two_ranks_msm_strong_scaling_over_gloo():
    """`python bench.py --gpus 2` (no launcher): two ranks stacked on GPU 0, BASELINE config 4 shape (terms cut across the
    ranks, one all_gather of the partial points), the folded point equals the oracle's MSM over ALL terms"""
    res = run_bench(["--gpus", "2", "--workload", "msm", "--scaling", "strong", "--log2n", "12", "--steps", "2", "--warmup", "1",
                     "--passes", "2", "--backend", "gloo", "--no-cpu-baseline"], {"JJ_BENCH_FORCE_DEVICE": "0"})
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["verified"] is True
    assert res["config"]["units_per_step"] == 2 * (1 << 12)
    assert res["msm_result"] == oracle_msm(1 << 12)
    assert res["rccl_world_size"] == 2 and res["config"]["msm_partition"] == "terms"


@pytest.mark.parametrize("log2n", [12, 16])
def test_self_spawned_two_ranks_msm_window_partition(log2n):
    """--msm-partition window: both ranks hold ALL terms, rank g reduces windows g, g + 2, ...; the records of window sums are
    all-gathered and combined in one host tail; the point equals the oracle's MSM over all terms (small-batch path and Pippenger)"""
    res = run_bench(["--gpus", "2", "--workload", "msm", "--scaling", "strong", "--msm-partition", "window", "--log2n", str(log2n), "--steps", "2",
                     "--warmup", "1", "--passes", "2", "--backend", "gloo", "--no-cpu-baseline"], {"JJ_BENCH_FORCE_DEVICE": "0"})
    assert res["n_gpus"] == 2 and res["verified"] is True and res["all_units_equal_second_pass"] is True
    assert res["verified_units"] == 1 << log2n                         # the timed output IS the whole MSM here: checked over all terms
    assert res["config"]["msm_partition"] == "window" and res["rccl_world_size"] == 2
    assert res["msm_result"] == oracle_msm(1 << log2n)


def test_bench_exits_3_when_verification_fails():
    """fault injection in the CHECKER (one bit of the oracle's expected values flipped, the product untouched): bench.py must print
    "verified": false and exit with status 3, for the default workload and the MSM"""
    env = dict(os.environ, JJ_BENCH_FAULT_INJECT="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    for extra in (["--log2n", "12", "--no-extras"], ["--workload", "msm", "--log2n", "12", "--passes", "2"]):
        r = subprocess.run([sys.executable, BENCH, "--steps", "1", "--warmup", "0", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
        res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert res["verified"] is False and res["all_units_equal_second_pass"] is True


def test_msm_several_contexts_in_bench():
    """--msm-contexts 3 --msm-async 2: three contexts (streams + workspaces) on the one GPU, one host thread each, two jobs in flight
    per context; the step's MSMs are dealt among them; same point, verified"""
    res = run_bench(["--workload", "msm", "--log2n", "15", "--steps", "2", "--warmup", "1", "--passes", "11", "--msm-contexts", "3", "--msm-async", "2",
                     "--no-cpu-baseline"], {})
    assert res["verified"] is True and res["all_units_equal_second_pass"] is True
    assert res["config"]["msm_contexts"] == 3 and res["config"]["msm_jobs_in_flight"] == 2
    assert res["msm_result"] == oracle_msm(1 << 15)


def test_msm_async_pipeline_in_bench():
    """--msm-async 3: three MSMs in flight on one context; same point, verified"""
    res = run_bench(["--workload", "msm", "--log2n", "15", "--steps", "2", "--warmup", "1", "--passes", "7", "--msm-async", "3", "--no-cpu-baseline"], {})
    assert res["verified"] is True and res["config"]["msm_jobs_in_flight"] == 3
    assert res["msm_result"] == oracle_msm(1 << 15)


def test_self_spawned_two_ranks_weak_scaling_independent_shards():
    res = run_bench(["--gpus", "2", "--workload", "varbase", "--log2n", "12", "--steps", "2", "--warmup", "1", "--backend", "gloo",
                     "--no-cpu-baseline", "--no-extras"], {"JJ_BENCH_FORCE_DEVICE": "0"})
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["verified"] is True
    assert res["config"]["units_per_step"] == 2 * (1 << 12) * res["config"]["passes_per_step"]
    res = run_bench(["--gpus", "2", "--workload", "decompress", "--scaling", "strong", "--log2n", "16", "--steps", "2", "--warmup", "1",
                     "--backend", "gloo", "--no-cpu-baseline"], {"JJ_BENCH_FORCE_DEVICE": "0"})
    assert res["n_gpus"] == 2 and res["verified"] is True and res["verified_units"] > 32


def test_one_rank_rccl_all_gather_leg():
    """the RCCL (backend nccl) exchange itself, with the one rank a 1-GPU box can hold"""
    res = run_bench(["--gpus", "1", "--workload", "msm", "--log2n", "12", "--steps", "2", "--warmup", "1", "--passes", "2", "--no-cpu-baseline"],
                    {"JJ_BENCH_FORCE_DIST": "1", "MASTER_PORT": str(free_port())})
    assert res["n_gpus"] == 1 and res["verified"] is True and res["rccl_world_size"] == 1
    assert res["msm_result"] == oracle_msm(1 << 12)
    assert "all_gather" in res["config"]["parallelism"]


@pytest.mark.parametrize("exchange", ["c", "torch"])
def test_one_rank_rccl_msm_both_exchanges(exchange):
    """--msm-exchange c: the whole exchange behind the C ABI (jj_msm_allgather on this process's own RCCL communicator, the default);
    torch: jj_msm_partial + torch.distributed.all_gather + jj_msm_combine.  Pippenger size, one rank over RCCL."""
    res = run_bench(["--gpus", "1", "--workload", "msm", "--log2n", "16", "--steps", "2", "--warmup", "1", "--passes", "2", "--no-cpu-baseline",
                     "--msm-exchange", exchange], {"JJ_BENCH_FORCE_DIST": "1", "MASTER_PORT": str(free_port())})
    assert res["verified"] is True and res["rccl_world_size"] == 1 and res["msm_result"] == oracle_msm(1 << 16)
    assert ("jj_msm_allgather" in res["config"]["parallelism"]) == (exchange == "c")
    assert len(res["rank_ms_per_step"]["per_rank"]) == 1


def test_one_rank_rccl_msm_jobs_in_flight():
    """--msm-async 3 over RCCL: jj_msm_allgather_begin / jj_msm_finish -- the real ncclAllGather queued on the jobs' lanes (two streams
    in turn on one communicator), three MSMs in flight; one rank, Pippenger size.  (G > 1 ranks: the loopback all-gather of
    test_gpu_parity.py::test_msm_allgather_of_G_ranks_played_on_one_gpu.)"""
    res = run_bench(["--gpus", "1", "--workload", "msm", "--log2n", "16", "--steps", "2", "--warmup", "1", "--passes", "7", "--no-cpu-baseline",
                     "--msm-async", "3"], {"JJ_BENCH_FORCE_DIST": "1", "MASTER_PORT": str(free_port())})
    assert res["verified"] is True and res["rccl_world_size"] == 1 and res["msm_result"] == oracle_msm(1 << 16)
    assert res["config"]["msm_jobs_in_flight"] == 3 and "jj_msm_allgather" in res["config"]["parallelism"]


@pytest.mark.parametrize("workload,extra", [("varbase", ["--log2n", "15", "--no-extras"]), ("fixedbase", ["--log2n", "16"]),
                                            ("decompress", ["--log2n", "16"])])
def test_one_rank_rccl_leg_every_workload(workload, extra):
    """what the driver's multi-GPU run does for the independent-shard workloads -- init_process_group(nccl), barrier, the all_reduce(MAX)
    of the step time, the all_gather of the per-rank times -- with the one rank a 1-GPU box can hold (JJ_BENCH_FORCE_DIST=1)"""
    res = run_bench(["--gpus", "1", "--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra,
                    {"JJ_BENCH_FORCE_DIST": "1", "MASTER_PORT": str(free_port())})
    assert res["n_gpus"] == 1 and res["rccl_world_size"] == 1 and res["verified"] is True and res["all_units_equal_second_pass"] is True
    assert res["verified_block"]["ok"] is True and res["verified_block"]["units"] == 1 << 14
    r = res["rank_ms_per_step"]
    assert r["min"] <= r["max"] <= res["ms_per_step"] * 1.001
    pk = res["roofline"]["peak_samples"]
    assert len(pk["before"]) == 5 and len(pk["after"]) == 5 and pk["min"] <= pk["median"] <= pk["max"]
    assert 0 < res["roofline"]["frac_nominal"] < res["roofline"]["frac_at_peak_min"] * 1.2


def test_c_exchange_failure_falls_back_to_torch_exchange():
    """a rank whose C-level exchange cannot be set up (here: forced) takes EVERY rank to --msm-exchange torch; the line says why and
    still verifies -- the driver's first real N > 1 run of config 4 must not end rc != 0 for plumbing reasons"""
    res = run_bench(["--gpus", "1", "--workload", "msm", "--log2n", "16", "--steps", "2", "--warmup", "1", "--passes", "2", "--no-cpu-baseline",
                     "--msm-async", "2"], {"JJ_BENCH_FORCE_DIST": "1", "MASTER_PORT": str(free_port()), "JJ_BENCH_BREAK_C_EXCHANGE": "0"})
    assert res["verified"] is True and res["msm_result"] == oracle_msm(1 << 16)
    assert "JJ_BENCH_BREAK_C_EXCHANGE" in res["msm_exchange_fallback"] and "jj_msm_allgather" not in res["config"]["parallelism"]
    assert res["config"]["msm_jobs_in_flight"] == 1
    ok = run_bench(["--gpus", "1", "--workload", "msm", "--log2n", "12", "--steps", "1", "--warmup", "1", "--passes", "2", "--no-cpu-baseline"],
                   {"JJ_BENCH_FORCE_DIST": "1", "MASTER_PORT": str(free_port())})
    assert ok["msm_exchange_fallback"] is None and "jj_msm_allgather" in ok["config"]["parallelism"]


def test_eight_stacked_ranks_msm_strong_scaling_and_independent_shards():
    """the driver's widest launch, `--gpus 8`, with the eight ranks stacked on GPU 0 over gloo: config 4's shape (2^14 terms cut in eight,
    eight records gathered, one host tail) and one independent-shard workload; shard bounds, the per-rank times and the verdict come from
    all eight"""
    res = run_bench(["--gpus", "8", "--workload", "msm", "--scaling", "strong", "--log2n", "14", "--steps", "2", "--warmup", "1",
                     "--passes", "2", "--backend", "gloo", "--no-cpu-baseline"], {"JJ_BENCH_FORCE_DEVICE": "0"}, timeout=900)
    assert res["n_gpus"] == 8 and res["scaling"] == "strong" and res["verified"] is True and res["rccl_world_size"] == 8
    assert res["config"]["units_per_step"] == 2 * (1 << 14) and res["msm_result"] == oracle_msm(1 << 14)
    assert len(res["rank_ms_per_step"]["per_rank"]) == 8
    res = run_bench(["--gpus", "8", "--workload", "fixedbase", "--log2n", "13", "--steps", "2", "--warmup", "1", "--backend", "gloo",
                     "--no-cpu-baseline"], {"JJ_BENCH_FORCE_DEVICE": "0"}, timeout=900)
    assert res["n_gpus"] == 8 and res["scaling"] == "weak" and res["verified"] is True
    assert res["config"]["units_per_step"] == 8 * (1 << 13) * res["config"]["passes_per_step"] and len(res["rank_ms_per_step"]["per_rank"]) == 8


@pytest.mark.parametrize("kind", ["pinned", "pageable", "pooled"])
def test_bench_host_buffers_line(kind):
    """--host-buffers: the timed region is the C-ABI call on host arrays; the line carries roofline.pcie and is verified"""
    res = run_bench(["--workload", "fixedbase", "--log2n", "20", "--host-buffers", kind, "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {})
    assert res["pcie_inclusive"] is True and "HOST buffers" in res["metric"]
    assert res["verified"] is True and res["all_units_equal_second_pass"] is True and res["verified_block"]["ok"] is True
    pc = res["roofline"]["pcie"]
    assert pc["bytes_per_unit"] == {"h2d": 32, "d2h": 64} and 0 < pc["frac"] < 1 and pc["peak_GBps"] == 63.0
    assert 0.2 < res["host_over_device_resident"] < 1.1


def test_too_many_gpus_is_refused_not_mislabelled():
    """--gpus 8 on a box with fewer devices must fail loudly instead of reporting a 1-GPU run"""
    import torch

    if torch.cuda.device_count() >= 8:
        pytest.skip("8 devices present")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "JJ_BENCH_FORCE_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_under_torch_distributed_run_launcher():
    """the way the driver starts N > 1: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
    (ranks come from the launcher's environment, bench.py must not spawn again); two ranks stacked on GPU 0 over gloo"""
    env = dict(os.environ, JJ_BENCH_FORCE_DEVICE="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), BENCH, "--gpus", "2", "--workload", "fixedbase", "--log2n", "14", "--steps", "2", "--warmup", "1",
                        "--backend", "gloo", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # exactly one JSON line, from rank 0
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["verified"] is True and res["config"]["units_per_step"] == 2 * (1 << 14) * res["config"]["passes_per_step"]
