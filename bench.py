#!/usr/bin/env python3
"""
bench.py — headline benchmark of the MI355X Jubjub engine (BASELINE.json metric: Jubjub scalar-muls/sec).

  python bench.py --gpus N --steps K --warmup W [--workload varbase|fixedbase|msm|decompress] [--log2n L]

One "step" = one pass of the hot path over one batch of synthetic inputs already resident in HBM.
N = 1 default workload: BASELINE.json configs[1] — 2^20 variable-base scalar-muls (random 252-bit scalars x random
curve points).  N > 1: one process per GPU (launched by torch.distributed.run), every rank runs the same batch
size on its own shard (weak scaling, no data-path collective for the independent-batch workloads; the MSM
workload all-gathers one 64-byte partial point per rank over RCCL).  Rank 0 prints ONE JSON line.

The JSON carries:
  roofline     integer-VALU roofline of the dominant kernel (this path is carry-free integer multiply-add work,
               not HBM- or MFMA-bound): achieved = algorithmic IMAD32/s with the SURVEY §8(d) convention
               (field mul = 128, square = 100 IMAD32) for the algorithm the kernel actually runs, divided by the
               kernel's HIP-event duration; peak = v_mad_u64_u32 rate measured live on this device
               (jj_peak_imad32).  `hbm` gives the algorithmic HBM bytes/s beside the 8 TB/s peak to show the
               kernel is not memory-limited.
  cpu_baseline the oracle's C port of the reference algorithm (exact 252-step ladder, 4x64 Montgomery limbs)
               timed with OpenMP on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec

# Field-operation counts of the algorithms the kernels actually run, per unit.
# IMAD32 convention: M = 128, S = 100.
WORK = {
    # signed 5-bit windows: table {1..16}P 139M (15 mixed adds + 17 to_niels); 51 adds x 8M; 250 dbl x (4S+3M); load 2M;
    # normalise 7M + (255S+78M)/32 inversion share
    "varbase": {"S": 250 * 4 + 8, "M": 139 + 51 * 8 + 250 * 3 + 2 + 7 + 3, "bytes": 32 + 64 + 64},
    # 43 mixed adds x 7M ; normalise as above
    "fixedbase": {"S": 8, "M": 43 * 7 + 7 + 3, "bytes": 32 + 64},
    # Pippenger, c = 16: per term 2M load + 2M to_niels + 16 windows x 7M mixed add; bucket reduce 2 x 10M per bucket
    # (16 x 2^15 buckets / 2^20 terms -> +10M); the 240-doubling Horner tail is per MSM (on the host), not per term
    "msm": {"S": 0, "M": 2 + 2 + 16 * 7 + 10, "bytes": 32 + 64},
    # decode kernel only (roofline.kernel_ms is k_decompress; the flag kernels run after it and show up in tail_ms):
    # two decode passes (2 x (1M + 1S + 1M)), shared inversion (3M + (253S+61M)/32), u^2 1M,
    # sqrt = a^((t-1)/2) (220S + 52M, sliding windows) + 2M + 24S + 6M digit extraction + 4 canon + 4M table multiplies
    # + verify (1S + 2M), 3 to_words
    "decompress": {"S": 2 + 8 + 220 + 24 + 1, "M": 4 + 3 + 2 + 1 + 52 + 2 + 6 + 4 + 4 + 2 + 3, "bytes": 32 + 65},
}
# the reference's own algorithm for comparison in the JSON
GEN_U = 0x62EDCBB8BF3787C88B0F03DDD60A8187CAF55D1B29BF81AFE4B3D35DF1A7ADFE   # generator (u, 11), reference src/lib.rs:1380-1396
REFERENCE_WORK = {"varbase": {"S": 1008, "M": 2774}, "fixedbase": {"S": 1008, "M": 2520}}


def imad32(w):
    return 100 * w["S"] + 128 * w["M"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="varbase", choices=sorted(WORK))
    ap.add_argument("--log2n", type=int, default=None, help="log2 of the per-GPU batch (default: 20 varbase/msm, 24 fixedbase, 23 decompress = 2^26 over 8 GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="varbase workload: skip the fixed-base side measurements (clean per-kernel profiles)")
    ap.add_argument("--fb-window", type=int, default=0, help="fixed-base window bits: 0/6 = LDS-staged constant-time table (default), 8..12 = L2-resident table")
    ap.add_argument("--decompress-flags", type=int, default=13,
                    help="jj_decompress flags: 1 ZIP-216 | 2 torsion-free (order-8 Tate pairing; JJ_TORSION_CHECK=ladder for the [r]P ladder) | 4 reject small order | 8 clear cofactor (default 13 = BASELINE config 5: decode + small-order check + mul_by_cofactor)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target wall time of the CPU baseline sample")
    return ap.parse_args()


def cpu_baseline(workload, target_s):
    """Times the oracle's C port of the reference algorithm on the host cores (reported baseline, not a target)."""
    import numpy as np

    from oracle import c_oracle as O
    from oracle import jubjub_ref as J

    cores = os.cpu_count() or 1
    try:
        import ctypes

        omp = ctypes.CDLL("libgomp.so.1")
        cores = int(omp.omp_get_max_threads())
    except Exception:
        pass
    base = np.frombuffer(J.GENERATOR[0].to_bytes(32, "little") + J.GENERATOR[1].to_bytes(32, "little"), dtype=np.uint8)
    rng = np.random.default_rng(2024)

    def run(n):
        s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        s[:, 31] &= 0x0F
        if workload == "fixedbase":
            t0 = time.perf_counter(); O.fixedbase_mul(s, base); return time.perf_counter() - t0
        pts = O.fixedbase_mul(s[::-1].copy(), base)
        if workload == "decompress":
            enc = O.compress(pts)
            t0 = time.perf_counter(); O.decompress(enc, 1); return time.perf_counter() - t0
        if workload == "msm":
            t0 = time.perf_counter(); O.msm(s, pts); return time.perf_counter() - t0
        t0 = time.perf_counter(); O.varbase_mul(s, pts); return time.perf_counter() - t0

    probe = max(64, 16 * cores)
    t = run(probe)                                                  # warm-up + first calibration
    n = int(max(probe, min(1 << 18, probe * 1.0 / max(t, 1e-6))))   # ~1 s sample for a stable rate estimate
    t = run(n)
    n = int(max(probe, min(1 << 22, n * target_s / max(t, 1e-6))))  # the reported sample: ~target_s of wall time
    t = run(n)
    unit = {"varbase": "scalar-muls/s", "fixedbase": "scalar-muls/s", "msm": "terms/s", "decompress": "points/s"}[workload]
    return {"value": n / t, "unit": unit, "cores": cores, "kind": "port",
            "sample": "%d units of the same synthetic workload, reference algorithm (exact 252-step ladder / per-point decode), "
                      "oracle/jubjub_oracle.c -O3 + OpenMP, %.1f s wall" % (n, t)}


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    from jubjub_amd import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("JJ_BENCH_FORCE_DIST") == "1"   # the latter: 1-rank RCCL plumbing check
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=a.backend)
    n_gpus = world if distributed else 1
    if a.gpus != n_gpus and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; using %d" % (a.gpus, world, n_gpus), file=sys.stderr)
    dev_index = int(os.environ.get("JJ_BENCH_FORCE_DEVICE", local_rank))   # plumbing tests: several ranks on one GPU
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    eng = Engine(dev_index)

    wl = a.workload
    log2n = a.log2n if a.log2n is not None else {"varbase": 20, "fixedbase": 24, "msm": 20, "decompress": 23}[wl]
    n = 1 << log2n

    # ---- synthetic inputs, generated on the device, resident in HBM before the timed region
    g = torch.Generator(device=dev)
    g.manual_seed(0x4A55424A5542 + rank)
    scalars = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    scalars[:, 31] &= 0x0F                                          # uniform below 2^252 (reference ladder width)
    base = torch.from_numpy(np.frombuffer(GEN_U.to_bytes(32, "little") + (11).to_bytes(32, "little"), dtype=np.uint8).copy()).to(dev)
    table = eng.fixedbase_table(base, a.fb_window)
    points = None
    if wl in ("varbase", "msm", "decompress"):
        ks = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
        points = eng.fixedbase_mul(table, ks)                       # random points of the full group (order 8r), on-curve by construction
        assert bool(eng.predicate("is_on_curve", points[:4096]).all())
    enc = eng.compress(points) if wl == "decompress" else None

    def step():
        if wl == "varbase":
            return eng.varbase_mul(scalars, points)
        if wl == "fixedbase":
            return eng.fixedbase_mul(table, scalars)
        if wl == "decompress":
            return eng.decompress(enc, a.decompress_flags)
        part = eng.msm(scalars, points)                             # one partial point per rank
        if distributed:
            if a.backend == "nccl":
                parts = [torch.empty_like(part) for _ in range(world)]
                dist.all_gather(parts, part)                        # 64 B per rank over RCCL/xGMI; EC add is not a reduce op
                part = eng.point_sum(torch.stack(parts))
            else:
                parts = [torch.empty(64, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, part.cpu())
                part = eng.point_sum(torch.stack(parts).to(dev))
        return part

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    barrier()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    main_ms, tail_ms = eng.profile_read()
    eng.profile(False)
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    units_per_step = n * n_gpus
    value = units_per_step * a.steps / dt
    res = {
        "metric": "Jubjub scalar-muls/sec (%s)" % wl if wl in ("varbase", "fixedbase") else "Jubjub %s units/sec" % wl,
        "value": value,
        "unit": {"varbase": "scalar-muls/s", "fixedbase": "scalar-muls/s", "msm": "terms/s", "decompress": "points/s"}[wl],
        "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32x9 (29-bit limbs, v_mad_u64_u32 integer multiply-add)",
        "data": "synthetic",
        "config": {"workload": "%s, 2^%d units per GPU per step (BASELINE.json configs[%d])" % (
            wl, log2n, {"varbase": 1, "fixedbase": 2, "msm": 3, "decompress": 4}[wl]),
            "scalars": "uniform 252-bit", "points": "random points of the full group (order 8r), affine 64 B",
            "parallelism": "independent shards, one process per GPU" + ("; all_gather of 64 B partial points" if wl == "msm" else "")},
    }
    if rank == 0:
        w = dict(WORK[wl])
        if wl == "fixedbase" and a.fb_window >= 8:
            w["M"] = -(-253 // a.fb_window) * 7 + 10          # ceil(253/w) mixed additions + normalise
        if main_ms:
            kern_ms = sum(main_ms) / len(main_ms)
            tail = sum(tail_ms) / len(tail_ms)
        else:                                                       # workloads without the event hooks: whole step
            kern_ms, tail = dt / a.steps * 1e3, 0.0
        peak = eng.peak_imad32()
        achieved = n * imad32(w) / (kern_ms * 1e-3)
        res["roofline"] = {
            "bound": "valu_int32", "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "TIMAD32/s",
            "frac": achieved / peak,
            # multiply-adds actually issued by the 9x29-bit representation (162 per mul, 126 per square) / measured peak
            "mad_issue_frac": n * (162 * w["M"] + 126 * w["S"]) / (kern_ms * 1e-3) / peak,
            "traffic": None,
            "kernel": {"varbase": "k_varbase", "fixedbase": "k_fixedbase" if a.fb_window < 8 else "k_fixedbase_gather(w=%d)" % a.fb_window, "msm": "k_msm_accumulate (+sort/fix-up/reduce; Horner on the host)", "decompress": "k_decompress"}[wl],
            "kernel_ms": kern_ms, "tail_ms": tail,
            "work_per_unit": {"field_squares": w["S"], "field_muls": w["M"], "imad32": imad32(w), "convention": "M=128,S=100 (SURVEY 8d)"},
            "reference_algorithm_imad32": imad32(REFERENCE_WORK[wl]) if wl in REFERENCE_WORK else None,
            "hbm": {"achieved": n * w["bytes"] / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": n * w["bytes"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "bytes_per_unit": w["bytes"]},
        }
        if wl == "varbase" and log2n == 20:
            # not collected live (PMC needs rocprofv3): the committed counter passes of this same launch
            res["roofline"]["traffic_profiled"] = {"bytes_per_launch": 11.39e9 + 3.08e9, "fetch_bytes": 11.39e9, "write_bytes": 3.08e9,
                                                   "source": "profiles/r1_varbase_pmc.txt (FETCH_SIZE x2 per the gfx950 note, WRITE_SIZE; separate --pmc passes)",
                                                   "note": "per-lane window tables; algorithmic I/O is 160 B per unit"}
        if wl == "varbase" and not a.no_extras:
            # the other half of BASELINE.json's metric, measured in the same process (outside the timed region above)
            fs = scalars if log2n >= 22 else torch.randint(0, 256, (1 << 22, 32), dtype=torch.uint8, device=dev, generator=g)
            eng.fixedbase_mul(table, fs)
            torch.cuda.synchronize(dev)
            eng.profile(True)
            t1 = time.perf_counter()
            for _ in range(5):
                eng.fixedbase_mul(table, fs)
            torch.cuda.synchronize(dev)
            fdt = (time.perf_counter() - t1) / 5
            fm, _ft = eng.profile_read()
            eng.profile(False)
            fw = WORK["fixedbase"]
            res["fixed_base"] = {"value": fs.shape[0] / fdt, "unit": "scalar-muls/s per GPU", "units_per_step": int(fs.shape[0]),
                                 "ms_per_step": fdt * 1e3, "kernel_ms": sum(fm) / max(len(fm), 1),
                                 "roofline_frac": fs.shape[0] * imad32(fw) / (sum(fm) / max(len(fm), 1) * 1e-3) / peak,
                                 "window_select": "LDS-staged table, ds_bpermute constant-time select"}
            wt = eng.fixedbase_table(base, 16)                      # wide-window alternative (64 MB table in the Infinity Cache, per-lane gather)
            eng.fixedbase_mul(wt, fs)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(5):
                eng.fixedbase_mul(wt, fs)
            torch.cuda.synchronize(dev)
            res["fixed_base_wide_window"] = {"value": fs.shape[0] * 5 / (time.perf_counter() - t1), "unit": "scalar-muls/s per GPU",
                                              "window_bits": 16, "table": "64 MB (one 128-byte line per entry), Infinity-Cache resident, variable-time gather: 16 additions per scalar"}
            wt.close()
        if not a.no_cpu_baseline and n_gpus == 1:
            res["cpu_baseline"] = cpu_baseline(wl, a.cpu_seconds)
        print(json.dumps(res))
    table.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
