"""
Multi-GPU host logic: one process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" on CPU
for tests).  The independent-batch workloads shard with NO data-path collective; MSM exchanges one 64-byte
affine partial point per rank with all_gather (elliptic-curve addition is not an RCCL reduction operator) and
every rank folds the partial points locally.

`engine` is any object with the Engine methods used here (varbase_mul, fixedbase_mul, decompress, msm,
point_sum); production code passes jubjub_amd.Engine — the CPU tests pass an oracle-backed stand-in.
"""


def shard_bounds(n, rank, world):
    """Contiguous slice [lo, hi) of n units owned by `rank` (sizes differ by at most one)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _rank_world(group=None):
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def sharded_map(fn, arrays, group=None):
    """Applies fn to this rank's slice of every array (no communication).  Returns (lo, hi, result)."""
    rank, world = _rank_world(group)
    n = len(arrays[0])
    lo, hi = shard_bounds(n, rank, world)
    return lo, hi, fn(*[a[lo:hi] for a in arrays])


def varbase_mul_sharded(engine, scalars, points, group=None):
    return sharded_map(engine.varbase_mul, [scalars, points], group)


def fixedbase_mul_sharded(engine, table, scalars, group=None):
    return sharded_map(lambda s: engine.fixedbase_mul(table, s), [scalars], group)


def decompress_sharded(engine, enc, flags=1, group=None):
    return sharded_map(lambda e: engine.decompress(e, flags), [enc], group)


def msm_distributed(engine, scalars, points, group=None, presharded=False, partition="terms"):
    """MSM over all ranks.  Every rank reduces its share to a RECORD of window sums that stays on its device
    (engine.msm_partial: 8 KB), the records are all-gathered (RCCL: elliptic-curve addition is not a reduction operator), copied to
    the host ONCE, and every rank runs ONE host tail over all of them (engine.msm_combine: window sums added window by window, one
    Horner chain, one inversion).
      partition="terms"   rank g owns the terms [g n/G, (g+1) n/G) and all their windows (presharded: the arrays given are already
                          this rank's terms)
      partition="window"  rank g owns windows g, g + G, ... of ALL terms (every rank holds the whole batch): its sort and bucket
                          reduce shrink G-fold, which term sharding does not give
    Returns the 64-byte affine sum as a numpy array (host memory) on every rank."""
    import torch
    import torch.distributed as dist

    rank, world = _rank_world(group)
    if partition == "window":
        rec = engine.msm_partial(scalars, points, rank, world)
    elif partition == "terms":
        if not presharded:
            lo, hi = shard_bounds(len(scalars), rank, world)
            scalars, points = scalars[lo:hi], points[lo:hi]
        rec = engine.msm_partial(scalars, points)                # a valid record without windows for an empty shard
    else:
        raise ValueError("partition must be 'terms' or 'window'")
    if world == 1:
        return engine.msm_combine(rec)
    is_torch = type(rec).__module__.startswith("torch")
    t = rec if is_torch else torch.from_numpy(rec.copy())
    t = t.reshape(-1).contiguous()
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t, group=group)                   # world x 8 KB; latency-bound, not bandwidth-bound
    stacked = torch.stack(gathered)
    return engine.msm_combine(stacked if is_torch else stacked.numpy())
