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


def msm_distributed(engine, scalars, points, group=None, presharded=False):
    """sum_i points[i] * scalars[i] over ALL ranks' terms; every rank returns the same 64-byte affine point.

    presharded=False: every rank holds the full arrays and reduces its own contiguous slice.
    presharded=True : every rank passes only its own terms."""
    import torch
    import torch.distributed as dist

    rank, world = _rank_world(group)
    if not presharded:
        lo, hi = shard_bounds(len(scalars), rank, world)
        scalars, points = scalars[lo:hi], points[lo:hi]
    part = engine.msm(scalars, points)                      # 64 bytes, identity for an empty shard
    if world == 1:
        return part
    is_torch = type(part).__module__.startswith("torch")
    t = part if is_torch else torch.from_numpy(part.copy())
    t = t.reshape(64).contiguous()
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t, group=group)               # world x 64 B; latency-bound, not bandwidth-bound
    stacked = torch.stack(gathered)
    # world partial points -> one: a short dependent chain, done on the host by the library's MSM tail when the engine offers it
    fold = getattr(engine, "fold_partials", None) or engine.point_sum
    total = fold(stacked if is_torch else stacked.numpy())
    return total
