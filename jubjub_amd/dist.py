"""
Multi-GPU host logic: one process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" on CPU
for tests).  The independent-batch workloads shard with NO data-path collective; MSM all-gathers one record of partial
window sums per rank (JJ_MSM_PARTIAL_BYTES = 8256 bytes; elliptic-curve addition is not an RCCL reduction operator),
folds the gathered records into one on the device (from 8 ranks; below that they are copied to the host once) and every rank runs
one host tail (jj_msm_combine / jj_msm_combine_dev).
C / C++ / Rust callers do the same exchange without Python: jj_ctx_set_comm + jj_msm_allgather, or in two halves for a stream of
MSMs, jj_msm_allgather_begin + jj_msm_finish (include/jubjub_hip.h, examples/msm_rccl.cpp; Engine.msm_allgather / msm_allgather_begin).

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


# ---------------------------------------------------------------------------------------------------------------- RCCL at the C level
class RcclComm:
    """An RCCL communicator of this process's own (ncclCommInitRank through ctypes), for the exchange that runs entirely behind the C
    ABI (Engine.set_comm + Engine.msm_allgather = jj_ctx_set_comm + jj_msm_allgather).  torch.distributed keeps its communicator
    private, so the ncclUniqueId of this one is created on rank 0 and carried to the other ranks by `broadcast` (a callable: bytes on
    rank 0 / None elsewhere -> the 128 bytes on every rank; default: torch.distributed.broadcast_object_list).  The library used is
    the RCCL the process already has (PyTorch's bundled librccl.so), else /opt/rocm's: its ncclAllGather goes to the context with
    the communicator, because the two must come from the same library instance.
    The current HIP device (torch.cuda.set_device) must be this rank's GPU when the communicator is created."""

    def __init__(self, rank, world, broadcast=None, lib_path=None, agree=None):
        """`agree` (a callable: this rank's error text or None -> the list of every rank's) makes the set-up fail on ALL ranks when it
        fails on one, BEFORE the collective ncclCommInitRank could leave the healthy ranks waiting for the broken one (default:
        torch.distributed.all_gather_object; not called for world == 1)."""
        import ctypes as C

        self.rank, self.world = int(rank), int(world)
        self.handle = None

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]

        # stage 1, local: the library, its symbols and (rank 0) the id -- nothing here waits for another rank
        err, raw = None, None
        uid = UniqueId()
        try:
            self._lib = C.CDLL(lib_path or self._find_library())
            self._lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
            self._lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
            self._lib.ncclCommDestroy.argtypes = [C.c_void_p]
            self._lib.ncclGetErrorString.restype = C.c_char_p
            self.all_gather_addr = C.cast(self._lib.ncclAllGather, C.c_void_p).value
            if self.rank == 0:
                self._ok(self._lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
                raw = C.string_at(C.addressof(uid), 128)
        except (OSError, AttributeError, RuntimeError) as e:
            err = "%s: %s" % (type(e).__name__, e)
        # stage 2: every rank learns whether every rank got this far (the broadcast below is only entered when all did)
        if self.world > 1:
            errs = (agree or self._torch_agree)(err)
            bad = [(r, e) for r, e in enumerate(errs) if e]
            if bad:
                raise RuntimeError("RCCL set-up failed on rank %d: %s" % bad[0])
            raw = (broadcast or self._torch_broadcast)(raw)
        elif err:
            raise RuntimeError("RCCL set-up failed: " + err)
        C.memmove(C.addressof(uid), raw, 128)
        h = C.c_void_p()
        self._ok(self._lib.ncclCommInitRank(C.byref(h), self.world, uid, self.rank), "ncclCommInitRank")
        self.handle = h.value

    @staticmethod
    def _find_library():
        import importlib.util
        import os

        try:
            spec = importlib.util.find_spec("torch")
        except (ImportError, ValueError):
            spec = None
        if spec is not None and spec.submodule_search_locations:
            cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
            if os.path.exists(cand):
                return cand
        return "librccl.so.1"

    @staticmethod
    def _torch_broadcast(raw):
        import torch.distributed as dist

        box = [raw]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    @staticmethod
    def _torch_agree(err):
        import torch.distributed as dist

        every = [None] * dist.get_world_size()
        dist.all_gather_object(every, err)
        return every

    def _ok(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self._lib.ncclGetErrorString(rc).decode()))

    def close(self):
        if getattr(self, "handle", None):
            self._lib.ncclCommDestroy(self.handle)
            self.handle = None
