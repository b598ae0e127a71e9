"""
Batched Jubjub engine: thin, typed Python front-end over the C ABI (include/jubjub_hip.h).

Arguments are arrays of wire-format bytes — numpy uint8 arrays (host; staged by the library) or torch uint8 CUDA
tensors (device; zero-copy, asynchronous on torch's current stream).  Results come back as the same kind.
Shapes: scalars / field elements (n, 32); affine points (n, 64); compressed points (n, 32).
"""
import ctypes as C
import functools
import threading

import numpy as np

from . import _lib

FLAG_ZIP216 = 1
FLAG_TORSION_FREE = 2
FLAG_NOT_SMALL_ORDER = 4
FLAG_CLEAR_COFACTOR = 8


class JubjubError(RuntimeError):
    pass


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class _Arg:
    """Normalises one array argument to (pointer, keepalive)."""

    def __init__(self, x, width):
        if _is_torch(x):
            import torch

            if x.dtype != torch.uint8:
                raise TypeError("torch arguments must be uint8")
            x = x.contiguous()
            if width is not None and x.numel() % width:
                raise ValueError("byte length %d is not a multiple of %d" % (x.numel(), width))
            self.n = x.numel() // width if width else x.numel()
            self.ptr = x.data_ptr()
            self.keep = x
            self.torch = True
            self.device = x.device
        else:
            a = np.ascontiguousarray(x, dtype=np.uint8)
            if width is not None and a.size % width:
                raise ValueError("byte length %d is not a multiple of %d" % (a.size, width))
            self.n = a.size // width if width else a.size
            self.ptr = a.ctypes.data if a.size else None
            self.keep = a
            self.torch = False
            self.device = None


class _HostBlock:
    """owner of one jj_host_alloc block"""

    def __init__(self, lib, ptr):
        self._lib, self._ptr = lib, ptr

    def __del__(self):
        try:
            if self._ptr:
                self._lib.jj_host_free(C.c_void_p(self._ptr))
                self._ptr = None
        except Exception:
            pass


class FixedBaseTable:
    def __init__(self, engine, handle):
        self._engine = engine
        self._h = handle

    def close(self):
        if self._h is not None and self._engine._ctx is not None:
            self._engine._lib.jj_fixedbase_table_destroy(self._engine._ctx, self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MsmJob:
    """Handle of an MSM in flight (Engine.msm_begin); finished exactly once by Engine.msm_finish."""

    def __init__(self, handle, keep):
        self._h = handle
        self._keep = keep


class Engine:
    def __init__(self, device=0, options=None):
        """options: {key: int} for jj_ctx_set_option (include/jubjub_hip.h lists the keys), applied before the first call.  Neither this class nor
        the library reads JJ_* environment variables."""
        self._mu = threading.RLock()      # stream selection + call form one critical section per Engine (see _locked below)
        self._lib = _lib.load()
        ctx = C.c_void_p()
        rc = self._lib.jj_ctx_create(int(device), C.byref(ctx))
        if rc != 0:
            self._ctx = None
            raise JubjubError("jj_ctx_create(device=%d) failed with %d (%s)" % (
                device, rc, "no gfx950 GPU visible; there is no CPU fallback" if rc == _lib.JJ_ERR_NODEVICE else "HIP error"))
        self._ctx = ctx
        self.device = int(device)
        for key, value in (options or {}).items():
            self.set_option(key, value)

    # -------------------------------------------------------------- plumbing
    def close(self):
        if self._ctx is not None:
            self._lib.jj_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise JubjubError("libjubjub_hip error %d: %s" % (rc, self._lib.jj_last_error(self._ctx).decode()))

    def sync(self):
        self._check(self._lib.jj_ctx_sync(self._ctx))

    def set_option(self, key, value):
        """jj_ctx_set_option: per-context tuning by key ("msm_lanes", "msm_fold_min", ...); "host_tail_scalar" is process-wide."""
        if key == "host_tail_scalar":
            rc = self._lib.jj_ctx_set_option(None, key.encode(), int(value))
            if rc:
                raise JubjubError("jj_ctx_set_option(%s=%r) failed with %d" % (key, value, rc))
            return
        self._check(self._lib.jj_ctx_set_option(self._ctx, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_longlong()
        rc = self._lib.jj_ctx_get_option(None if key == "host_tail_scalar" else self._ctx, key.encode(), C.byref(v))
        if rc:
            raise JubjubError("jj_ctx_get_option(%s) failed with %d" % (key, rc))
        return v.value

    def device_info(self):
        out = (C.c_int64 * 4)()
        self._check(self._lib.jj_device_info(self._ctx, out))
        return {"cus": out[0], "clock_khz": out[1], "wavefront": out[2]}

    # -------------------------------------------------------------- page-locked host arrays (jj_host_alloc / jj_host_register)
    def host_alloc(self, shape):
        """numpy uint8 array in page-locked host memory (jj_host_alloc): host batches in such arrays are pipelined over the copy
        streams without any per-call registration.  Freed when the array (and every view of it) is gone."""
        shape = tuple(shape) if isinstance(shape, (tuple, list)) else (int(shape),)
        nbytes = int(np.prod(shape, dtype=np.int64))
        if nbytes == 0:
            return np.empty(shape, np.uint8)
        p = C.c_void_p()
        rc = self._lib.jj_host_alloc(C.c_size_t(nbytes), C.byref(p))
        if rc:
            raise JubjubError("jj_host_alloc(%d bytes) failed with %d" % (nbytes, rc))
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        buf._jj_owner = _HostBlock(self._lib, p.value)   # numpy keeps `buf` alive through .base; the block is freed with it
        return np.frombuffer(buf, dtype=np.uint8).reshape(shape)

    def result_acquire(self, shape):
        """numpy uint8 array over a page-locked result buffer from the context's pool (jj_result_acquire): pass it as `out=`; give it back with
        result_release(array) when the result has been consumed.  Several may be out at a time: every call can return a different result object."""
        shape = tuple(shape) if isinstance(shape, (tuple, list)) else (int(shape),)
        nbytes = int(np.prod(shape, dtype=np.int64))
        if nbytes == 0:
            return np.empty(shape, np.uint8)
        p = C.c_void_p()
        self._check(self._lib.jj_result_acquire(self._ctx, C.c_size_t(nbytes), C.byref(p)))
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint8).reshape(shape)

    def result_release(self, array):
        if array is not None and array.size:
            self._check(self._lib.jj_result_release(self._ctx, C.c_void_p(array.ctypes.data)))

    def result_pool_stats(self):
        nb, by, iu = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._check(self._lib.jj_result_pool_stats(self._ctx, C.byref(nb), C.byref(by), C.byref(iu)))
        return {"buffers": nb.value, "bytes": by.value, "in_use": iu.value}

    def host_register(self, array, nbytes=None):
        """page-locks an existing numpy array once (jj_host_register); pair with host_unregister(array).  The array must consist of whole pages
        (page-aligned start AND a whole number of pages: np.frombuffer over an anonymous mmap; or host_alloc instead): anything else is refused."""
        a = np.ascontiguousarray(array)
        if a is not array and a.ctypes.data != array.ctypes.data:
            raise ValueError("a contiguous array is required")
        # nbytes: the length of the mapping the array lies at the start of, when that is longer than the array (a mapping is whole pages, an array of
        # n x 64 bytes usually is not: the caller vouches that [data, data + nbytes) is mapped and its own)
        rc = self._lib.jj_host_register(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes if nbytes is None else int(nbytes)))
        if rc:
            raise JubjubError("jj_host_register failed with %d" % rc)

    def host_unregister(self, array):
        rc = self._lib.jj_host_unregister(C.c_void_p(array.ctypes.data))
        if rc:
            raise JubjubError("jj_host_unregister failed with %d" % rc)

    def profile(self, enable=True):
        self._check(self._lib.jj_ctx_profile(self._ctx, 1 if enable else 0))

    def profile_read(self, max_records=4096):
        a = (C.c_float * max_records)()
        b = (C.c_float * max_records)()
        k = C.c_int(0)
        self._check(self._lib.jj_ctx_profile_read(self._ctx, max_records, a, b, C.byref(k)))
        return [a[i] for i in range(k.value)], [b[i] for i in range(k.value)]

    def peak_imad32(self):
        out = C.c_double(0)
        self._check(self._lib.jj_peak_imad32(self._ctx, C.byref(out)))
        return out.value

    def peak_imad32_samples(self, count=7):
        """`count` single measurements of the integer multiply-add peak (jj_peak_imad32_samples), in launch order"""
        out = (C.c_double * count)()
        self._check(self._lib.jj_peak_imad32_samples(self._ctx, count, out))
        return [out[i] for i in range(count)]

    def _bind_stream(self, args):
        """When any argument is a torch CUDA tensor, run on torch's current stream."""
        if any(a.torch for a in args):
            import torch

            s = torch.cuda.current_stream(args[[a.torch for a in args].index(True)].device).cuda_stream
            self._lib.jj_ctx_set_stream(self._ctx, C.c_void_p(s))   # s == 0 is torch's default (null) stream
        else:
            self._lib.jj_ctx_use_own_stream(self._ctx)

    def _alloc(self, like, n, width):
        if like.torch:
            import torch

            t = torch.empty((n, width) if width > 1 else (n,), dtype=torch.uint8, device=like.device)
            return t, t.data_ptr()
        a = np.empty((n, width) if width > 1 else (n,), dtype=np.uint8)
        return a, (a.ctypes.data if a.size else None)

    def _call(self, name, ins, in_widths, out_widths, extra_before=(), extra_mid=(), n_override=None, out=None):
        """`out`: caller-owned result array(s) (one per output, same kind as the inputs, exact byte size) instead of fresh ones —
        e.g. page-locked buffers from host_alloc() that are reused across calls."""
        args = [_Arg(x, w) for x, w in zip(ins, in_widths)]
        n = args[0].n if n_override is None else n_override
        for a in args[1:]:
            if a.n != n and n_override is None:
                raise ValueError("length mismatch: %d vs %d" % (a.n, n))  # cf. the assert at reference src/lib.rs:841
        self._bind_stream(args)
        outs, optrs = [], []
        if out is not None:
            given = list(out) if isinstance(out, (tuple, list)) else [out]
            if len(given) != len(out_widths):
                raise ValueError("%d output arrays expected" % len(out_widths))
            for o, w in zip(given, out_widths):
                oa = _Arg(o, w)
                if oa.n != (n if n_override is None else 1) or oa.torch != args[0].torch or oa.keep is not o:
                    raise ValueError("out: a contiguous uint8 array of %d x %d bytes of the inputs' kind expected" % (n, w))
                outs.append(o)
                optrs.append(oa.ptr)
        for w in out_widths[len(outs):]:
            o, p = self._alloc(args[0], n if n_override is None else 1, w)
            outs.append(o)
            optrs.append(p)
        fn = getattr(self._lib, name)
        rc = fn(self._ctx, *extra_before, C.c_size_t(n), *[a.ptr for a in args], *extra_mid, *optrs)
        self._check(rc)
        return outs[0] if len(outs) == 1 else tuple(outs)

    # -------------------------------------------------------------- fields (reference src/fr.rs; Fq = bls12_381::Scalar)
    def field_binary(self, field, op, a, b):
        return self._call("jj_%s_%s" % (field, op), [a, b], [32, 32], [32])

    def field_unary(self, field, op, a):
        return self._call("jj_%s_%s" % (field, op), [a], [32], [32])

    def field_unary_ok(self, field, op, a):
        return self._call("jj_%s_%s" % (field, op), [a], [32], [32, 1])

    def from_bytes_wide(self, field, a):
        return self._call("jj_%s_from_bytes_wide" % field, [a], [64], [32])

    # -------------------------------------------------------------- points
    def point_double(self, p):
        return self._call("jj_point_double", [p], [64], [64])

    def point_add(self, p, q):
        return self._call("jj_point_add", [p, q], [64, 64], [64])

    def point_sub(self, p, q):
        return self._call("jj_point_sub", [p, q], [64, 64], [64])

    def point_neg(self, p):
        return self._call("jj_point_neg", [p], [64], [64])

    def mul_by_cofactor(self, p):
        return self._call("jj_point_mul_by_cofactor", [p], [64], [64])

    def to_niels(self, p):
        return self._call("jj_point_to_niels", [p], [64], [96])

    def predicate(self, name, p):
        return self._call("jj_" + name, [p], [64], [1])

    def point_sum(self, p):
        return self._sum_like("jj_point_sum", [p], [64])

    def fold_partials(self, parts):
        """Last step of an MSM cut across devices / processes: the sum of a few partial points (count x 64 bytes, numpy or a torch
        tensor on any device) on the calling host thread (jj_msm_fold_partials: the MSM's own host tail; ~5 us where the GPU
        launches of point_sum take ~180 us).  Returns the same kind of array as it was given, in host memory."""
        is_torch = type(parts).__module__.startswith("torch")
        host = np.ascontiguousarray((parts.detach().cpu().numpy() if is_torch else np.asarray(parts)).reshape(-1, 64), dtype=np.uint8)
        out = np.empty((64,), np.uint8)
        rc = self._lib.jj_msm_fold_partials(C.c_size_t(host.shape[0]), host.ctypes.data if host.shape[0] else None, out.ctypes.data)
        if rc:
            raise RuntimeError("jj_msm_fold_partials failed (%d): host pointers to 64-byte affine points expected" % rc)
        if is_torch:
            import torch

            return torch.from_numpy(out)
        return out

    def _sum_like(self, name, ins, widths):
        args = [_Arg(x, w) for x, w in zip(ins, widths)]
        n = args[0].n
        for a in args[1:]:
            if a.n != n:
                raise ValueError("length mismatch")
        self._bind_stream(args)
        out, optr = self._alloc(args[0], 1, 64)
        # zero-length numpy arrays have no data pointer; the library ignores inputs when n == 0
        rc = getattr(self._lib, name)(self._ctx, C.c_size_t(n), *[a.ptr for a in args], optr)
        self._check(rc)
        return out.reshape(64)

    # -------------------------------------------------------------- scalar multiplication
    def varbase_mul(self, scalars, points, out=None):
        return self._call("jj_varbase_mul", [scalars, points], [32, 64], [64], out=out)

    def varbase_mul_ct(self, scalars, points):
        """the constant-time ladder under the name rounds 3-4 gave it (jj_varbase_mul_ct): what varbase_mul runs by default since round 5"""
        return self._call("jj_varbase_mul_ct", [scalars, points], [32, 64], [64])

    def varbase_mul_vartime(self, scalars, points, out=None):
        """variable-time ladder for PUBLIC scalars (jj_varbase_mul_vartime): per-lane window table in memory, digit-dependent addresses"""
        return self._call("jj_varbase_mul_vartime", [scalars, points], [32, 64], [64], out=out)

    def varbase_mul_vartime_compressed(self, scalars, points, out=None):
        return self._call("jj_varbase_mul_vartime_compressed", [scalars, points], [32, 64], [32], out=out)

    def varbase_mul_scalar(self, scalar, points):
        """points[i] * scalar for one 32-byte scalar (numpy or torch), every point of the batch."""
        a, p = _Arg(scalar, 32), _Arg(points, 64)
        if a.n != 1:
            raise ValueError("scalar must be 32 bytes")
        self._bind_stream([a, p])
        out, optr = self._alloc(p, p.n, 64)
        self._check(self._lib.jj_varbase_mul_scalar(self._ctx, C.c_size_t(p.n), a.ptr, p.ptr, optr))
        return out

    def varbase_mul_compressed(self, scalars, points, out=None):
        return self._call("jj_varbase_mul_compressed", [scalars, points], [32, 64], [32], out=out)

    def fixedbase_mul_compressed(self, table, scalars, out=None):
        return self._call("jj_fixedbase_mul_compressed", [scalars], [32], [32], extra_before=(table._h,), out=out)

    def varbase_mul_exact(self, scalars, points):
        return self._call("jj_varbase_mul_exact", [scalars, points], [32, 64], [160])

    def fixedbase_table(self, base, window_bits=0):
        a = _Arg(base, 64)
        if a.n != 1:
            raise ValueError("base must be one affine point (64 bytes)")
        self._bind_stream([a])
        h = C.c_void_p()
        self._check(self._lib.jj_fixedbase_table_create(self._ctx, a.ptr, int(window_bits), C.byref(h)))
        return FixedBaseTable(self, h)

    def fixedbase_mul(self, table, scalars, out=None):
        return self._call("jj_fixedbase_mul", [scalars], [32], [64], extra_before=(table._h,), out=out)

    def fixedbase_multi_mul(self, tables, scalars):
        """out[i] = sum_j tables[j] * scalars[j][i]; scalars: (len(tables), n, 32) bytes, base-major."""
        nb = len(tables)
        a = _Arg(scalars, 32)
        if nb < 1 or a.n % nb:
            raise ValueError("scalars must hold len(tables) x n x 32 bytes")
        n = a.n // nb
        self._bind_stream([a])
        out, optr = self._alloc(a, n, 64)
        handles = (C.c_void_p * nb)(*[t._h for t in tables])
        self._check(self._lib.jj_fixedbase_multi_mul(self._ctx, handles, C.c_int(nb), C.c_size_t(n), a.ptr, optr))
        return out

    def fixedbase_composite_table(self, bases, scalar_bits):
        """One LDS table set for several bases with short scalars (jj_fixedbase_composite_create): bases (nb, 64) bytes, scalar_bits a
        list of nb bit lengths with sum(ceil((bits + 2) / 6)) <= 42."""
        a = _Arg(bases, 64)
        if a.n != len(scalar_bits) or a.n < 1:
            raise ValueError("one bit length per base")
        self._bind_stream([a])
        bits = (C.c_int * a.n)(*[int(b) for b in scalar_bits])
        h = C.c_void_p()
        self._check(self._lib.jj_fixedbase_composite_create(self._ctx, C.c_int(a.n), a.ptr, bits, C.byref(h)))
        t = FixedBaseTable(self, h)
        t.nbases = a.n
        return t

    def fixedbase_composite_mul(self, table, scalars):
        """out[i] = sum_b bases[b] * (scalars[b][i] mod 2^bits[b]) in ONE pass (43 additions per unit); scalars: (nb, n, 32) bytes."""
        a = _Arg(scalars, 32)
        nb = table.nbases
        if a.n % nb:
            raise ValueError("scalars must hold nbases x n x 32 bytes")
        n = a.n // nb
        self._bind_stream([a])
        out, optr = self._alloc(a, n, 64)
        self._check(self._lib.jj_fixedbase_composite_mul(self._ctx, table._h, C.c_size_t(n), a.ptr, optr))
        return out

    def msm(self, scalars, points):
        return self._sum_like("jj_msm", [scalars, points], [32, 64])

    def msm_dev(self, scalars, points, out=None):
        """The same sum finished ON THE DEVICE (jj_msm_dev): no host hop, nothing waits; returns a torch uint8 tensor of 64 bytes on
        the inputs' device (torch inputs), queued on torch's current stream."""
        import torch

        a, p = _Arg(scalars, 32), _Arg(points, 64)
        if a.n != p.n:
            raise ValueError("length mismatch: %d vs %d" % (a.n, p.n))
        if out is None:
            out = torch.empty(64, dtype=torch.uint8, device=a.device if a.torch else torch.device("cuda", self.device))
        self._bind_stream([a, p, _Arg(out, 64)])
        self._check(self._lib.jj_msm_dev(self._ctx, C.c_size_t(a.n), a.ptr, p.ptr, out.data_ptr()))
        return out

    def msm_begin(self, scalars, points):
        """Queues one MSM (jj_msm_begin) and returns a job; msm_finish(job) waits for it and runs the host tail.  Several jobs
        may be in flight: the host tail of one overlaps the kernels of the next.  The inputs are kept alive by the job."""
        a, p = _Arg(scalars, 32), _Arg(points, 64)
        if a.n != p.n:
            raise ValueError("length mismatch: %d vs %d" % (a.n, p.n))
        self._bind_stream([a, p])
        h = C.c_void_p()
        self._check(self._lib.jj_msm_begin(self._ctx, C.c_size_t(a.n), a.ptr, p.ptr, C.byref(h)))
        return MsmJob(h, (a.keep, p.keep))

    def msm_finish(self, job):
        """-> the 64-byte affine sum as a numpy array (host memory)."""
        if job._h is None:
            raise JubjubError("MSM job already finished")
        out = np.empty((64,), np.uint8)
        h, job._h = job._h, None
        try:
            self._check(self._lib.jj_msm_finish(h, out.ctypes.data))      # waits for the job's kernels: they run on a lane no other stream is ordered with ...
        finally:
            job._keep = None                                              # ... so the inputs are released only now (a caching allocator could hand the block on at once)
        return out

    def msm_partial(self, scalars, points, part_index=0, part_count=1):
        """First half of an MSM cut across devices / ranks (jj_msm_partial): the record of partial window sums
        (MSM_PARTIAL_BYTES bytes, same kind of array as the inputs: a CUDA tensor is ready for all_gather over RCCL).
        part_index = g, part_count = G: windows g, g + G, ... of ALL the terms given (window partition); 0 / 1: all windows
        of the terms given (term partition)."""
        a, p = _Arg(scalars, 32), _Arg(points, 64)
        if a.n != p.n:
            raise ValueError("length mismatch: %d vs %d" % (a.n, p.n))
        self._bind_stream([a, p])
        out, optr = self._alloc(a, _lib.MSM_PARTIAL_BYTES, 1)
        self._check(self._lib.jj_msm_partial(self._ctx, C.c_size_t(a.n), a.ptr, p.ptr, C.c_int(part_index), C.c_int(part_count), optr))
        return out

    def set_comm(self, comm):
        """Lends the context an RCCL communicator (jj_ctx_set_comm): `comm` is a jubjub_amd.dist.RcclComm (or None to detach)."""
        if comm is None:
            self._check(self._lib.jj_ctx_set_comm(self._ctx, None, 0, 1, None))
            self._comm = None
            return
        self._check(self._lib.jj_ctx_set_comm(self._ctx, C.c_void_p(comm.handle), C.c_int(comm.rank), C.c_int(comm.world), C.c_void_p(comm.all_gather_addr)))
        self._comm = comm                                          # keep the communicator (and its library) alive

    def msm_allgather(self, scalars, points, partition="terms"):
        """One MSM over all ranks of the communicator lent with set_comm, entirely behind the C ABI (jj_msm_allgather): record of
        window sums -> ncclAllGather over xGMI -> the gathered records folded into one on the device (from 8 ranks; below that one copy of all of them) -> one host tail.  partition "terms": the arrays are THIS
        rank's terms; "window": ALL terms on every rank.  Returns the 64-byte affine sum as a numpy array (host) on every rank."""
        a, p = _Arg(scalars, 32), _Arg(points, 64)
        if a.n != p.n:
            raise ValueError("length mismatch: %d vs %d" % (a.n, p.n))
        self._bind_stream([a, p])
        out = np.empty((64,), np.uint8)
        self._check(self._lib.jj_msm_allgather(self._ctx, C.c_size_t(a.n), a.ptr, p.ptr, C.c_int({"terms": 0, "window": 1}[partition]), out.ctypes.data))
        return out

    def msm_allgather_begin(self, scalars, points, partition="terms"):
        """msm_allgather in two halves (jj_msm_allgather_begin): queues this rank's window sums, the ncclAllGather and the fold of the
        gathered records, returns a job; msm_finish(job) waits for it and runs the host tail.  With several jobs in flight the exchange
        and the host tail of one MSM overlap the kernels of the next.  A collective: every rank begins the same jobs in the same order."""
        a, p = _Arg(scalars, 32), _Arg(points, 64)
        if a.n != p.n:
            raise ValueError("length mismatch: %d vs %d" % (a.n, p.n))
        self._bind_stream([a, p])
        h = C.c_void_p()
        self._check(self._lib.jj_msm_allgather_begin(self._ctx, C.c_size_t(a.n), a.ptr, p.ptr, C.c_int({"terms": 0, "window": 1}[partition]), C.byref(h)))
        return MsmJob(h, (a.keep, p.keep))

    def msm_combine(self, records):
        """Second half: any number of records (count x MSM_PARTIAL_BYTES bytes) -> the 64-byte affine sum as a numpy array.  A CUDA
        tensor takes jj_msm_combine_dev (the records are added on the device, ONE record is copied to the host tail); numpy / CPU
        tensors take jj_msm_combine (host only)."""
        is_torch = type(records).__module__.startswith("torch")
        if is_torch and records.is_cuda:
            # device-resident records (what all_gather delivered): folded window by window on the device, one record goes to the host tail
            t = records.detach().contiguous().view(-1, _lib.MSM_PARTIAL_BYTES)
            self._bind_stream([_Arg(t, _lib.MSM_PARTIAL_BYTES)] if t.shape[0] else [])
            out = np.empty((64,), np.uint8)
            self._check(self._lib.jj_msm_combine_dev(self._ctx, C.c_size_t(t.shape[0]), C.c_void_p(t.data_ptr()) if t.shape[0] else None, out.ctypes.data))
            return out
        host = np.ascontiguousarray((records.detach().cpu().numpy() if is_torch else np.asarray(records)).reshape(-1, _lib.MSM_PARTIAL_BYTES), dtype=np.uint8)
        out = np.empty((64,), np.uint8)
        rc = self._lib.jj_msm_combine(C.c_size_t(host.shape[0]), host.ctypes.data if host.shape[0] else None, out.ctypes.data)
        if rc:
            raise JubjubError("jj_msm_combine failed (%d): damaged or mismatched MSM records" % rc)
        return out

    # -------------------------------------------------------------- encodings
    def decompress(self, enc, flags=FLAG_ZIP216, out=None):
        return self._call("jj_decompress", [enc], [32], [64, 1], extra_mid=(C.c_uint(flags),), out=out)

    def compress(self, points):
        return self._call("jj_compress", [points], [64], [32])

    def batch_normalize(self, ext160):
        return self._call("jj_batch_normalize", [ext160], [160], [64])

    # -------------------------------------------------------------- bit decomposition (reference src/fr.rs:746-785)
    def to_le_bits(self, field, a):
        return self._call("jj_%s_to_le_bits" % field, [a], [32], [256])

    def char_le_bits(self):
        out = (C.c_uint8 * 256)()
        self._check(self._lib.jj_fr_char_le_bits(out))
        return np.frombuffer(bytes(out), dtype=np.uint8)

    # -------------------------------------------------------------- synthetic inputs (reference Group::random, src/lib.rs:1244-1298)
    def _generate(self, name, n, width, device, extra):
        """device: None -> numpy result, else a torch device -> CUDA tensor on torch's current stream"""
        if device is None:
            out = np.empty((n, width), dtype=np.uint8)
            ptr = out.ctypes.data if n else None
            self._lib.jj_ctx_use_own_stream(self._ctx)
        else:
            import torch

            out = torch.empty((n, width), dtype=torch.uint8, device=device)
            ptr = out.data_ptr()
            self._lib.jj_ctx_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(device).cuda_stream))
        self._check(getattr(self._lib, name)(self._ctx, C.c_size_t(n), *extra(ptr)))
        return out

    def synth_scalars(self, n, seed, first_index=0, device=None):
        return self._generate("jj_synth_scalars", n, 32, device, lambda p: (C.c_uint64(seed), C.c_uint64(first_index), p))

    def synth_bytes32(self, n, seed, first_index=0, device=None):
        return self._generate("jj_synth_bytes32", n, 32, device, lambda p: (C.c_uint64(seed), C.c_uint64(first_index), p))

    def random_points(self, n, seed, first_index=0, subgroup=False, device=None):
        return self._generate("jj_random_points", n, 64, device,
                              lambda p: (C.c_uint64(seed), C.c_uint64(first_index), C.c_int(1 if subgroup else 0), p, None))



def _locked(fn):
    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        with self._mu:
            return fn(self, *a, **k)
    return wrapper


# Engine methods select the launch stream and then call the library: two host threads sharing an Engine must not
# interleave those two steps (the C entry points themselves serialise on the context lock).
for _name, _fn in list(vars(Engine).items()):
    if callable(_fn) and not _name.startswith("__"):
        setattr(Engine, _name, _locked(_fn))


class MultiEngine:
    """Several devices of one node driven from one process (include/jubjub_hip.h jj_multi_*): one context, host thread and
    stream per listed device, contiguous shards, host (numpy) arrays in and out; the MSM's per-device partial points are
    added on the host.  A device may be listed more than once."""

    def __init__(self, devices):
        self._lib = _lib.load()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self._lib.jj_multi_create(devs, C.c_int(len(devices)), C.byref(h))
        self._h = None
        if rc != 0:
            raise JubjubError("jj_multi_create(%r) failed with %d" % (list(devices), rc))
        self._h = h
        self._tables = []

    def close(self):
        if self._h is not None:
            for t in self._tables:
                self._lib.jj_multi_fixedbase_table_destroy(self._h, t)
            self._tables = []
            self._lib.jj_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise JubjubError("rc=%d: %s" % (rc, (self._lib.jj_multi_last_error(self._h) or b"").decode()))

    @property
    def device_count(self):
        return self._lib.jj_multi_device_count(self._h)

    @staticmethod
    def _np(x, width):
        a = np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, width)
        return a, (a.ctypes.data if a.size else None)

    @staticmethod
    def _out(out, shape):
        """a caller-owned result array (e.g. page-locked, reused across calls) or a fresh one"""
        if out is None:
            return np.empty(shape, np.uint8)
        if out.dtype != np.uint8 or not out.flags["C_CONTIGUOUS"] or out.size != int(np.prod(shape)):
            raise ValueError("out: a contiguous uint8 array of shape %r expected" % (shape,))
        return out

    def varbase_mul(self, scalars, points, out=None):
        s, sp = self._np(scalars, 32)
        p, pp = self._np(points, 64)
        if len(s) != len(p):
            raise ValueError("length mismatch")
        out = self._out(out, (len(s), 64))
        self._check(self._lib.jj_multi_varbase_mul(self._h, C.c_size_t(len(s)), sp, pp, out.ctypes.data if len(s) else None))
        return out

    def fixedbase_table(self, base, window_bits=0):
        b, bp = self._np(base, 64)
        t = C.c_void_p()
        self._check(self._lib.jj_multi_fixedbase_table_create(self._h, bp, C.c_int(window_bits), C.byref(t)))
        self._tables.append(t)
        return t

    def fixedbase_mul(self, table, scalars, out=None):
        s, sp = self._np(scalars, 32)
        out = self._out(out, (len(s), 64))
        self._check(self._lib.jj_multi_fixedbase_mul(self._h, table, C.c_size_t(len(s)), sp, out.ctypes.data if len(s) else None))
        return out

    def decompress(self, enc, flags=FLAG_ZIP216, out=None):
        e, ep = self._np(enc, 32)
        out, ok = (self._out(out[0], (len(e), 64)), self._out(out[1], (len(e),))) if out is not None else (np.empty((len(e), 64), np.uint8), np.empty((len(e),), np.uint8))
        self._check(self._lib.jj_multi_decompress(self._h, C.c_size_t(len(e)), ep, C.c_uint(flags), out.ctypes.data if len(e) else None,
                                                  ok.ctypes.data if len(e) else None))
        return out, ok

    def msm(self, scalars, points):
        s, sp = self._np(scalars, 32)
        p, pp = self._np(points, 64)
        if len(s) != len(p):
            raise ValueError("length mismatch")
        out = np.empty((64,), np.uint8)
        self._check(self._lib.jj_multi_msm(self._h, C.c_size_t(len(s)), sp, pp, out.ctypes.data))
        return out
