"""
The path a drop-in caller takes: HOST arrays handed to the C ABI (include/jubjub_hip.h "host buffers"): page-locked buffers from
jj_host_alloc, buffers registered once with jj_host_register, and plain pageable memory, each through the chunked copy / compute
pipeline (run_pipelined, jj_engine.h) -- bit-exact against the device-resident path and the oracle.
Also the C-level multi-rank MSM exchange (jj_ctx_set_comm + jj_msm_allgather, examples/msm_rccl.cpp) with one rank over RCCL.
Boundary semantics: /root/reference/src/lib.rs:873-879 (ExtendedPoint * Fr), 1109-1115 (AffineNielsPoint * Fr), 469-627 (from_bytes).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import c_oracle as O
from oracle import jubjub_ref as J
from util import pt64, rand_points, rand_scalars

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PIPE = (1 << 19) + (1 << 18) + 77          # pipelined: cut into chunks of 2^17 units, the last one ragged


@pytest.fixture(scope="module")
def eng():
    from jubjub_amd import Engine

    e = Engine(0)
    yield e
    e.close()


def _inputs(n, seed):
    s = rand_scalars(seed, n)
    small = rand_points(seed + 1, 4096)
    p = small[np.arange(n) % 4096]                      # 4096 distinct points repeated: the oracle side stays cheap
    return s, np.ascontiguousarray(p)


def _own_mapping(shape):
    """a uint8 array that is a mapping of its own (anonymous mmap, page-aligned, written once): what include/jubjub_hip.h asks callers to hand
    to jj_host_register -- an array on the C heap shares its first and last page with other heap objects, and page-locking such pages for the
    GPU and releasing them again left the runtime with stale mappings that a later transfer into recycled heap memory tripped over
    (GPU memory access fault in this file, once in ~6 runs, until round 4)"""
    import mmap

    nbytes = int(np.prod(shape))
    m = mmap.mmap(-1, max(nbytes, mmap.PAGESIZE))
    a = np.frombuffer(m, dtype=np.uint8, count=nbytes).reshape(shape)
    a[...] = 0
    return a


def _pages(nbytes):
    import mmap

    return (max(nbytes, 1) + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE


def _host(eng, kind, a):
    """the array in the requested kind of host memory: 'pinned' (jj_host_alloc), 'registered' (jj_host_register), 'pageable'"""
    if kind == "pinned":
        h = eng.host_alloc(a.shape)
        h[...] = a
        return h
    if kind == "registered":
        h = _own_mapping(a.shape)
        h[...] = a
        eng.host_register(h, nbytes=_pages(h.nbytes))           # whole pages: the anonymous mapping behind h is that long
        return h
    return np.array(a, copy=True)


@pytest.mark.parametrize("kind", ["pinned", "registered", "pageable"])
def test_varbase_fixedbase_host_pipeline(eng, kind):
    import torch

    n = N_PIPE
    s, p = _inputs(n, 901)
    hs, hp = _host(eng, kind, s), _host(eng, kind, p)
    out = _host(eng, kind, np.zeros((n, 64), np.uint8))
    out32 = _host(eng, kind, np.zeros((n, 32), np.uint8))
    try:
        r = eng.varbase_mul(hs, hp, out=out)
        assert r is out
        dev = eng.varbase_mul(torch.from_numpy(s).cuda(), torch.from_numpy(p).cuda()).cpu().numpy()
        assert (out == dev).all()
        idx = np.concatenate([np.arange(0, n, 4099), [n - 1, (1 << 18) - 1, 1 << 18, (1 << 19) - 1, 1 << 19]])
        assert (out[idx] == O.varbase_mul(s[idx], p[idx])).all()
        eng.varbase_mul_compressed(hs, hp, out=out32)
        assert (out32[idx] == O.compress(O.varbase_mul(s[idx], p[idx]))).all()
        if kind == "pageable":                                   # the variable-time table ladder through the same pipeline
            out[...] = 0
            eng.varbase_mul_vartime(hs, hp, out=out)
            assert (out == dev).all()
            eng.varbase_mul_vartime_compressed(hs, hp, out=out32)
            assert (out32[idx] == O.compress(O.varbase_mul(s[idx], p[idx]))).all()
        base = pt64(J.GENERATOR)
        tab = eng.fixedbase_table(base)
        eng.fixedbase_mul(tab, hs, out=out)
        dev = eng.fixedbase_mul(tab, torch.from_numpy(s).cuda()).cpu().numpy()
        assert (out == dev).all()
        assert (out[idx] == O.fixedbase_mul(s[idx], base)).all()
        eng.fixedbase_mul_compressed(tab, hs, out=out32)
        assert (out32[idx] == O.compress(O.fixedbase_mul(s[idx], base))).all()
        tab.close()
    finally:
        if kind == "registered":
            for a in (hs, hp, out, out32):
                eng.host_unregister(a)


@pytest.mark.parametrize("kind", ["pinned", "pageable"])
@pytest.mark.parametrize("flags", [1, 13])
def test_decompress_host_pipeline(eng, kind, flags):
    """jj_decompress on host arrays runs the chunked pipeline too (points + validity bytes come back per chunk)."""
    import torch

    n = N_PIPE
    pts = rand_points(77, 4096)[np.arange(n) % 4096]
    enc = O.compress(np.ascontiguousarray(pts))
    rng = np.random.default_rng(5)
    bad = rng.integers(0, n, size=n // 16)
    enc[bad] = rng.integers(0, 256, size=(len(bad), 32), dtype=np.uint8)         # off-curve / non-canonical noise
    he = _host(eng, kind, enc)
    out = _host(eng, kind, np.zeros((n, 64), np.uint8))
    ok = _host(eng, kind, np.zeros((n,), np.uint8))
    eng.decompress(he, flags, out=(out, ok))
    d_out, d_ok = eng.decompress(torch.from_numpy(enc).cuda(), flags)
    assert (ok == d_ok.cpu().numpy()).all() and (out == d_out.cpu().numpy()).all()
    idx = np.concatenate([np.arange(0, n, 1021), bad[:200], [n - 1, (1 << 18) - 1, 1 << 18]])
    eo, ek = O.decompress(enc[idx], flags)
    assert (ok[idx] == ek).all() and (out[idx] == eo).all()
    assert 0 < int(ok.sum()) < n


def test_pageable_arrays_page_locked_in_place(monkeypatch):
    """JJ_PIPE_PAGEABLE=register: pageable arrays of 64 MB and more are page-locked in place for the call instead of passing through the staging
    buffers (round 3's way, kept as an option; smaller arrays -- C-heap memory -- are staged in this mode too); with uniform chunks and a freshly
    allocated result array, at sizes where all three arrays, none, and only the 64-byte ones are registered."""
    import torch

    from jubjub_amd import Engine

    opts = {}
    opts['pipe_pageable_register'] = 1
    opts['pipe_ramp'] = 0
    e = Engine(0, options=opts)
    for n in (1 << 21, N_PIPE, (1 << 20) + 5):                  # all arrays registered | all staged | points + result registered, scalars staged
        s, p = _inputs(n, 77)
        out = e.varbase_mul(s, p)                               # a new numpy result array
        dev = e.varbase_mul(torch.from_numpy(s).cuda(), torch.from_numpy(p).cuda()).cpu().numpy()
        assert (out == dev).all(), n
    e.close()


def test_host_pipeline_chunk_schedule_edges(eng):
    """batches just around the sizes where the chunk schedule changes shape (4 chunks, the short first / last chunk, a ragged tail),
    pageable (bounce path: three staging slots) and page-locked"""
    import torch

    base = pt64(J.GENERATOR)
    tab = eng.fixedbase_table(base)
    for n in ((1 << 18) - 1, 1 << 18, (1 << 18) + 1, (1 << 20) + 4097, (1 << 22) + 3):
        s = rand_scalars(1000 + (n & 0xffff), n)
        want = eng.fixedbase_mul(tab, torch.from_numpy(s).cuda()).cpu().numpy()
        assert (eng.fixedbase_mul(tab, s) == want).all(), n                                   # pageable in, fresh pageable out
        hs, ho = _host(eng, "pinned", s), eng.host_alloc((n, 64))
        assert (eng.fixedbase_mul(tab, hs, out=ho) == want).all(), n
    tab.close()


def test_host_alloc_api(eng):
    lib = eng._lib
    p = C.c_void_p()
    assert lib.jj_host_alloc(0, C.byref(p)) == 0 and not p.value
    assert lib.jj_host_alloc(1 << 20, C.byref(p)) == 0 and p.value
    assert lib.jj_host_free(p) == 0
    assert lib.jj_host_free(None) == 0
    assert lib.jj_host_register(None, 16) != 0
    heap = np.zeros((1 << 20) + 64, np.uint8)
    off = 16 if heap.ctypes.data % 4096 == 0 else 0
    assert lib.jj_host_register(C.c_void_p(heap.ctypes.data + off), C.c_size_t(1 << 20)) != 0       # not page-aligned: a C-heap array is refused
    a = _own_mapping((1 << 20,))
    assert lib.jj_host_register(C.c_void_p(a.ctypes.data), C.c_size_t(5000)) != 0                    # a page-aligned start, but not whole pages: refused
    eng.host_register(a)
    eng.host_unregister(a)
    assert lib.jj_host_unregister(C.c_void_p(a.ctypes.data)) != 0            # not registered any more
    h = eng.host_alloc((1000, 32))
    v = h[10:20]
    del h
    v[...] = 7                                                               # a view keeps the block alive
    assert int(v.sum()) == 7 * 320


# ------------------------------------------------------------------------------------------------ C-level RCCL exchange
_ALLGATHER_SCRIPT = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
from jubjub_amd import Engine
from jubjub_amd.dist import RcclComm
from oracle import c_oracle as O
from util import rand_points, rand_scalars
torch.cuda.set_device(0)
eng = Engine(0)
comm = RcclComm(0, 1)
eng.set_comm(comm)
for n in (0, 1, 777, 40000):
    s, p = rand_scalars(31 + n, n), rand_points(32 + n, n)
    ds, dp = torch.from_numpy(s).cuda(), torch.from_numpy(p).cuda()
    want = O.msm(s, p).reshape(64)
    assert (eng.msm_allgather(ds, dp, "terms") == want).all(), n
    assert (eng.msm_allgather(ds, dp, "window") == want).all(), n
    assert (eng.msm_allgather(s, p, "terms") == want).all(), n         # host arrays are staged
    j1, j2 = eng.msm_allgather_begin(ds, dp, "terms"), eng.msm_allgather_begin(s, p, "window")     # the same in two halves, two jobs in flight
    assert (eng.msm_finish(j2) == want).all() and (eng.msm_finish(j1) == want).all(), n
eng.set_comm(None)
comm.close()
try:
    eng.msm_allgather(s, p)                                             # no communicator any more
    print("NOT REFUSED")
except Exception:
    print("ALLGATHER OK", flush=True)
eng.close()
"""


def test_msm_allgather_one_rank_rccl():
    """jj_ctx_set_comm + jj_msm_allgather with a real RCCL communicator of one rank (ncclCommInitRank through ctypes on the process's
    librccl): both partitions equal the oracle, host and device inputs; without a communicator the call is refused.  Runs in a process of
    its own: RCCL's teardown at interpreter exit must not be able to take the test session with it."""
    import sys

    r = subprocess.run([sys.executable, "-c", _ALLGATHER_SCRIPT % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=600)
    assert "ALLGATHER OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_msm_rccl_example_one_rank(tmp_path):
    """examples/msm_rccl.cpp (one process per GPU, ncclUniqueId through a file) built against /opt/rocm's librccl and run with one
    rank: its four ways agree and the point equals the oracle's MSM over the same synthetic terms."""
    import torch

    from jubjub_amd import Engine

    lib = os.path.join(ROOT, "jubjub_amd", "lib")
    exe = tmp_path / "msm_rccl"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "msm_rccl.cpp"),
                           "-L", lib, "-ljubjub_hip", "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    n = 50000
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", JJ_ID_FILE=str(tmp_path / "id"))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([str(exe), str(n), "2"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "allgather == partial+ncclAllGather+combine: ok; term partition == window partition: ok" in r.stdout, r.stdout
    assert "jobs in flight: ok" in r.stdout, r.stdout                     # way D: jj_msm_allgather_begin / jj_msm_finish, three in flight
    got = [ln.split("result=")[1].strip() for ln in r.stdout.splitlines() if "result=" in ln][0]
    e = Engine(0)
    SEED = 0x4A55424A5542
    s = e.synth_scalars(n, SEED, 0, device=torch.device("cuda", 0)).cpu().numpy()
    p = e.random_points(n, SEED ^ 0x9E3779B97F4A7C15, 0, subgroup=False, device=torch.device("cuda", 0)).cpu().numpy()
    e.close()
    assert got == bytes(O.msm(s, p).reshape(64).tolist()).hex()


# ------------------------------------------------------------------------------------------------ ADVICE r3 regressions
def test_msm_windows_override_out_of_range_is_refused():
    """msm_windows below 16 would need more than 128 coarse bins per window in the two-pass sort (LDS overrun): jj_ctx_set_option refuses the
    value (JJ_ERR_INVALID, the range in jj_last_error), the context keeps its planner and the MSM is still right."""
    from jubjub_amd import Engine
    from jubjub_amd.engine import JubjubError

    e = Engine(0, options={"msm_small_max": 0})
    for bad in (12, 15, 37, -1):
        with pytest.raises(JubjubError, match="out of range"):
            e.set_option("msm_windows", bad)
    with pytest.raises(JubjubError, match="unknown key"):
        e.set_option("varbase_default", 0)                      # no option reaches the timing discipline of an entry point
    assert e.get_option("msm_windows") == 0 and e.get_option("msm_small_max") == 0
    n = 70000
    s, p = rand_scalars(8, n), rand_points(9, n)
    got = e.msm(s, p)
    e.close()
    assert (got == O.msm(s, p).reshape(64)).all()


def test_msm_window_partition_with_more_parts_than_windows(eng):
    """part_index >= W: an empty but valid record (no zero-sized launches); all the parts still combine to the MSM."""
    n = 300                                       # small-batch path: W = 64 windows
    s, p = rand_scalars(41, n), rand_points(42, n)
    G = 70
    recs = np.stack([eng.msm_partial(s, p, g, G) for g in range(G)])
    assert (eng.msm_combine(recs) == O.msm(s, p).reshape(64)).all()
    hdr = recs[G - 1][:32].view("<u4")
    assert hdr[0] == 0x504D4A4A and hdr[4] == 0 and hdr[5] == 0          # no windows present


def test_msm_finish_without_output_releases_the_job(eng):
    import torch

    n = 20000
    s, p = torch.from_numpy(rand_scalars(51, n)).cuda(), torch.from_numpy(rand_points(52, n)).cuda()
    lib = eng._lib
    for _ in range(4):
        h = C.c_void_p()
        assert lib.jj_msm_begin(eng._ctx, C.c_size_t(n), C.c_void_p(s.data_ptr()), C.c_void_p(p.data_ptr()), C.byref(h)) == 0
        assert lib.jj_msm_finish(h, None) != 0                              # invalid argument; the job is released after its kernels finished
    assert (eng.msm(s, p).cpu().numpy() == O.msm(s.cpu().numpy(), p.cpu().numpy()).reshape(64)).all()


# ------------------------------------------------------------------------------------------------ device-side MSM finish
@pytest.mark.parametrize("n", [0, 1, 2, 300, 20000, 200000])
def test_msm_dev_finish(eng, golden, n):
    """jj_msm_dev: Horner over the record's windows + one inversion on a quad of lanes, result left in device memory -- equal to
    jj_msm (host tail) and the oracle, for the small-batch layout (64 windows) and both Pippenger layouts (23 / 17 windows), with
    8-torsion points, the identity and edge scalars among the terms."""
    import torch

    from util import EDGE_SCALARS, arr32, torsion_points

    s, p = rand_scalars(61 + n, n, full_width=True), rand_points(62 + n, n)
    if n >= 300:
        t = torsion_points(golden)
        p[:8] = t
        s[8:8 + len(EDGE_SCALARS)] = arr32([k % (1 << 256) for k in EDGE_SCALARS])
    ds, dp = torch.from_numpy(s).cuda(), torch.from_numpy(p).cuda()
    got = eng.msm_dev(ds, dp)
    assert got.is_cuda and got.shape == (64,)
    want = O.msm(s, p).reshape(64)
    assert (got.cpu().numpy() == want).all()
    assert (eng.msm(ds, dp).cpu().numpy() == want).all()
    out = torch.zeros(64, dtype=torch.uint8, device="cuda")
    assert eng.msm_dev(s, p, out=out) is out and (out.cpu().numpy() == want).all()       # host inputs are staged, the result still stays on the device
    with pytest.raises(Exception):
        eng._check(eng._lib.jj_msm_dev(eng._ctx, C.c_size_t(n), C.c_void_p(ds.data_ptr()), C.c_void_p(dp.data_ptr()), np.zeros(64, np.uint8).ctypes.data))   # host result pointer


def test_large_pageable_arrays_of_unpipelined_entry_points(eng):
    """entry points outside the chunked pipeline (MSM inputs, batched field and point operations) move pageable arrays of 16 MB and more
    through the page-locked staging slots (host_to_dev_bounced / dev_to_host_bounced): 2^20 + 5 units, several slots' worth each way"""
    n = (1 << 20) + 5
    s = rand_scalars(4242, n)
    small = rand_points(4243, 2048)
    p = np.ascontiguousarray(small[np.arange(n) % 2048])
    assert (eng.msm(s, p) == O.msm_pippenger(s, p, 13).reshape(64)).all()
    got = eng.field_binary("fq", "mul", s, s[::-1].copy())                    # 2 x 33.5 MB in, 33.5 MB out
    idx = np.concatenate([np.arange(0, n, 997), [n - 1, (1 << 19) - 1, 1 << 19]])
    assert (got[idx] == O.field_op(O.FQ, "mul", s[idx], s[::-1][idx])[0]).all()
    dbl = eng.point_double(p)                                                  # 67 MB in, 67 MB out
    assert (dbl[idx] == O.point_op("double", p[idx])).all()


@pytest.mark.parametrize("n", [(1 << 19) + 77, 3 * (1 << 19) + 5])
def test_msm_host_arrays_reduced_in_overlapped_passes(n, monkeypatch):
    """Host arrays of 2^19 terms and more: the sum is taken in two to eight passes whose copies overlap the previous pass's kernels
    (msm_begin_locked).  Same 64 bytes as the device-resident call, as one pass after the whole copy (JJ_MSM_HOST_SPLIT=0), and as the
    CPU Pippenger of the oracle; page-locked and pageable arrays; the async begin/finish pair and the raw partial record too."""
    import torch
    from jubjub_amd import Engine

    s, p = rand_scalars(61, n), rand_points(62, n)
    p[5] = p[4]
    s[7] = 0
    want = O.msm_pippenger(s, p).reshape(64)
    e = Engine(0)
    dev = e.msm(torch.from_numpy(s).cuda(), torch.from_numpy(p).cuda()).cpu().numpy()
    assert (dev == want).all()
    assert (e.msm(s, p) == want).all()                                   # pageable: staged through the page-locked slots
    hs, hp = e.host_alloc((n, 32)), e.host_alloc((n, 64))
    hs[:], hp[:] = s, p
    for _ in range(3):                                                   # the copy-stream events are reused call after call
        assert (e.msm(hs, hp) == want).all()
    rec = e.msm_partial(hs, hp, 0, 1)
    assert (e.msm_combine(rec[None]) == want).all()
    e.close()
    opts = {}
    opts['msm_host_split'] = 0
    e = Engine(0, options=opts)
    assert (e.msm(s, p) == want).all()
    e.close()


def test_msm_host_arrays_more_passes_than_eight(monkeypatch):
    """Host arrays longer than eight full passes (JJ_MSM_PASS_LOG2 = 18 stands in for 2^24 here: 3 * 2^19 + 5 terms are 7 passes of 2^18):
    every pass's copy still runs beside the previous pass's kernels, the records of all passes -- the last one of another window layout --
    meet in one host tail."""
    import torch
    from jubjub_amd import Engine

    opts = {}
    opts['msm_pass_log2'] = 18
    n = 3 * (1 << 19) + 5
    s, p = rand_scalars(71, n), rand_points(72, n)
    want = O.msm_pippenger(s, p).reshape(64)
    e = Engine(0, options=opts)
    assert (e.msm(s, p) == want).all()
    hs, hp = e.host_alloc((n, 32)), e.host_alloc((n, 64))
    hs[:], hp[:] = s, p
    assert (e.msm(hs, hp) == want).all()
    assert (e.msm(torch.from_numpy(s).cuda(), torch.from_numpy(p).cuda()).cpu().numpy() == want).all()      # device-resident: passes of 2^18, no split
    e.close()


def test_result_pool(eng):
    """jj_result_acquire / jj_result_release (the buffers a caller that returns a NEW result per call takes its results in; reference API shape:
    `-> Vec<..>`, /root/reference/src/lib.rs:541-627, 1084-1107): page-locked (the pipeline copies straight into them), several out at a time and all
    different, recycled after release (the smallest free buffer that fits), foreign / double releases refused; results equal the device-resident path."""
    import torch

    n = N_PIPE
    s, p = _inputs(n, 9100)
    want = eng.varbase_mul(torch.from_numpy(s).cuda(), torch.from_numpy(p).cuda()).cpu().numpy()
    a = eng.result_acquire((n, 64))
    b = eng.result_acquire((n, 64))
    assert a.ctypes.data != b.ctypes.data and eng.result_pool_stats()["in_use"] >= 2
    ra = eng.varbase_mul(s, p, out=a)
    rb = eng.varbase_mul(s, p, out=b)                       # the previous result object is still out: a different one is written
    assert ra.ctypes.data == a.ctypes.data and (a == want).all() and (b == want).all()
    addr_a = a.ctypes.data
    eng.result_release(a)
    c = eng.result_acquire((n, 64))                         # recycled: the buffer just released
    assert c.ctypes.data == addr_a
    small = eng.result_acquire((1000, 32))                  # a much smaller request does not take a 50 MB buffer
    assert small.ctypes.data not in (addr_a, b.ctypes.data)
    tab = eng.fixedbase_table(pt64(J.GENERATOR))
    enc = eng.result_acquire((n, 32))
    assert (eng.fixedbase_mul_compressed(tab, s, out=enc) == eng.fixedbase_mul_compressed(tab, torch.from_numpy(s).cuda()).cpu().numpy()).all()
    ok = eng.result_acquire((n,))
    o1, k1 = eng.decompress(np.array(enc), 13, out=(c, ok))
    o2, k2 = eng.decompress(torch.from_numpy(np.array(enc)).cuda(), 13)
    assert (o1 == o2.cpu().numpy()).all() and (k1 == k2.cpu().numpy()).all() and k1.all()
    tab.close()
    for h in (b, c, small, enc, ok):
        eng.result_release(h)
    assert eng.result_pool_stats()["in_use"] == 0
    with pytest.raises(Exception):
        eng.result_release(b)                               # released twice
    with pytest.raises(Exception):
        eng.result_release(np.zeros((64,), np.uint8))      # not a pool buffer
    assert eng.result_acquire((0, 64)).shape == (0, 64)
