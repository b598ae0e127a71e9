"""CPU-side checks of the C-ABI boundary: the library loads without a GPU, exports every symbol that
include/jubjub_hip.h declares, refuses to create a context without a gfx950 device (no CPU fallback), and the
product package never touches the oracle."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "jubjub_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jj_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge

    ge.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "jubjub_amd", "lib", "libjubjub_hip.so"))
    names = declared_symbols()
    assert len(names) >= 50
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    from jubjub_amd import _lib

    for n in _lib.EXPORTS:
        assert n in names, "python binding uses an undeclared symbol: " + n
    assert lib.jj_version() >= 100


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from jubjub_amd import Engine, JubjubError

    with pytest.raises(JubjubError):
        Engine(0)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "jubjub_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|#include\s+[\"<].*oracle|libjj_oracle", src, re.M), os.path.join(dirpath, f)
    for f in ("include/jubjub_hip.h", "include/jubjub_hip.hpp"):
        assert "oracle" not in open(os.path.join(ROOT, f)).read()
    # tools/ and examples/ are product-side helpers too: only tests/, smoke() and bench.py's cpu_baseline use the oracle
    for sub in ("tools", "examples"):
        for f in os.listdir(os.path.join(ROOT, sub)):
            if f.endswith((".py", ".c", ".cpp", ".h")):
                src = open(os.path.join(ROOT, sub, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|#include\s+[\"<].*oracle|libjj_oracle", src, re.M), os.path.join(sub, f)


def test_wnaf_recommendation_matches_reference(golden):
    """reference src/lib.rs:1320-1335 through the C ABI (host-only function, no GPU needed)."""
    from jubjub_amd import _lib

    lib = _lib.load()
    tab = golden["wnaf_recommendations"]["table"]
    assert lib.jj_recommended_wnaf_for_num_scalars(0) == 4
    for i, r in enumerate(tab):
        assert lib.jj_recommended_wnaf_for_num_scalars(r) == 4 + i
        assert lib.jj_recommended_wnaf_for_num_scalars(r + 1) == 5 + i


def test_c_example_compiles(tmp_path):
    """The plain-C caller in examples/ compiles against the header and links the library (C, not C++: checks the ABI is C)."""
    import subprocess

    out = tmp_path / "scalar_mul"
    lib = os.path.join(ROOT, "jubjub_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "scalar_mul.c"), "-L", lib, "-ljubjub_hip",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-o", str(out)])
    assert out.exists()


def test_rccl_example_compiles(tmp_path):
    """examples/msm_rccl.cpp (the C-level multi-rank MSM exchange) compiles against the header, RCCL and the library."""
    import subprocess

    out = tmp_path / "msm_rccl"
    lib = os.path.join(ROOT, "jubjub_amd", "lib")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "msm_rccl.cpp"),
                           "-L", lib, "-ljubjub_hip", "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", str(out)])
    assert out.exists()


def test_host_buffer_api_without_a_gpu():
    """jj_host_alloc / jj_host_register need a device to page-lock for: without one they fail cleanly (no CPU fallback, no crash);
    jj_host_free(NULL) is a no-op everywhere"""
    import torch

    from jubjub_amd import _lib

    lib = _lib.load()
    assert lib.jj_host_free(None) == 0
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_host_path.py")
    p = ctypes.c_void_p()
    assert lib.jj_host_alloc(1 << 20, ctypes.byref(p)) == _lib.JJ_ERR_NODEVICE and not p.value
    import mmap

    m = mmap.mmap(-1, 1 << 16)                                            # a page-aligned mapping of its own: what jj_host_register takes
    buf = (ctypes.c_char * (1 << 16)).from_buffer(m)
    assert lib.jj_host_register(buf, 1 << 16) == _lib.JJ_ERR_NODEVICE
    assert lib.jj_host_register(ctypes.c_void_p(ctypes.addressof(buf) + 64), 1 << 12) == _lib.JJ_ERR_INVALID      # not page-aligned: refused before anything else
    # a page-aligned START is not enough: 5000 bytes from a 4096-aligned address share their last page with whatever follows
    assert lib.jj_host_register(buf, 5000) == _lib.JJ_ERR_INVALID
    libc = ctypes.CDLL(None)
    libc.aligned_alloc.restype = ctypes.c_void_p
    libc.aligned_alloc.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
    pa = libc.aligned_alloc(4096, 8192)
    assert lib.jj_host_register(ctypes.c_void_p(pa), 5000) == _lib.JJ_ERR_INVALID
    assert lib.jj_host_register(ctypes.c_void_p(pa), 8192) == _lib.JJ_ERR_NODEVICE       # whole pages: accepted as far as the (absent) device
    libc.free.argtypes = [ctypes.c_void_p]
    libc.free(pa)
    del buf


def test_job_entry_points_refuse_null_arguments():
    """the two-halves MSM entry points (jj_msm_begin, jj_msm_allgather_begin, jj_msm_finish) check their arguments before touching a device:
    no context, no job slot, a partition that is neither terms nor windows -> JJ_ERR_INVALID (no crash, no CPU fallback)"""
    from jubjub_amd import _lib

    lib = _lib.load()
    job = ctypes.c_void_p()
    out = (ctypes.c_uint8 * 64)()
    assert lib.jj_msm_begin(None, 0, None, None, ctypes.byref(job)) == _lib.JJ_ERR_INVALID and not job.value
    assert lib.jj_msm_allgather_begin(None, 0, None, None, 0, ctypes.byref(job)) == _lib.JJ_ERR_INVALID and not job.value
    assert lib.jj_msm_finish(None, out) == _lib.JJ_ERR_INVALID
    assert lib.jj_msm_allgather(None, 0, None, None, 0, out) == _lib.JJ_ERR_INVALID
    assert lib.jj_msm_combine_dev(None, 0, None, out) == _lib.JJ_ERR_INVALID


def test_host_batch_chunk_schedule_properties():
    """jj_plan_host_chunks is the function run_pipelined cuts a host batch with (jj_pipeline.hip pipe_chunk_bounds): the chunks tile [0, n)
    in order, none is empty or longer than chunk + the edge, the ramp's short first and last chunk appear exactly when the batch has four
    chunks of at least 2^18 units, and with a quantum the edges are whole rounds of the kernel's lanes."""
    import random

    from jubjub_amd import _lib

    lib = _lib.load()

    def plan(n, ch, q=0, ramp=1):
        cnt = ctypes.c_size_t()
        assert lib.jj_plan_host_chunks(n, ch, q, ramp, None, 0, ctypes.byref(cnt)) == 0
        b = (ctypes.c_size_t * cnt.value)()
        assert lib.jj_plan_host_chunks(n, ch, q, ramp, b, cnt.value, ctypes.byref(cnt)) == 0
        return list(b)

    rnd = random.Random(7)
    cases = [(1, 1 << 20, 0, 1), (1 << 24, 1 << 20, 0, 1), (1 << 24, 1 << 20, 0, 0), (1 << 24, 1 << 20, 196608, 1), ((1 << 22) - 1, 1 << 20, 0, 1),
             (4 << 20, 1 << 20, 0, 1), ((4 << 20) - 1, 1 << 20, 0, 1), (1 << 20, 1 << 16, 0, 1), (5 << 18, 1 << 18, 0, 1), ((1 << 23) + 12345, 1 << 21, 0, 1)]
    cases += [(rnd.randrange(1, 1 << 25), 1 << rnd.randrange(14, 22), rnd.choice([0, 0, 49152, 196608]), rnd.randrange(2)) for _ in range(300)]
    for n, ch, q, ramp in cases:
        b = plan(n, ch, q, ramp)
        assert b[0] == 0 and b[-1] == n and all(x < y for x, y in zip(b, b[1:])), (n, ch, q, ramp, b[:4])
        ramped = bool(ramp) and n >= 4 * ch and ch >= (1 << 18)
        edge = ch // 4 if ramped else 0
        if ramped and q and ch % q == 0:
            edge = max(q, edge // q * q)
        sizes = [y - x for x, y in zip(b, b[1:])]
        assert max(sizes) <= ch + edge, (n, ch, q, ramp, max(sizes))
        if ramped:
            assert sizes[0] == edge and sizes[-1] <= max(edge, ch + edge) and len(sizes) >= 4
            assert all(sz == ch for sz in sizes[1:-2]), (n, ch, sizes[:3], sizes[-3:])       # whole chunks between the edges (the one before the last may be longer)
        else:
            assert all(sz == ch for sz in sizes[:-1]) or len(sizes) == 1
        assert len(sizes) <= n // ch + 3
    cnt = ctypes.c_size_t()
    assert lib.jj_plan_host_chunks(0, 1 << 20, 0, 1, None, 0, ctypes.byref(cnt)) == _lib.JJ_ERR_INVALID
    assert lib.jj_plan_host_chunks(100, 0, 0, 1, None, 0, ctypes.byref(cnt)) == _lib.JJ_ERR_INVALID
    small = (ctypes.c_size_t * 2)()
    assert lib.jj_plan_host_chunks(1 << 24, 1 << 20, 0, 1, small, 2, ctypes.byref(cnt)) == _lib.JJ_ERR_INVALID and cnt.value > 2     # too small a buffer: the count is still reported


def test_msm_host_pass_plan_properties():
    """jj_plan_msm_host_passes is the function msm_begin_locked cuts host arrays with: below 2^19 terms (or with the split off) one pass per
    2^pass_log2 terms; from 2^19 terms two to eight passes of at least 2^18 terms, a multiple of 64 each, that cover n (more passes of
    2^pass_log2 terms when eight would be longer than that)."""
    from jubjub_amd import _lib

    lib = _lib.load()

    def plan(n, lg=24, split=1):
        pt, ps = ctypes.c_size_t(), ctypes.c_size_t()
        assert lib.jj_plan_msm_host_passes(n, lg, split, ctypes.byref(pt), ctypes.byref(ps)) == 0
        return pt.value, ps.value

    assert plan(0) == (1 << 24, 0) and plan(1) == (1 << 24, 1) and plan((1 << 19) - 1) == (1 << 24, 1)
    assert plan(1 << 19) == (1 << 18, 2) and plan(1 << 20) == (1 << 19, 2) and plan(1 << 22) == (1 << 19, 8) and plan(1 << 24) == (1 << 21, 8)
    assert plan(1 << 22, split=0) == (1 << 24, 1) and plan((1 << 24) + 1, split=0) == (1 << 24, 2) and plan(1 << 25) == (1 << 22, 8)
    assert plan(1 << 28) == (1 << 24, 16)                             # eight passes would exceed 2^24 terms each: more passes of 2^24
    assert plan(1 << 20, lg=19) == (1 << 19, 2) and plan(3 << 20, lg=18) == (1 << 18, 12)
    for n in [(1 << 19) + 1, (1 << 19) + 77, 3 * (1 << 19) + 5, (1 << 21) - 1, (1 << 23) + 4097, (1 << 24) - 63, (1 << 24) + 1, (1 << 26) + 12345]:
        pt, ps = plan(n)
        assert pt % 64 == 0 and pt >= (1 << 18) and 2 <= ps <= 8 and pt * ps >= n > pt * (ps - 1), (n, pt, ps)
    pt, ps = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.jj_plan_msm_host_passes(1, 9, 1, ctypes.byref(pt), ctypes.byref(ps)) == _lib.JJ_ERR_INVALID


def test_ifma_host_tail_unit(tmp_path):
    """tests/cpp/test_host_tail_ifma.cpp: the four-lane AVX-512 IFMA field product, point doubling / addition and Horner chain of
    jj_host_tail_ifma.h against the scalar host tail, lane by lane and with the ranges of long chains (skipped on CPUs without avx512ifma)."""
    import subprocess

    flags = open("/proc/cpuinfo").read()
    if "avx512ifma" not in flags or "avx512vl" not in flags:
        pytest.skip("no AVX-512 IFMA on this CPU (the library then takes the scalar chain)")
    exe = tmp_path / "test_host_tail_ifma"
    src = os.path.join(ROOT, "tests", "cpp", "test_host_tail_ifma.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-mavx512f", "-mavx512vl", "-mavx512ifma", "-o", str(exe), src])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "IFMA HOST TAIL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    for marker in ("field mul ok", "point ops ok", "horner ok"):
        assert marker in r.stdout


def test_msm_fold_partials_host_only():
    """jj_msm_fold_partials (the last step of an MSM cut across devices / ranks) is a host-only function: partial points incl. the
    identity, 8-torsion points and P, -P pairs against the oracle's fold (reference `Sum`, src/lib.rs:183-193)."""
    import numpy as np

    from jubjub_amd import _lib
    from oracle import c_oracle as O
    from oracle import jubjub_ref as J
    from util import pt64, rand_points

    lib = _lib.load()
    g8 = J.scalar_mul_fast(J.GENERATOR, J.R_MOD)                      # order-8 component of the generator
    special = np.stack([pt64(J.AFFINE_IDENTITY), pt64(g8), pt64(J.scalar_mul_fast(g8, 4)), pt64(J.GENERATOR), pt64(J.affine_neg(J.GENERATOR))])
    for count in (0, 1, 2, 8, 64):
        parts = np.concatenate([rand_points(900 + count, count), special])[: max(count, 0) + (5 if count else 0)]
        out = np.empty(64, np.uint8)
        assert lib.jj_msm_fold_partials(ctypes.c_size_t(len(parts)), parts.ctypes.data if len(parts) else None, out.ctypes.data) == 0
        want = pt64(J.AFFINE_IDENTITY)
        for p in parts:
            want = O.point_op("add", want[None, :], p[None, :])[0]
        assert (out == want).all(), count
    assert lib.jj_msm_fold_partials(ctypes.c_size_t(1), None, None) != 0


def test_gpu_tests_fail_not_skip_on_a_broken_gpu_box(monkeypatch):
    """tests/conftest.py: without a GPU the -m gpu tests are skipped; with a GPU visible but the library unable to open it (bad
    build, runtime mismatch) they must FAIL -- a dead library must not turn into a green run"""
    import conftest
    from jubjub_amd import _lib

    monkeypatch.setattr(conftest, "_GPU_STATE", None)
    monkeypatch.setattr(conftest, "gpu_visible", lambda: False)
    assert conftest.gpu_state()[0] == "absent"
    monkeypatch.setattr(conftest, "_GPU_STATE", None)
    monkeypatch.setattr(conftest, "gpu_visible", lambda: True)

    class DeadLib:
        def jj_ctx_create(self, dev, ref):
            return -2

    monkeypatch.setattr(_lib, "load", lambda: DeadLib())
    state, why = conftest.gpu_state()
    assert state == "broken" and "-2" in why

    class Item:
        def get_closest_marker(self, name):
            return object() if name == "gpu" else None

    with pytest.raises(pytest.fail.Exception):
        conftest.pytest_runtest_setup(Item())
    monkeypatch.setattr(conftest, "_GPU_STATE", None)

    def boom():
        raise RuntimeError("libjubjub_hip.so is not built")

    monkeypatch.setattr(_lib, "load", boom)
    assert conftest.gpu_state()[0] == "broken"
    monkeypatch.setattr(conftest, "_GPU_STATE", None)              # leave no cached verdict behind for the other tests


def test_shipped_library_reads_no_jj_environment_variable():
    """Round 5's library read 34 JJ_* variables, two of which turned constant-time entry points into variable-time ones.  The shipped library
    now holds no such name at all (options go through jj_ctx_set_option; experiment switches exist in -DJJ_EXPERIMENTS probe builds only)."""
    import re
    import subprocess

    from jubjub_amd import _lib

    blob = open(_lib.LIB_PATH, "rb").read()
    names = sorted(set(m.decode() for m in re.findall(rb"JJ_[A-Z0-9_]{3,}", blob)) - {"JJ_MSM_PARTIAL_BYTES"})      # (a constant's name inside HIPCHK messages)
    assert names == [], names
    src = "".join(open(os.path.join(ROOT, "jubjub_amd", "csrc", f)).read() for f in sorted(os.listdir(os.path.join(ROOT, "jubjub_amd", "csrc"))))
    outside = re.sub(r"#ifdef JJ_EXPERIMENTS.*?#e(?:lse|ndif)", "", src, flags=re.S)
    assert re.findall(r'getenv\("(\w+)"\)', outside) == ["GPU_MAX_HW_QUEUES"]        # a HIP runtime variable, read for a one-time warning only
