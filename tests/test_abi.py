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
    buf = ctypes.create_string_buffer(1 << 16)
    assert lib.jj_host_register(buf, 1 << 16) == _lib.JJ_ERR_NODEVICE


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
