import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


_GPU_STATE = None


def gpu_visible():
    """A GPU is present on this box when the kernel driver node exists or torch sees a device (no library of ours involved)."""
    if os.path.exists("/dev/kfd"):
        return True
    try:
        import torch

        return bool(torch.cuda.is_available())
    except Exception:
        return False


def gpu_state():
    """'ok'      the C-ABI library opened a gfx950 device;
    'absent'  no GPU on this box (no /dev/kfd, torch sees none): GPU tests are skipped -- the product has no CPU fallback;
    'broken'  a GPU is visible but jj_ctx_create failed (bad build, runtime mismatch, wrong arch): GPU tests FAIL, they are
              never skipped -- a dead library must not turn into a green run."""
    global _GPU_STATE
    if _GPU_STATE is None:
        if not gpu_visible():
            _GPU_STATE = ("absent", "no GPU on this box (no /dev/kfd, torch.cuda.is_available() is false)")
        else:
            try:
                import ctypes

                from jubjub_amd import _lib

                lib = _lib.load()
                ctx = ctypes.c_void_p()
                rc = lib.jj_ctx_create(0, ctypes.byref(ctx))
                if rc == 0:
                    lib.jj_ctx_destroy(ctx)
                    _GPU_STATE = ("ok", "")
                else:
                    _GPU_STATE = ("broken", "a GPU is visible but jj_ctx_create(0) returned %d" % rc)
            except Exception as e:  # library missing / unloadable on a GPU box
                _GPU_STATE = ("broken", "a GPU is visible but libjubjub_hip.so could not be used: %r" % (e,))
    return _GPU_STATE


def have_gpu():
    return gpu_state()[0] == "ok"


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box WITHOUT a GPU skips the GPU tests; on a box WITH a GPU they run, and if the library
    cannot open the device every one of them fails."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items:
        return
    state, why = gpu_state()
    if state == "absent":
        skip = pytest.mark.skip(reason=why + "; the product has no CPU fallback")
        for it in gpu_items:
            it.add_marker(skip)


def pytest_runtest_setup(item):
    if item.get_closest_marker("gpu") and gpu_state()[0] == "broken":
        pytest.fail("GPU present but unusable: " + gpu_state()[1], pytrace=False)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "reference_vectors.json")) as f:
        return json.load(f)


def limbs(hexlist):
    """['0x..', ...] little-endian u64 limbs -> int."""
    return sum(int(h, 16) << (64 * i) for i, h in enumerate(hexlist))
