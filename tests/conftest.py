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


_HAVE_GPU = None


def have_gpu():
    """True when the C-ABI library can open a gfx950 device (JJ_ERR_NODEVICE = -4 otherwise: there is no CPU fallback)."""
    global _HAVE_GPU
    if _HAVE_GPU is None:
        try:
            import ctypes

            from jubjub_amd import _lib

            lib = _lib.load()
            ctx = ctypes.c_void_p()
            rc = lib.jj_ctx_create(0, ctypes.byref(ctx))
            if rc == 0:
                lib.jj_ctx_destroy(ctx)
            _HAVE_GPU = rc == 0
        except Exception:
            _HAVE_GPU = False
    return _HAVE_GPU


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without an MI355X skips the GPU tests instead of failing them."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items or have_gpu():
        return
    skip = pytest.mark.skip(reason="no gfx950 device (jj_ctx_create -> JJ_ERR_NODEVICE); the product has no CPU fallback")
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "reference_vectors.json")) as f:
        return json.load(f)


def limbs(hexlist):
    """['0x..', ...] little-endian u64 limbs -> int."""
    return sum(int(h, 16) << (64 * i) for i, h in enumerate(hexlist))
