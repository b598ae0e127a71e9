#!/usr/bin/env python3
"""gfx950 assembly of the library's translation units (hipcc -save-temps, compiled side by side), concatenated: the input of
tools/kernel_resources.py, tools/instr_mix.py and tests/test_codegen.py.  Honours JJ_CXXFLAGS."""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from jubjub_amd.build import CSRC, UNITS, hipcc  # noqa: E402


def assembly(units=None, extra_flags=()):
    units = list(units or UNITS)

    def one(u):
        with tempfile.TemporaryDirectory() as td:
            subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-save-temps", "-c", "-x", "hip", os.path.join(CSRC, u + ".hip"),
                                   "-I", CSRC, "-o", os.path.join(td, "e.o")] + list(extra_flags) + os.environ.get("JJ_CXXFLAGS", "").split(), cwd=td, stderr=subprocess.DEVNULL)
            return open(os.path.join(td, u + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()

    with ThreadPoolExecutor(max_workers=len(units)) as ex:
        return "\n".join(ex.map(one, units))


if __name__ == "__main__":
    sys.stdout.write(assembly(sys.argv[1:] or None))
