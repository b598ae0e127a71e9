"""Builds libjubjub_hip.so (gfx950) in-tree with hipcc.  No torch involved: the product is a plain C-ABI library."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "jj_engine.hip")
OUT = os.path.join(HERE, "lib", "libjubjub_hip.so")
DEPS = [os.path.join(HERE, "csrc", f) for f in ("jj_engine.hip", "jj_kernels.h", "jj_curve.h", "jj_field.h", "jj_constants.h", "jj_host_tail.h", "jj_host_tail_ifma.h", "jj_msm_kernels.h")] + [
    os.path.join(HERE, "..", "include", "jubjub_hip.h")]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden",
           "-Wall", "-Wno-unused-function", "-o", OUT, SRC] + os.environ.get("JJ_CXXFLAGS", "").split()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
