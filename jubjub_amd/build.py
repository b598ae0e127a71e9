"""Builds libjubjub_hip.so (gfx950) in-tree with hipcc.  No torch involved: the product is a plain C-ABI library.

Four translation units (csrc/jj_pipeline.hip, jj_abi.hip, jj_msm.hip, jj_multi.hip) are compiled side by side into lib/obj/*.o and
linked; a unit is recompiled when one of the files it includes is newer than its object."""
import os
import shutil
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libjubjub_hip.so")
OBJ = os.path.join(HERE, "lib", "obj")
COMMON = ["jj_engine.h", "jj_kernels.h", "jj_curve.h", "jj_field.h", "jj_constants.h", "jj_host_tail.h", "jj_host_tail_ifma.h", os.path.join("..", "..", "include", "jubjub_hip.h")]
UNITS = {"jj_pipeline": [], "jj_abi": [], "jj_msm": ["jj_msm_kernels.h"], "jj_multi": []}
SOURCES = [os.path.join(CSRC, u + ".hip") for u in UNITS]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"] + os.environ.get("JJ_CXXFLAGS", "").split()


def _deps(unit):
    return [os.path.join(CSRC, unit + ".hip")] + [os.path.join(CSRC, f) for f in COMMON + UNITS[unit]]


def _obj(unit):
    return os.path.join(OBJ, unit + ".o")


def _stale(unit):
    o = _obj(unit)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in _deps(unit)) or os.path.getmtime(os.path.abspath(__file__)) > t


def needs_build():
    if not os.path.exists(OUT):
        return True
    return any(_stale(u) or os.path.getmtime(_obj(u)) > os.path.getmtime(OUT) for u in UNITS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    todo = [u for u in UNITS if force or _stale(u)]

    def compile_unit(u):
        t0 = time.time()
        cmd = [cc] + flags() + ["-c", "-o", _obj(u), os.path.join(CSRC, u + ".hip")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return u, time.time() - t0

    with ThreadPoolExecutor(max_workers=max(1, len(todo))) as ex:
        for u, dt in ex.map(compile_unit, todo):
            if verbose:
                print("  %s: %.1f s" % (u, dt), flush=True)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + [_obj(u) for u in UNITS]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


def build_variant(out, extra_flags, units=("jj_abi",)):
    """A/B build for experiments (JJ_LIB_PATH=<out> selects it): `units` recompiled with extra flags, the other objects taken from the regular build."""
    build()
    cc = hipcc()
    objs = []
    for u in UNITS:
        if u in units:
            o = os.path.join(OBJ, "%s.variant.%s.o" % (u, os.path.basename(out)))
            subprocess.check_call([cc] + flags() + list(extra_flags) + ["-c", "-o", o, os.path.join(CSRC, u + ".hip")])
            objs.append(o)
        else:
            objs.append(_obj(u))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
