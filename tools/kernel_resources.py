#!/usr/bin/env python3
"""Registers and scratch per kernel, from hipcc's gfx950 assembly of jj_engine.hip (-save-temps).
Usage: python tools/kernel_resources.py   (honours JJ_CXXFLAGS)"""
import os
import re
import subprocess
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SRC = os.path.join(ROOT, "jubjub_amd", "csrc", "jj_engine.hip")


def main():
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-save-temps", "-c", "-x", "hip", SRC,
                               "-I", os.path.dirname(SRC), "-o", os.path.join(td, "e.o")] + os.environ.get("JJ_CXXFLAGS", "").split(),
                              cwd=td, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(td, "jj_engine-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    print("%-58s %5s %5s %7s" % ("kernel", "vgpr", "sgpr", "scratch"))
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        name, body = m.group(1), m.group(2)
        g = lambda k: re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1)
        short = re.sub(r"^_ZN2jj\d+", "", name)[:56]
        print("%-58s %5s %5s %7s" % (short, g("next_free_vgpr"), g("next_free_sgpr"), g("private_segment_fixed_size")))


if __name__ == "__main__":
    main()
