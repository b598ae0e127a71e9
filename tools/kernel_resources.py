#!/usr/bin/env python3
"""Registers and scratch per kernel, from hipcc's gfx950 assembly of the library's translation units (tools/gfx_asm.py).
Usage: python tools/kernel_resources.py   (honours JJ_CXXFLAGS)"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gfx_asm import assembly  # noqa: E402


def main():
    asm = assembly()
    print("%-58s %5s %5s %7s" % ("kernel", "vgpr", "sgpr", "scratch"))
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        name, body = m.group(1), m.group(2)
        g = lambda k: re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1)
        short = re.sub(r"^_ZN2jj\d+", "", name)[:56]
        print("%-58s %5s %5s %7s" % (short, g("next_free_vgpr"), g("next_free_sgpr"), g("private_segment_fixed_size")))


if __name__ == "__main__":
    main()
