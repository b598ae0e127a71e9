"""
Guards two code-generation properties the measured throughput depends on, straight from hipcc's gfx950
assembly of the shipped source (no GPU needed):
  * the association pins work: the field products of the hot kernels contain (almost) no v_lshl_add_u64 re-joining a
    column carry, i.e. ~187 instead of ~204 instructions per product;
  * the products are selected as single v_mad_i64_i32 / v_mad_u64_u32 instructions: no signed x unsigned 64-bit
    expansions (pairs of v_mad_u64_u32 glued by v_mov_b32), see JJ_OPAQUE_MODE in jj_field.h;
  * no kernel of the library spills to scratch.
"""
import collections
import os
import re
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gfx_asm import assembly  # noqa: E402


@pytest.fixture(scope="module")
def asm():
    return assembly()              # every translation unit of the library, compiled side by side


def kernel_body(asm, needle):
    for k in re.split(r"\n(?=_Z\w+:\s)", asm):
        name = k.split(":", 1)[0]
        if name.startswith("_Z") and needle in name:
            return k.split("s_endpgm")[0]
    raise AssertionError("kernel %s not found" % needle)


def resources(asm, needle):
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        if needle in m.group(1):
            g = lambda key: int(re.search(r"\.amdhsa_%s (\d+)" % key, m.group(2)).group(1))
            return g("next_free_vgpr"), g("private_segment_fixed_size")
    raise AssertionError("kernel %s not found" % needle)


def ladder_block(asm, needle):
    # the basic block with the most multiply-adds is the ladder body; 64-bit adds elsewhere are address arithmetic
    best = None
    for blk in re.split(r"\n(?=\.LBB\d+_\d+:)", kernel_body(asm, needle)):
        ops = collections.Counter(l.split()[0] for l in blk.splitlines() if re.match(r"^\s+[vs]_", l))
        ops["mads"] = ops["v_mad_u64_u32"] + ops["v_mad_i64_i32"]
        if best is None or ops["mads"] > best["mads"]:
            best = ops
    return best


HOT = ["k_varbaseILb0ELb0", "k_fixedbaseILb1", "k_fixedbase_gather", "k_varbase_quadILb0", "k_msm_accumulate_seg", "k_decompressILi32"]


@pytest.mark.parametrize("needle", HOT)
def test_products_are_pinned(asm, needle):
    best = ladder_block(asm, needle)
    mads, merges = best["mads"], best["v_lshl_add_u64"]
    assert mads > 600
    assert merges * 40 < mads, "column carries are re-joined with 64-bit adds again (%d for %d multiply-adds)" % (merges, mads)


@pytest.mark.parametrize("needle", HOT)
def test_products_are_single_multiply_adds(asm, needle):
    best = ladder_block(asm, needle)
    # a product is 153 multiply-adds (a square 117); register moves come from the expansion of mixed-sign products
    assert best["v_mov_b32_e32"] * 6 < best["mads"], "%d v_mov_b32 for %d multiply-adds: products are being expanded" % (best["v_mov_b32_e32"], best["mads"])


def all_kernels(asm):
    return [m.group(1) for m in re.finditer(r"\.amdhsa_kernel (\S+)", asm)]


def test_no_kernel_spills(asm):
    bad = []
    for name in all_kernels(asm):
        vgpr, scratch = resources(asm, name)
        if scratch or vgpr > 256:
            bad.append((name, vgpr, scratch))
    assert not bad, "kernels with scratch / more than 256 VGPRs: %r" % bad
