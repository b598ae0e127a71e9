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


def ladder_loop(asm, needle):
    """the basic block of the kernel that branches back to its own label and holds the most multiply-adds, up to its back edge"""
    blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", kernel_body(asm, needle))
    loops = [b for b in blocks if b.startswith(".LBB") and re.search(r"s_cbranch_\w+\s+" + re.escape(b.split(":", 1)[0]) + r"\b", b)]
    assert loops, "no loop found in %s" % needle
    loop = max(loops, key=lambda b: len(re.findall(r"v_mad_i64_i32", b)))
    return loop[:max(m.end() for m in re.finditer(r"s_cbranch_\w+\s+" + re.escape(loop.split(":", 1)[0]) + r"\b", loop))]


def test_constant_time_ladders_have_no_scalar_dependent_memory_access_or_branch(asm):
    """The default variable-base ladder (jj_varbase_mul, reference discipline: conditional_select, /root/reference/src/lib.rs:334-343, 357-379): the loop
    of k_varbase_ct3 holds exactly the eighteen sixteen-byte reads of the lane's own LDS slot (both entries, every window) and the two words of the parked
    k' (address = unit and window index), no store, no shuffle, no scratch, and ONE branch (the window counter's); the loop of k_varbase_ct_quad holds no
    memory instruction at all.  A table lookup at a digit-dependent address or a branch on a digit would show up here as another load / branch."""
    mem = r"^\s+((?:global|ds|buffer|scratch|flat|s_load)[a-z0-9_]*)\s"
    ct3 = ladder_loop(asm, "k_varbase_ct3")
    ops = collections.Counter(m.group(1) for m in re.finditer(mem, ct3, re.M))
    assert ops == {"ds_read_b128": 18, "global_load_dword": 2}, ops
    assert len(re.findall(r"^\s+s_cbranch", ct3, re.M)) == 1 and not re.search(r"^\s+(s_setpc|s_swappc|s_call)", ct3, re.M)
    quad = ladder_loop(asm, "k_varbase_ct_quad")
    assert not re.search(mem, quad, re.M), "k_varbase_ct_quad touches memory inside its loop"
    assert len(re.findall(r"^\s+s_cbranch", quad, re.M)) == 1
    # for contrast: the variable-time table ladder does load inside its window loop (its per-lane table, at a digit-dependent address)
    vt = kernel_body(asm, "k_varbaseILb0ELb0")
    assert len(re.findall(r"^\s+global_load_dwordx4", vt, re.M)) >= 9
