"""
Guards two code-generation properties the measured throughput depends on, straight from hipcc's gfx950
assembly of the shipped source (no GPU needed):
  * the association pins work: the field products of the hot kernels contain (almost) no v_lshl_add_u64 re-joining a
    column carry, i.e. ~205 instead of ~223 instructions per product;
  * the hot kernels do not spill to scratch.
"""
import collections
import os
import re
import subprocess

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SRC = os.path.join(ROOT, "jubjub_amd", "csrc", "jj_engine.hip")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    td = str(tmp_path_factory.mktemp("codegen"))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-save-temps", "-c", "-x", "hip", SRC,
                           "-I", os.path.dirname(SRC), "-o", os.path.join(td, "e.o")], cwd=td, stderr=subprocess.DEVNULL)
    return open(os.path.join(td, "jj_engine-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


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


@pytest.mark.parametrize("needle", ["k_varbaseEm", "k_fixedbaseILb1", "k_fixedbase_gather", "k_varbase_quadILb0"])
def test_products_are_pinned(asm, needle):
    # the basic block with the most multiply-adds is the ladder body; 64-bit adds elsewhere are address arithmetic
    best = None
    for blk in re.split(r"\n(?=\.LBB\d+_\d+:)", kernel_body(asm, needle)):
        ops = collections.Counter(l.split()[0] for l in blk.splitlines() if re.match(r"^\s+[vs]_", l))
        if best is None or ops["v_mad_u64_u32"] > best["v_mad_u64_u32"]:
            best = ops
    mads, merges = best["v_mad_u64_u32"], best["v_lshl_add_u64"]
    assert mads > 600
    assert merges * 40 < mads, "column carries are re-joined with 64-bit adds again (%d for %d multiply-adds)" % (merges, mads)


@pytest.mark.parametrize("needle", ["k_varbaseEm", "k_fixedbaseILb1", "k_fixedbase_gather", "k_msm_accumulateEm", "k_msm_accumulate_seg",
                                    "k_msm_bucket_reduce", "k_varbase_quadILb0"])
def test_hot_kernels_do_not_spill(asm, needle):
    vgpr, scratch = resources(asm, needle)
    assert scratch == 0, "%s spills %d bytes per lane" % (needle, scratch)
    assert vgpr <= 256
