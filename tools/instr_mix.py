#!/usr/bin/env python3
"""Static instruction mix of the shipped kernels, straight from hipcc's gfx950 assembly (-save-temps): how many
v_mad_u64_u32 / other VALU / SALU / memory instructions each kernel (and each of its loops) contains.
This is where the per-field-op numbers in DESIGN.md (162 multiply-adds + ~61 other VALU per multiplication) come from.
Usage: python tools/instr_mix.py [kernel-substring ...]      (default: k_varbase k_fixedbase k_field_op)"""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gfx_asm import assembly  # noqa: E402


def main():
    want = sys.argv[1:] or ["k_varbase_ct3", "k_varbase_ct_quad", "k_varbaseILb0ELb0", "k_fixedbase_combILb1", "k_fixedbaseILb1", "k_field_opINS_3FqPELi2", "k_field_opINS_3FqPELi4"]
    asm = assembly()
    kernels = re.split(r"\n(?=_Z\w+:\s)", asm)
    for k in kernels:
        name = k.split(":", 1)[0]
        if not name.startswith("_Z") or not any(w in name for w in want):
            continue
        body = k.split("s_endpgm")[0]
        print("==", name)
        blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
        total = collections.Counter()
        for b in blocks:
            label = b.split(":", 1)[0] if b.startswith(".LBB") else "entry"
            ops = [l.split()[0] for l in b.splitlines() if re.match(r"^\s+[vsdgb][a-z0-9_]*_", l)]
            total.update(ops)
            if len(ops) >= 400:
                c = collections.Counter(ops)
                mad = c["v_mad_u64_u32"]
                valu = sum(v for o, v in c.items() if o.startswith("v_"))
                print("   block %-10s %5d instr: %5d v_mad_u64_u32, %5d other VALU, %4d SALU, %4d memory/LDS" % (
                    label, len(ops), mad, valu - mad, sum(v for o, v in c.items() if o.startswith("s_")),
                    sum(v for o, v in c.items() if o.startswith(("global_", "ds_", "buffer_", "scratch_", "flat_")))))
        if "k_varbase_ct" in name:
            # the constant-time claim, checkable: every memory and control-flow instruction of the ladder's loop (the block with the most multiply-adds)
            loops = [b for b in blocks if b.startswith(".LBB") and re.search(r"s_cbranch_\w+\s+" + re.escape(b.split(":", 1)[0]) + r"\b", b)]   # blocks that branch back to their own label
            loop = max(loops or blocks, key=lambda b: len(re.findall(r"v_mad_i64_i32", b)))
            loop = loop[:max(m.end() for m in re.finditer(r"s_cbranch_\w+\s+" + re.escape(loop.split(":", 1)[0]) + r"\b", loop))] if loops else loop      # up to the back edge
            mc = collections.Counter(m.group(1) for m in re.finditer(r"^\s+((?:global|ds|buffer|scratch|flat|s_load|s_cbranch|s_branch|v_cmp|v_cndmask)[a-z0-9_]*)\s", loop, re.M))
            print("   loop block: memory / control instructions: %s" % ", ".join("%s x%d" % kv for kv in sorted(mc.items())))
            print("   (ct3: the two global loads are the words of k' that hold window i -- address = f(unit, i) --, the ds_read_b128 the lane's own LDS slot, read whole;"
                  " ct_quad: no memory instruction at all.  The compare / select pairs build an all-ones mask or clamp the word index: data flow only; the one branch is the loop counter's)")
        mad = total["v_mad_u64_u32"]
        valu = sum(v for o, v in total.items() if o.startswith("v_"))
        print("   TOTAL %d instr: %d v_mad_u64_u32 (%.0f%% of VALU), %d other VALU; top other ops: %s" % (
            sum(total.values()), mad, 100.0 * mad / max(valu, 1), valu - mad,
            ", ".join("%s x%d" % (o, v) for o, v in total.most_common(8) if o != "v_mad_u64_u32")))


if __name__ == "__main__":
    main()
