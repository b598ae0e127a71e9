#!/usr/bin/env python3
"""Compiles probe.hip for gfx950 and counts the instructions of the two kernels (VALU / LDS-pipe / memory)."""
import os, re, subprocess, tempfile
from collections import Counter
here = os.path.dirname(os.path.abspath(__file__))
with tempfile.TemporaryDirectory() as td:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-save-temps", "-c", os.path.join(here, "probe.hip"), "-o", os.path.join(td, "p.o")], cwd=td, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(td, "probe-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
starts = {m.group(1): i for i, l in enumerate(asm) for m in [re.match(r"^(_Z\S+):", l)] if m}
res = {}
for key in ("k_valu_reduction", "k_mfma_overheads"):
    name = [k for k in starts if key in k][0]
    i = starts[name]; j = i
    while "s_endpgm" not in asm[j]: j += 1
    ins = [l.split()[0] for l in (x.strip() for x in asm[i + 1:j]) if l and not l.startswith((".", ";")) and not l.split()[0].endswith(":")]
    c = Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    res[key] = (valu, c["v_mad_i64_i32"] + c["v_mad_u64_u32"], c["ds_bpermute_b32"], sum(v for k, v in c.items() if k.startswith("global_")))
    print("%-18s VALU %4d (multiply-adds %3d)  ds_bpermute %3d  global memory %3d" % ((key,) + res[key]))
a, b = res["k_valu_reduction"][0] - 17 - 9, res["k_mfma_overheads"][0] - 18 - 9     # minus the address arithmetic / moves of the probe's loads and stores (one v_ per global access)
print("reduction half on the VALU as shipped: ~%d VALU instructions per product" % a)
print("VALU work around the matrix instructions (MFMA themselves free): ~%d VALU + %d ds_bpermute per product" % (b, res["k_mfma_overheads"][2]))
print("%s: the MFMA route needs %.2fx the VALU instructions of the half it replaces (whole product: %d vs 187), before counting its %d LDS-pipe shuffles"
      % ("FAIL" if b >= a / 1.15 else "PASS", b / a, 187 - a + b, res["k_mfma_overheads"][2]))
