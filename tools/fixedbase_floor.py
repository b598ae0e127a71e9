#!/usr/bin/env python3
"""Builds the probe libraries tools/fixedbase_floor.sh and tools/fixedbase_select_pmc.sh use (CPU only; they travel to the GPU box with the snapshot):
   python tools/fixedbase_floor.py build"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from jubjub_amd import build as jb  # noqa: E402

# the library with the experiment switches compiled in (JJ_<KEY> presets of every option, JJ_VARBASE_DEFAULT, JJ_FIXEDBASE_SELECT, ...): tools/fixedbase_select_pmc.sh
print(jb.build_variant(os.path.join(ROOT, "experiments", "probe_lib", "libjj_experiments.so"), ["-DJJ_EXPERIMENTS"], units=("jj_pipeline", "jj_abi", "jj_msm", "jj_multi")))
for probe in (1, 2):
    out = os.path.join(ROOT, "experiments", "probe_lib", "libjj_fbc_probe%d.so" % probe)
    print(jb.build_variant(out, ["-DJJ_EXPERIMENTS", "-DJJ_FBC_PROBE=%d" % probe]))
