#!/usr/bin/env python3
"""The headline bench lines of a round on every box they were taken on: profiles/<tag>_bench_<workload>.json (the evidence run; the fixed-base
line of that run is <tag>_fixedbase_bench.json) and the <tag>_bench_<workload>_box<k>.json samples -- value, roofline fraction, ms per pass, build.
  python tools/box_samples.py [tag]"""
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
for wl in ("default", "fixedbase", "msm20", "msm22", "msm17"):
    rows = []
    ev = os.path.join(prof, "%s_bench_%s.json" % (tag, wl))
    if wl == "fixedbase":
        ev = os.path.join(prof, "%s_fixedbase_bench.json" % tag)
    files = [("evidence", ev)] + sorted((re.search(r"_(box\d+)\.json$", f).group(1), f) for f in glob.glob(os.path.join(prof, "%s_bench_%s_box*.json" % (tag, wl))))
    for name, f in files:
        if not os.path.exists(f):
            continue
        d = json.loads(open(f).read().strip().splitlines()[-1])
        rows.append((name, d["value"] / 1e6, d["roofline"]["frac"], d["config"]["ms_per_pass"], d["roofline"].get("build_id", "")[:8]))
    print("%s:" % wl)
    for r in rows:
        print("  %-9s %8.1f M/s  frac %.3f  %8.3f ms/pass  build %s" % r)
    if rows:
        print("  range     %.1f-%.1f M/s  frac %.3f-%.3f  %.3f-%.3f ms" % (min(r[1] for r in rows), max(r[1] for r in rows), min(r[2] for r in rows), max(r[2] for r in rows),
                                                                         min(r[3] for r in rows), max(r[3] for r in rows)))
