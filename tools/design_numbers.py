#!/usr/bin/env python3
"""Prints the numbers DESIGN.md §6 quotes, straight from the committed bench lines and PMC summaries in profiles/
(python tools/design_numbers.py [tag]).  Nothing is typed by hand in that table."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"


def line(name):
    p = os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, name))
    return json.load(open(p)) if os.path.exists(p) else None


def pmc(wl):
    p = os.path.join(ROOT, "profiles", "%s_%s_pmc.txt" % (tag, wl))
    if not os.path.exists(p):
        return {}
    t = open(p).read()
    g = lambda pat: (re.search(pat, t) or [None, "?"])[1]
    return {"instr": g(r"VALU lane-instructions per unit\s+=\s+(\d+)"), "interval": g(r"one every ([\d.]+) cycles"), "ghz": g(r"=> ([\d.]+) GHz"),
            "bytes": g(r"=>\s+(\d+) B per unit")}


for name in ("varbase_bench", "fixedbase_bench", "msm_bench", "decompress_bench", "bench_default", "bench_dec1", "bench_dec3", "bench_msm20", "bench_msm17", "bench_msm22", "bench_msm20_async2", "bench_msm17_async2", "bench_msm17_async4", "bench_msm20_async4", "bench_msm17_ctx2", "bench_msm17_ctx4", "bench_msm20_ctx2", "bench_msm10", "bench_fb16", "bench_fb6"):
    d = line(name)
    if not d:
        continue
    r = d["roofline"]
    print("%-18s %8.1f M %-13s ms/pass %7.3f  kernel %7.3f ms  tail %6.3f  frac %.3f  whole-pass %.3f  verified %s  build %s" % (
        name, d["value"] / 1e6, d["unit"], d["config"]["ms_per_pass"], r["kernel_ms"], r["tail_ms"], r["frac"], r["whole_pass_frac"], d.get("verified"), r["build_id"]))
d = line("bench_default")
if d:
    fb, fw, cb = d.get("fixed_base"), d.get("fixed_base_wide_window"), d.get("cpu_baseline")
    if fb:
        print("default.fixed_base      %.1f M/s  kernel %.3f ms  frac %.3f  verified %s" % (fb["value"] / 1e6, fb["kernel_ms"], fb["roofline_frac"], fb.get("verified")))
    if fw:
        print("default.wide_window     %.1f M/s  verified %s" % (fw["value"] / 1e6, fw.get("verified")))
    ct = d.get("varbase_vartime") or d.get("varbase_constant_time")      # round 5: the default IS the constant-time ladder, the side measurement the table ladder
    if ct:
        print("default.%s   %.1f M/s  kernel %.3f ms  frac %.3f  x%.3f of the default ladder  verified %s  equal on all units %s" % ("vartime_ladder" if "varbase_vartime" in d else "constant_time",
            ct["value"] / 1e6, ct["kernel_ms"], ct["roofline_frac"], ct["relative_to_default"], ct.get("verified"), ct.get("equals_default_ladder_all_units")))
    print("default.frac_min        %.4f (fastest dispatch), frac %.4f" % (d["roofline"].get("frac_min") or 0, d["roofline"]["frac"]))
    if cb:
        print("cpu_baseline            one thread %.0f /s, %d threads %.0f /s (x%.1f); %s, logical %d, cgroup quota %s" % (
            cb["single_thread"]["value"], cb["all_cores"]["threads"], cb["all_cores"]["value"], cb["all_cores"]["speedup_over_one_thread"],
            cb["cpu"]["model"], cb["cpu"]["logical_cpus"], cb["cpu"]["cgroup_cpu_quota"]))
for wl in ("varbase", "fixedbase", "msm", "decompress"):
    print("pmc %-11s %s" % (wl, pmc(wl)))
