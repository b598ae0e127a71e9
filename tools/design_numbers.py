#!/usr/bin/env python3
"""Prints the numbers DESIGN.md §6 quotes, straight from the committed bench lines and PMC summaries in profiles/
(python tools/design_numbers.py [tag]).  Nothing is typed by hand in that table."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
tag = args[0] if args else "r6"
MARKDOWN, WRITE = "--markdown" in sys.argv, "--write" in sys.argv      # --markdown: DESIGN.md section 6 as a table; --write: replace it in DESIGN.md between the markers


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


def markdown():
    """one row per bench line of the evidence run (profiles/<tag>_bench_*.json), the PMC figures of the four dominant kernels beside them."""
    out = []
    rows = [("bench_default", "var-base 2^20 (configs[1]; headline: constant-time ladder `jj_varbase_mul`)", "varbase"),
            ("bench_fb16", "fixed-base 2^24, 16-bit windows (64 MB table, variable-time gather)", None),
            ("bench_fb6", "fixed-base 2^24, signed 6-bit windows in LDS (round 2's kernel)", None),
            ("fixedbase_bench", "fixed-base 2^24 (configs[2]): signed comb, LDS table, shuffle select (profiled run)", "fixedbase"),
            ("bench_msm20", "MSM 2^20 terms (configs[3] on one GPU), one synchronous call", "msm"),
            ("bench_msm20_async2", "MSM 2^20, two jobs in flight", None), ("bench_msm20_async4", "MSM 2^20, four jobs in flight", None),
            ("bench_msm22", "MSM 2^22 terms", None), ("bench_msm19", "MSM 2^19 terms", None), ("bench_msm18", "MSM 2^18 terms", None),
            ("bench_msm17", "MSM 2^17 terms (configs[3]'s per-GPU share at 8 GPUs), one synchronous call", None),
            ("bench_msm17_async2", "MSM 2^17, two jobs in flight", None), ("bench_msm17_async4", "MSM 2^17, four jobs in flight", None),
            ("bench_msm17_ctx2", "MSM 2^17, two contexts x two jobs", None), ("bench_msm10", "MSM 2^10 terms (small-batch path)", None),
            ("bench_dec1", "decompress 2^23, decode only (ZIP-216 flag)", "decompress"),
            ("decompress_bench", "decompress 2^23 + small-order check + cofactor clear (configs[4] per GPU)", None),
            ("bench_dec3", "decompress 2^23 + full torsion-free check (Tate pairing)", None)]
    out.append("| Workload | Throughput | ms per pass | Dominant kernel (HIP events) | `frac` (measured multiply-add peak) | `frac_nominal` (39.32 T) | VALU lane-instr per unit; issue interval; clock; fabric B per unit | verified |")
    out.append("|---|---|---|---|---|---|---|---|")
    bid = None
    for name, what, pw in rows:
        d = line(name)
        if not d:
            continue
        r = d["roofline"]; bid = bid or r.get("build_id")
        c = pmc(pw) if pw else {}
        pm = "%s; %s cycles; %s GHz; %s B" % (c["instr"], c["interval"], c["ghz"], c["bytes"]) if c else ""
        out.append("| %s | **%.1f M %s** | %.3f | `%s` %.3f ms | **%.3f** | %.3f | %s | %s |" % (
            what, d["value"] / 1e6, d["unit"].replace(" per GPU", ""), d["config"]["ms_per_pass"], (r.get("kernel", "?").split(":")[0]), r["kernel_ms"], r["frac"], r.get("frac_nominal") or 0, pm,
            "yes" if d.get("verified") else str(d.get("verified"))))
    d = line("bench_default")
    if d:
        fb, fw, ct, cb = d.get("fixed_base"), d.get("fixed_base_wide_window"), d.get("varbase_vartime"), d.get("cpu_baseline")
        out.append("")
        if fb:
            out.append("* fixed-base 2^24 as an extra of the default run (the driver's line): **%.1f M scalar-muls/s**, `k_fixedbase_comb<true>` %.3f ms, frac **%.3f**, verified %s." % (fb["value"] / 1e6, fb["kernel_ms"], fb["roofline_frac"], fb.get("verified")))
        if ct:
            out.append("* the table ladder (`jj_varbase_mul_vartime`) on the same batch: %.1f M/s = %.3f x the constant-time default, frac %.3f, equal on all units: %s." % (ct["value"] / 1e6, ct["relative_to_default"], ct["roofline_frac"], ct.get("equals_default_ladder_all_units")))
        r = d["roofline"]
        out.append("* roofline denominator of that line: median of 5 + 5 `k_peak_mad` samples = %.2f T multiply-adds/s (min %.2f, max %.2f); nominal 1024 SIMDs x 16 lanes x 2.4 GHz = %.2f T; fastest dispatch frac %.4f." % (
            r["peak"], r["peak_samples"]["min"], r["peak_samples"]["max"], r["peak_nominal"], r.get("frac_min") or 0))
        if cb:
            out.append("* CPU baseline on the same box (`cpu_baseline`, kind port: the C restatement of the reference's 252-step ladder): %.0f scalar-muls/s on one thread, %.0f /s on %d threads (%s, cgroup quota %s CPUs)." % (
                cb["single_thread"]["value"], cb["all_cores"]["value"], cb["all_cores"]["threads"], cb["cpu"]["model"], cb["cpu"]["cgroup_cpu_quota"]))
    out.append("* build id of every line: `%s` (sha256 over `jubjub_amd/csrc/*`), commit `%s`." % (bid, open(os.path.join(ROOT, "profiles", "BUILD_COMMIT")).read().strip()))
    return "\n".join(out)


def baseline_table():
    """one row per BASELINE config with frac (measured peak) and frac_nominal (39.32 T) side by side"""
    out = ["| Config | GPU (1 x MI355X) | `frac` (measured multiply-add peak) | `frac_nominal` (1024 SIMDs x 16 lanes x 2.4 GHz) | CPU port of the reference algorithm |", "|---|---|---|---|---|"]
    d, fb, m20, m17, m17a, dec = line("bench_default"), line("fixedbase_bench"), line("bench_msm20"), line("bench_msm17"), line("bench_msm17_async4"), line("decompress_bench")
    d1, m20a = line("bench_dec1"), line("bench_msm20_async4")
    cb = d["cpu_baseline"]
    rn = lambda x: x["roofline"].get("frac_nominal") or 0
    out.append("| 2. var-base 2^20 (constant-time ladder) | **%.1f M scalar-muls/s** (%.2f ms per pass) | **%.3f** | %.3f | %.1f k/s one thread, %.0f k/s on %d threads |" % (
        d["value"] / 1e6, d["config"]["ms_per_pass"], d["roofline"]["frac"], rn(d), cb["single_thread"]["value"] / 1e3, cb["all_cores"]["value"] / 1e3, cb["all_cores"]["threads"]))
    out.append("| 3. fixed-base 2^24 (signed comb, shuffle select) | **%.0f M scalar-muls/s** (%.2f ms) | %.3f | %.3f | |" % (fb["value"] / 1e6, fb["config"]["ms_per_pass"], fb["roofline"]["frac"], rn(fb)))
    out.append("| 4. MSM 2^20 terms on one GPU | %.0f M terms/s per call (%.3f ms); four jobs in flight %.0f M terms/s (%.3f ms) | %.3f per call, %.3f in flight | %.3f, %.3f | |" % (
        m20["value"] / 1e6, m20["config"]["ms_per_pass"], m20a["value"] / 1e6, m20a["config"]["ms_per_pass"], m20["roofline"]["frac"], m20a["roofline"]["frac"], rn(m20), rn(m20a)))
    out.append("| 4. one rank's share at 8 GPUs: MSM 2^17 terms | %.4f ms per call (%s ms without the profiling events); four jobs in flight %.4f ms per MSM | %.3f per call, %.3f in flight | %.3f, %.3f | |" % (
        m17["config"]["ms_per_pass"], ("%.4f" % m17["config"]["ms_per_call_without_events"]["median"]) if "ms_per_call_without_events" in m17["config"] else "?", m17a["config"]["ms_per_pass"],
        m17["roofline"]["frac"], m17a["roofline"]["frac"], rn(m17), rn(m17a)))
    out.append("| 5. decompress 2^23 per GPU (= 2^26 over 8) | decode only %.0f M points/s; + small-order check + `mul_by_cofactor` %.0f M/s | %.3f | %.3f | |" % (d1["value"] / 1e6, dec["value"] / 1e6, d1["roofline"]["frac"], rn(d1)))
    return "\n".join(out)


if "--baseline" in sys.argv:
    print(baseline_table())
    sys.exit(0)
if MARKDOWN:
    md = markdown()
    if WRITE:
        p = os.path.join(ROOT, "DESIGN.md")
        t = open(p).read()
        b, e = t.index("<!-- BEGIN GENERATED"), t.index("<!-- END GENERATED -->")
        b = t.index("\n", b) + 1
        open(p, "w").write(t[:b] + md + "\n" + t[e:])
        print("DESIGN.md section 6 rewritten from profiles/%s_*" % tag)
    else:
        print(md)
    sys.exit(0)

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
