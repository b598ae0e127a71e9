#!/usr/bin/env python3
"""
Collects the round's profiling evidence on the GPU box in ONE go and at ONE build, and stamps every file with the build:

  python tools/profile_round.py --tag r3 [--workloads varbase,fixedbase,msm,decompress] [--skip-pmc]

  profiles/<tag>_<workload>_kernel_stats.txt   rocprofv3 --kernel-trace --stats summary of `python bench.py --workload <w> ...`
  profiles/<tag>_<workload>_bench.json         the JSON line that same command printed (roofline.kernel_ms to compare with)
  profiles/<tag>_<workload>_pmc.txt            PMC passes for the dominant kernel (one `--pmc` set per run, kernel-trace only:
                                               SQ set | GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE) with the derived numbers
  profiles/traffic.json                        PMC fabric bytes per launch of the dominant kernel per workload, keyed by the
                                               hash of the kernel sources; bench.py reads it and refuses a stale one
  profiles/<tag>_kernel_resources.txt, <tag>_instr_mix.txt   static tables from hipcc's assembly of the same sources

Header of every file: commit (profiles/BUILD_COMMIT, written before the snapshot leaves the repository), build id (sha256
of jubjub_amd/csrc/*), the command.  Needs rocprofv3 and a GPU; run through gpurun, then copy nothing: it writes profiles/
in place and mirrors the files into gpurun_out/profiles_<tag>/ so that they travel back.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (build_id, WORK)

DOMINANT = {"varbase": "k_varbase_ct3", "fixedbase": "k_fixedbase_comb<", "msm": "k_msm_accumulate_seg", "decompress": "k_decompress<"}
SQ_SET = "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"


def commit():
    p = os.path.join(ROOT, "profiles", "BUILD_COMMIT")
    return open(p).read().strip() if os.path.exists(p) else "unknown (profiles/BUILD_COMMIT missing)"


def header(cmd):
    return "# commit %s | build_id %s (sha256 of jubjub_amd/csrc) | MI355X gfx950\n# command: %s\n" % (commit(), bench.build_id(), cmd)


def run(cmd, log):
    env = dict(os.environ, TMPDIR="/tmp")
    with open(log, "w") as f:
        r = subprocess.run(cmd, shell=True, cwd=ROOT, env=env, stdout=f, stderr=subprocess.STDOUT)
    return r.returncode


def bench_line(log):
    for line in open(log, errors="replace"):
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    return None


def kernel_stats(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), "
        "max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = ["%-62s %6s %14s %12s %12s %12s %6s %5s %5s %5s %7s %8s %10s %5s" % (
        "kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct", "vgpr", "agpr", "sgpr", "lds_B", "scratch", "grid_x", "wg_x")]
    for r in rows:
        name = r[0] if len(r[0]) <= 60 else r[0][:57] + "..."
        out.append("%-62s %6d %14d %12.0f %12d %12d %6.2f %5d %5d %5d %7d %8d %10d %5d" % (
            name, r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total, r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0, r[11] or 0, r[12] or 0))
    return "\n".join(out) + "\n"


def pmc_pass(tag, wl, counters, scratch, bench_args):
    """one rocprofv3 --pmc run (kernel-trace only, as gpurun requires); returns {counter: value of the LAST dispatch of the dominant
    kernel in the pass} (steady state: the first one pays the first touch of the workspaces) and that dispatch's duration"""
    d = os.path.join(scratch, "pmc_%s_%s" % (wl, counters.split()[0]))
    shutil.rmtree(d, ignore_errors=True)
    cmd = "rocprofv3 --kernel-trace --pmc %s --output-format csv -d %s -o pmc -- python bench.py %s" % (counters, d, bench_args)
    run(cmd, d + ".log")
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return {}, cmd
    best = {}
    for row in csv.DictReader(open(files[0])):
        if DOMINANT[wl] not in row["Kernel_Name"]:
            continue
        key = row["Counter_Name"]
        start = int(row["Start_Timestamp"])
        if key not in best or start > best[key][0]:
            best[key] = (start, float(row["Counter_Value"]), int(row["End_Timestamp"]) - start)
    out = {k: v[1] for k, v in best.items()}
    if best:
        out["_dispatch_ns(%s)" % counters.split()[0]] = float(max(v[2] for v in best.values()))
    return out, cmd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r3")
    ap.add_argument("--workloads", default="varbase,fixedbase,msm,decompress")
    ap.add_argument("--skip-pmc", action="store_true")
    a = ap.parse_args()
    prof = os.path.join(ROOT, "profiles")
    scratch = os.path.join(ROOT, "gpurun_out", "prof_%s" % a.tag)
    os.makedirs(scratch, exist_ok=True)
    traffic = {"build_id": bench.build_id(), "commit": commit(), "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate kernel-trace-only passes; "
               "FETCH_SIZE (KB) x 1024 x 2 (gfx950 counts 128-byte read requests as 64 B: MI355X_MICROARCH.md, HBM), WRITE_SIZE (KB) x 1024; "
               "the last (steady-state) dispatch of the dominant kernel in the pass", "workloads": {}}
    for wl in a.workloads.split(","):
        log2n = bench.DEFAULT_LOG2N[wl]
        args = "--workload %s --steps 3 --warmup 1 --passes %d --no-cpu-baseline --no-extras" % (wl, {"msm": 32}.get(wl, 4))
        d = os.path.join(scratch, "stats_%s" % wl)
        shutil.rmtree(d, ignore_errors=True)
        cmd = "rocprofv3 --kernel-trace --stats -d %s -o %s -- python bench.py %s" % (d, wl, args)
        run(cmd, d + ".log")
        dbs = glob.glob(os.path.join(d, "**", "*results.db"), recursive=True)
        line = bench_line(d + ".log")
        with open(os.path.join(prof, "%s_%s_kernel_stats.txt" % (a.tag, wl)), "w") as f:
            f.write(header(cmd))
            if line:
                f.write("# bench line of this run: value %.4g %s, roofline.kernel_ms %.4f (HIP events), frac %.4f, verified %s\n" % (
                    line["value"], line["unit"], line["roofline"]["kernel_ms"], line["roofline"]["frac"], line.get("verified")))
            f.write(kernel_stats(dbs[0]) if dbs else "# no rocpd database produced; see %s.log\n" % d)
        if line:
            json.dump(line, open(os.path.join(prof, "%s_%s_bench.json" % (a.tag, wl)), "w"))
        if a.skip_pmc:
            continue
        pargs = "--workload %s --steps 1 --warmup 1 --passes 1 --no-cpu-baseline --no-extras --no-verify" % wl
        vals, cmds = {}, []
        for cs in (SQ_SET, "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"):
            v, c = pmc_pass(a.tag, wl, cs, scratch, pargs)
            vals.update(v)
            cmds.append(c)
        n = 1 << log2n
        with open(os.path.join(prof, "%s_%s_pmc.txt" % (a.tag, wl)), "w") as f:
            f.write(header(" ; ".join(cmds)))
            f.write("# values of the last (steady-state) %s...> dispatch of each pass (2^%d units per launch); _dispatch_ns: its duration under the profiler\n" % (DOMINANT[wl], log2n))
            for k in sorted(vals):
                f.write("%-28s %18.0f\n" % (k, vals[k]))
            f.write("\nderived:\n")
            if "GRBM_GUI_ACTIVE" in vals:
                f.write("  GRBM_GUI_ACTIVE / 8 XCDs             = %.4g shader cycles per dispatch" % (vals["GRBM_GUI_ACTIVE"] / 8))
                d = vals.get("_dispatch_ns(GRBM_GUI_ACTIVE)")
                f.write("  (=> %.2f GHz over the %.3f ms the dispatch took under the profiler)\n" % (vals["GRBM_GUI_ACTIVE"] / 8 / d, d / 1e6) if d else "\n")
            if "SQ_INSTS_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
                f.write("  VALU wave-instructions per SIMD       = %.4g  => one every %.3f cycles (wave64 minimum: 4)\n" % (
                    vals["SQ_INSTS_VALU"] / 1024, vals["GRBM_GUI_ACTIVE"] / 8 / (vals["SQ_INSTS_VALU"] / 1024)))
            if "SQ_INSTS_VALU" in vals:
                f.write("  VALU lane-instructions per unit      = %.0f\n" % (vals["SQ_INSTS_VALU"] * 64 / n))
            if "SQ_ACTIVE_INST_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
                f.write("  VALU issue utilisation = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GUI cycles / 8 XCDs) = %.3f\n" % (
                    vals["SQ_ACTIVE_INST_VALU"] / (1024 * vals["GRBM_GUI_ACTIVE"] / 8 / 4)))
            if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
                fb, wb = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
                w = bench.WORK[wl]
                f.write("  FETCH_SIZE x 1024 x 2 = %.4g B, WRITE_SIZE x 1024 = %.4g B per launch  =>  %.0f B per unit (algorithmic: %d B)\n" % (
                    fb, wb, (fb + wb) / n, w["bytes"]))
                traffic["workloads"]["%s:%d" % (wl, log2n)] = {
                    "kernel": DOMINANT[wl], "units_per_launch": n, "fetch_bytes": fb, "write_bytes": wb, "bytes_per_launch": fb + wb,
                    "bytes_per_unit": (fb + wb) / n, "algorithmic_bytes_per_unit": w["bytes"]}
    if not a.skip_pmc:
        json.dump(traffic, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
    for tool, name in (("kernel_resources.py", "kernel_resources"), ("instr_mix.py", "instr_mix")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)], capture_output=True, text=True).stdout
        with open(os.path.join(prof, "%s_%s.txt" % (a.tag, name)), "w") as f:
            f.write(header("python tools/%s" % tool) + out)
    mirror = os.path.join(ROOT, "gpurun_out", "profiles_%s" % a.tag)
    os.makedirs(mirror, exist_ok=True)
    for f in glob.glob(os.path.join(prof, "%s_*" % a.tag)) + [os.path.join(prof, "traffic.json")]:
        if os.path.exists(f):
            shutil.copy(f, mirror)
    print("profiles written for build", bench.build_id())


if __name__ == "__main__":
    main()
