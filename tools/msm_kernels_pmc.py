#!/usr/bin/env python3
"""Runs ON THE GPU BOX: issue-side counters of EVERY kernel of one MSM size (not only the dominant one as tools/profile_round.py does):
per kernel, the last dispatch of a short bench run -- waves, VALU wave-instructions, the share of its cycles a SIMD issued VALU work, clock.
Two rocprofv3 passes (kernel-trace + --pmc only, as gpurun requires): the SQ counters, then GRBM_GUI_ACTIVE.
  python tools/msm_kernels_pmc.py <log2n> [bench.py arguments]  ->  gpurun_out/msm<log2n>_kernels_pmc.txt (stdout too)"""
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import profile_round as P  # noqa: E402

SETS = ["SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]


def main():
    lg = sys.argv[1]
    extra = " ".join(sys.argv[2:])
    bargs = "--workload msm --log2n %s --steps 1 --warmup 1 --passes 4 --no-cpu-baseline --no-extras --no-verify %s" % (lg, extra)
    scratch = os.path.join(ROOT, "gpurun_out", "prof_kpmc_msm" + lg)
    os.makedirs(scratch, exist_ok=True)
    per = {}          # kernel -> counter -> (start, value, duration)
    cmds = []
    for cs in SETS:
        d = os.path.join(scratch, cs.split()[0])
        shutil.rmtree(d, ignore_errors=True)
        cmd = "rocprofv3 --kernel-trace --pmc %s --output-format csv -d %s -o pmc -- python bench.py %s" % (cs, d, bargs)
        cmds.append(cmd.replace(ROOT + "/", ""))
        P.run(cmd, d + ".log")
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        for row in csv.DictReader(open(files[0])):
            name = row["Kernel_Name"]
            if "k_msm" not in name and "k_seg" not in name:
                continue
            name = name.split("(")[0].replace("void ", "")
            start = int(row["Start_Timestamp"])
            slot = per.setdefault(name, {})
            key = row["Counter_Name"]
            if key not in slot or start > slot[key][0]:
                slot[key] = (start, float(row["Counter_Value"]), int(row["End_Timestamp"]) - start)
    out = [P.header(" ; ".join(cmds)).rstrip("\n"),
           "# last dispatch of every MSM kernel of a 2^%s-term call.  issue = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the share of the" % lg,
           "# dispatch's SIMD cycles that issued VALU work (1.0 = every SIMD busy for the whole dispatch); interval = SIMD cycles per VALU wave-instruction",
           "# if the work were spread over all 1024 SIMDs; waves/SIMD = SQ_WAVES / 1024 (how many rounds of waves a SIMD sees, not the residency)",
           "%-34s %9s %8s %13s %8s %9s %8s %10s" % ("kernel", "us (pmc)", "waves", "VALU w-instr", "issue", "interval", "GHz", "waves/SIMD")]
    order = sorted(per.items(), key=lambda kv: max(v[0] for v in kv[1].values()))
    for name, c in order:
        g = lambda k: c[k][1] if k in c else float("nan")
        dur = c["GRBM_GUI_ACTIVE"][2] if "GRBM_GUI_ACTIVE" in c else float("nan")
        cyc = g("GRBM_GUI_ACTIVE") / 8.0
        issue = g("SQ_ACTIVE_INST_VALU") * 4.0 / (1024.0 * cyc) if cyc == cyc and cyc > 0 else float("nan")
        interval = 1024.0 * cyc / g("SQ_INSTS_VALU") if g("SQ_INSTS_VALU") else float("nan")
        out.append("%-34s %9.1f %8d %13d %8.3f %9.2f %8.2f %10.2f" % (name[:34], dur / 1e3, g("SQ_WAVES"), g("SQ_INSTS_VALU"), issue, interval, cyc / dur if dur == dur else float("nan"), g("SQ_WAVES") / 1024.0))
    text = "\n".join(out) + "\n"
    with open(os.path.join(ROOT, "gpurun_out", "msm%s_kernels_pmc.txt" % lg), "w") as f:
        f.write(text)
    print(text)
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
