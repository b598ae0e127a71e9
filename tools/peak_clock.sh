#!/bin/bash
# the 10 % between the measured roofline denominator (k_peak_mad, ~35.3 T mads/s) and the nominal 1024 SIMDs x 16 lanes x 2.4 GHz
# = 39.3 T: clock or issue rate?  GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration = the clock a kernel ran at; that x 1024 / SQ_INSTS_VALU =
# cycles per wave-instruction per SIMD.  Runs on the GPU box; -> gpurun_out/peak_clock.txt
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; export TMPDIR=/tmp; cd "$ROOT"
D=gpurun_out/pmc_peak; rm -rf $D; mkdir -p $D
[ -x experiments/peak_clock/probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o experiments/peak_clock/probe experiments/peak_clock/probe.hip 2>/dev/null
./experiments/peak_clock/probe > $D/probe_plain.log 2>&1                              # un-profiled, for the HIP-event rates
(cd /tmp && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $ROOT/$D/probe -o pmc -- $ROOT/experiments/peak_clock/probe > $ROOT/$D/probe.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $ROOT/$D/lib -o pmc -- python $ROOT/tools/peak_clock.py > $ROOT/$D/lib.log 2>&1)
python tools/peak_clock.py > $D/lib_plain.log 2>&1
python - <<'PY' | tee gpurun_out/peak_clock.txt
import csv, glob, collections
print("# tools/peak_clock.sh: rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES; per kernel: mean over its dispatches but the first")
print("# clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; cycles per VALU wave-instruction per SIMD = clock cycles x 1024 SIMDs / SQ_INSTS_VALU; busy = SQ_BUSY_CYCLES / 32 SEs / clock cycles")
for tag in ("probe", "lib"):
    dur = {}
    for f in glob.glob("gpurun_out/pmc_peak/%s/**/*kernel_trace.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    cnt = collections.defaultdict(dict)
    for f in glob.glob("gpurun_out/pmc_peak/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    per = collections.defaultdict(list)
    for did, (name, ns) in sorted(dur.items(), key=lambda kv: int(kv[0])):
        c = cnt.get(did, {})
        if "GRBM_GUI_ACTIVE" in c:
            per[name.split("(")[0]].append((ns, c["GRBM_GUI_ACTIVE"], c.get("SQ_INSTS_VALU", 0.0), c.get("SQ_BUSY_CYCLES", 0.0)))
    print("== %s" % ("experiments/peak_clock/probe" if tag == "probe" else "tools/peak_clock.py (the library: jj_peak_imad32_samples + jj_varbase_mul, one process)"))
    for name, rows in per.items():
        if not any(k in name for k in ("k_pure", "k_mix", "k_peak_mad", "k_varbase_ct3")):
            continue
        rows = rows[1:] if len(rows) > 1 else rows
        ns = sum(r[0] for r in rows) / len(rows); g = sum(r[1] for r in rows) / len(rows) / 8; v = sum(r[2] for r in rows) / len(rows); b = sum(r[3] for r in rows) / len(rows) / 32
        print("%-34s %2d dispatches  %9.3f ms  clock %.3f GHz  VALU wave-instr %.4g  cycles/instr/SIMD %.3f  SQ busy %.3f" % (name[:34], len(rows), ns / 1e6, g / ns, v, g * 1024 / max(v, 1), b / max(g, 1)))
print("# un-profiled HIP-event rates of the same kernels:")
for f in ("gpurun_out/pmc_peak/probe_plain.log", "gpurun_out/pmc_peak/lib_plain.log"):
    for ln in open(f):
        if "rep 4" in ln or "k_peak_mad" in ln or ln.startswith("#"):
            print("  " + ln.rstrip())
PY
