#!/bin/bash
# the floor of the fixed-base comb.  Three builds of the library -- the shipped comb (two ds_bpermute rounds + mask select), a probe
# with ONE shuffle round, a probe WITHOUT any select (both probes give wrong points: -DJJ_EXPERIMENTS -DJJ_FBC_PROBE) -- each timed by bench.py and
# counted by rocprofv3 --pmc (separate passes, kernel trace only).   Build the probes first (CPU): python tools/fixedbase_floor.py build
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; export TMPDIR=/tmp; cd "$ROOT"
D=gpurun_out/pmc_fbfloor; rm -rf $D; mkdir -p $D
declare -A LIBS=( [two_rounds]="" [one_round]="$ROOT/experiments/probe_lib/libjj_fbc_probe2.so" [no_select]="$ROOT/experiments/probe_lib/libjj_fbc_probe1.so" )
for v in two_rounds one_round no_select; do
  export JJ_LIB_PATH="${LIBS[$v]}"; [ -z "$JJ_LIB_PATH" ] && unset JJ_LIB_PATH
  python bench.py --workload fixedbase --steps 3 --warmup 1 --passes 4 --no-cpu-baseline --no-extras --no-verify > $D/$v.json 2> $D/$v.err
  for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    tag=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $D/${v}_$tag -o pmc -- python bench.py --workload fixedbase --steps 1 --warmup 1 --passes 1 --no-cpu-baseline --no-extras --no-verify > $D/${v}_$tag.log 2>&1
  done
done
unset JJ_LIB_PATH
python - <<'PY'
import csv, glob, json
print("# tools/fixedbase_floor.sh: k_fixedbase_comb<true>, 2^24 units per dispatch; probes one_round / no_select compute WRONG points (select partly / wholly compiled out)")
print("# bench: python bench.py --workload fixedbase --steps 3 --warmup 1 --passes 4 --no-verify;  counters: rocprofv3 --kernel-trace --pmc <set> (last dispatch)")
print("%-11s %9s %9s %8s | %12s %12s %9s | %11s %11s %12s %12s" % ("variant", "M units/s", "kernel ms", "frac", "VALU instr", "busy cycles", "VALU util", "LDS instr", "LDS active", "bank confl.", "addr confl."))
for v in ("two_rounds", "one_round", "no_select"):
    try:
        d = json.loads(open("gpurun_out/pmc_fbfloor/%s.json" % v).read().strip().splitlines()[-1])
    except Exception as e:
        print(v, "bench failed:", e); continue
    vals = {}
    for f in sorted(glob.glob("gpurun_out/pmc_fbfloor/%s_*/**/*counter_collection.csv" % v, recursive=True)):
        for r in csv.DictReader(open(f)):
            if "k_fixedbase_comb" in r.get("Kernel_Name", ""):
                vals[r["Counter_Name"]] = float(r["Counter_Value"])
    g = lambda k: vals.get(k, float("nan"))
    print("%-11s %9.1f %9.3f %8.4f | %12.4g %12.4g %9.3f | %11.4g %11.4g %12.4g %12.4g" % (v, d["value"] / 1e6, d["roofline"]["kernel_ms"], d["roofline"]["frac"], g("SQ_INSTS_VALU"), g("SQ_BUSY_CYCLES"),
          g("SQ_INSTS_VALU") / g("SQ_BUSY_CYCLES") / 8, g("SQ_INSTS_LDS"), g("SQ_ACTIVE_INST_LDS"), g("SQ_LDS_BANK_CONFLICT"), g("SQ_LDS_ADDR_CONFLICT")))
print("# round 6 (VERDICT r5 next #7): where the waves' cycles go -- fractions of SQ_WAVE_CYCLES (summed over the resident waves): waiting for anything, waiting for an instruction to issue, waiting on an LDS instruction")
print("%-11s %12s %10s %14s %14s" % ("variant", "wave cycles", "WAIT_ANY", "WAIT_INST_ANY", "WAIT_INST_LDS"))
for v in ("two_rounds", "one_round", "no_select"):
    vals = {}
    for f in sorted(glob.glob("gpurun_out/pmc_fbfloor/%s_*/**/*counter_collection.csv" % v, recursive=True)):
        for r in csv.DictReader(open(f)):
            if "k_fixedbase_comb" in r.get("Kernel_Name", ""):
                vals[r["Counter_Name"]] = float(r["Counter_Value"])
    g = lambda k: vals.get(k, float("nan"))
    print("%-11s %12.4g %10.3f %14.3f %14.3f" % (v, g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES")))
print("# frac of the probes uses the shipped kernel's credited work (31 108 IMAD32 per unit): the additions and doublings are all there, only the entry is the wrong one")
PY
