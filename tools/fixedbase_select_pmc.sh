#!/bin/bash
# LDS counters of the fixed-base comb with the shuffle select (ds_bpermute, the default) and with the per-lane LDS gather: how busy the LDS
# unit is, and how many of its cycles are bank / address conflicts of the secret-index-dependent lane pattern.  Runs on the GPU box.
# The gather select exists in -DJJ_EXPERIMENTS builds only (the shipped library has no switch that reaches the timing discipline): build the probe
# library first (CPU): python tools/fixedbase_floor.py build
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; export TMPDIR=/tmp; cd "$ROOT"
export JJ_LIB_PATH="$ROOT/experiments/probe_lib/libjj_experiments.so"
[ -f "$JJ_LIB_PATH" ] || { echo "build the probe library first: python tools/fixedbase_floor.py build"; exit 1; }
D=gpurun_out/pmc_lds; rm -rf $D; mkdir -p $D
for sel in shuffle gather; do
  for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
    tag=$(echo $set | cut -d' ' -f1)
    JJ_FIXEDBASE_SELECT=$sel rocprofv3 --kernel-trace --pmc $set --output-format csv -d $D/${sel}_$tag -o pmc -- python bench.py --workload fixedbase --steps 1 --warmup 1 --passes 1 --no-cpu-baseline --no-extras --no-verify > $D/${sel}_$tag.log 2>&1
  done
done
python - <<'PY'
import csv, glob
print("# rocprofv3 --kernel-trace --pmc <set> -- python bench.py --workload fixedbase --steps 1 --warmup 1 --passes 1 --no-cpu-baseline --no-extras --no-verify")
print("# last k_fixedbase_comb dispatch (2^24 units) of each pass; JJ_FIXEDBASE_SELECT=shuffle (ds_bpermute select, default) | gather (per-lane LDS read)")
for sel in ("shuffle", "gather"):
    print("==", sel)
    for f in sorted(glob.glob("gpurun_out/pmc_lds/%s_*/**/*counter_collection.csv" % sel, recursive=True)):
        last = {}
        for r in csv.DictReader(open(f)):
            if "k_fixedbase_comb" in r.get("Kernel_Name", ""):
                last[r["Counter_Name"]] = float(r["Counter_Value"])
        for k, v in last.items():
            print("  %-28s %.4g" % (k, v))
PY
