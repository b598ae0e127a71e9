cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_lds; mkdir -p gpurun_out/pmc_lds
for sel in shuffle gather; do
  for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"; do
    tag=$(echo $set | cut -d' ' -f1)
    JJ_FIXEDBASE_SELECT=$sel rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_lds/${sel}_$tag -o pmc -- python bench.py --workload fixedbase --steps 1 --warmup 1 --passes 1 --no-cpu-baseline --no-extras --no-verify > gpurun_out/pmc_lds/${sel}_$tag.log 2>&1
  done
done
python - <<'PY'
import csv, glob, os
for sel in ("shuffle", "gather"):
    print("==", sel)
    for f in sorted(glob.glob("gpurun_out/pmc_lds/%s_*/**/*counter_collection.csv" % sel, recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if "k_fixedbase_comb" in r.get("Kernel_Name", "")]
        last = {}
        for r in rows:
            last.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in last.items():
            print("  %-28s %.4g (dispatches %d)" % (k, v[-1], len(v)))
PY
