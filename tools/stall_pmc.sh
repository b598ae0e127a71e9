#!/bin/bash
# Where the issue slots of the main kernels go: instruction fetch, vector-memory latency, waves resident -- for the var-base ladder (issue
# utilisation 0.96-0.99), the fixed-base comb, the decoder and the MSM's accumulation (0.90-0.93).  Runs on the GPU box.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; export TMPDIR=/tmp; cd "$ROOT"
D=gpurun_out/pmc_stall; rm -rf $D; mkdir -p $D
for wl in varbase fixedbase decompress msm; do
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LEVEL_WAVES SQ_INSTS_SALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH SQ_CYCLES"; do
    tag=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $D/${wl}_$tag -o pmc -- python bench.py --workload $wl --steps 1 --warmup 1 --passes 1 --no-cpu-baseline --no-extras --no-verify > $D/${wl}_$tag.log 2>&1
  done
done
python - <<'PY'
import csv, glob
KER = {"varbase": "k_varbase_ct3", "fixedbase": "k_fixedbase_comb", "decompress": "k_decompress<", "msm": "k_msm_accumulate_seg"}
print("# rocprofv3 --kernel-trace --pmc <set> -- python bench.py --workload W --steps 1 --warmup 1 --passes 1 --no-cpu-baseline --no-extras --no-verify   (last dispatch of the named kernel)")
for wl, ker in KER.items():
    vals = {}
    for f in sorted(glob.glob("gpurun_out/pmc_stall/%s_*/**/*counter_collection.csv" % wl, recursive=True)):
        for r in csv.DictReader(open(f)):
            if ker in r.get("Kernel_Name", ""):
                vals[r["Counter_Name"]] = float(r["Counter_Value"])
    print("==", wl, ker)
    for k in sorted(vals):
        print("  %-24s %.4g" % (k, vals[k]))
    try:
        # SQ_BUSY_CYCLES is summed over the 32 shader engines, SQ_INSTS_VALU over the whole GPU (1024 SIMDs, 4 cycles per wave64 instruction)
        print("  derived: VALU issue utilisation while the SQs are busy = SQ_INSTS_VALU x 4 / 1024 / (SQ_BUSY_CYCLES / 32) = %.3f; "
              "instruction fetch: %.2f fetches in flight on average per fetch issued; vector-memory: %.0f wave-cycles in flight per load; waves launched %d" % (
                  vals["SQ_INSTS_VALU"] / vals["SQ_BUSY_CYCLES"] / 8, vals["SQ_IFETCH_LEVEL"] / max(vals["SQ_IFETCH"], 1),
                  vals["SQ_INST_LEVEL_VMEM"] / max(vals["SQ_INSTS_VMEM_RD"], 1), vals["SQ_WAVES"]))
    except KeyError as e:
        print("  (missing %s)" % e)
PY
