#!/bin/bash
# Runs ON THE GPU BOX: per-kernel breakdown of one MSM size under rocprofv3, in the format of profiles/*_kernel_stats.txt.
#   bash tools/msm_profile.sh <tag> <log2n> [env assignments...]   ->  gpurun_out/<tag>_msm<log2n>_kernel_stats.txt
TAG=$1; LOG2N=$2; shift; shift
for kv in "$@"; do export "$kv"; done
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
D=$R/gpurun_out/prof_${TAG}_msm$LOG2N
rm -rf $D
CMD="python bench.py --workload msm --log2n $LOG2N --steps 3 --warmup 1 --passes 32 --no-cpu-baseline --no-extras"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $D -o msm -- python $R/bench.py --workload msm --log2n $LOG2N --steps 3 --warmup 1 --passes 32 --no-cpu-baseline --no-extras > $D.log 2>&1)
python3 - "$TAG" "$LOG2N" "$D" "$CMD" <<'PY'
import glob, json, os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
sys.path.insert(0, os.getcwd())
import profile_round as P
tag, lg, d, cmd = sys.argv[1:5]
dbs = glob.glob(os.path.join(d, "**", "*results.db"), recursive=True)
line = P.bench_line(d + ".log")
out = os.path.join("gpurun_out", "%s_msm%s_kernel_stats.txt" % (tag, lg))
with open(out, "w") as f:
    f.write(P.header("rocprofv3 --kernel-trace --stats -- " + cmd))
    if line:
        f.write("# bench line of this run: value %.4g %s, ms per MSM %.4f, roofline.kernel_ms %.4f (HIP events), frac %.4f, verified %s\n" % (
            line["value"], line["unit"], line["config"]["ms_per_pass"], line["roofline"]["kernel_ms"], line["roofline"]["frac"], line.get("verified")))
    f.write(P.kernel_stats(dbs[0]) if dbs else "# no rocpd database; see log\n")
print(open(out).read())
PY
rm -rf $D
