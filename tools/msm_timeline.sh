#!/bin/bash
# Runs ON THE GPU BOX: start / end of every kernel of ONE MSM call (the last of a short bench run) under rocprofv3 --kernel-trace:
# where the time of a call goes between the kernels.   bash tools/msm_timeline.sh <log2n> [bench.py arguments, e.g. --opt msm_front1=0 ...]  -> gpurun_out/msm<log2n>_timeline.txt (stdout too)
LOG2N=$1; shift
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
D=$R/gpurun_out/prof_tl_msm$LOG2N
rm -rf $D
(cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $D -o msm -- python $R/bench.py --workload msm --log2n $LOG2N --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $D.log 2>&1)
python3 - "$LOG2N" "$D" <<'PY' | tee gpurun_out/msm${LOG2N}_timeline.txt
import glob, os, sqlite3, sys
lg, d = sys.argv[1:3]
db = sqlite3.connect(glob.glob(os.path.join(d, "**", "*results.db"), recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = db.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
# the last MSM: the kernels after the second-to-last record-writing kernel (reduce_fold / small_sum) up to the last one
rows = [r for r in rows if "jj::" in r[0] and "k_peak_mad" not in r[0]]
ends = [i for i, r in enumerate(rows) if "k_msm_reduce_fold" in r[0] or "k_msm_small_sum" in r[0] or "k_msm_reduce_l2" in r[0]]
sel = rows[ends[-2] + 1: ends[-1] + 1]
t0 = sel[0][1]
print("# 2^%s-term MSM, last call of the run: kernel, queue, start us, duration us, gap to the previous END on any queue us" % lg)
prev_end = t0
for name, s, e, qid in sel:
    print("%-46s q%-4s %9.1f %9.1f %8.1f" % (name.split("(")[0][:46], qid, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = max(prev_end, e)
print("# first kernel start -> last kernel end: %.1f us; sum of kernel durations %.1f us" % ((prev_end - t0) / 1e3, sum(e - s for _, s, e, _ in sel) / 1e3))
PY
rm -rf $D
