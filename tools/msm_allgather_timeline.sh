#!/bin/bash
# Runs ON THE GPU BOX: start / end of every kernel and device copy of a stream of distributed MSMs -- one rank of G with the other ranks played by
# tools/loopback_comm.cpp, four jj_msm_allgather_begin jobs in flight (experiments/misc/msm_allgather_pipeline.py's inner loop) -- under
# rocprofv3 --kernel-trace --memory-copy-trace: which kernels of the next MSM run beside the gather, the fold and the host tail of the one before.
#   bash tools/msm_allgather_timeline.sh [log2n] [G]   -> gpurun_out/msm_allgather_timeline.txt (stdout too)
LOG2N=${1:-20}; G=${2:-8}
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
D=$R/gpurun_out/prof_tl_allgather
rm -rf $D
cat > /tmp/jj_allgather_stream.py <<PY
import os, sys
sys.path.insert(0, "$R"); sys.path.insert(0, os.path.join("$R", "tests"))
import torch
from jubjub_amd import Engine
from util import LoopbackComm
G, n = $G, 1 << $LOG2N
m = n // G
eng = Engine(0); dev = torch.device("cuda", 0)
S = eng.synth_scalars(n, 7, 0, device=dev); P = eng.random_points(n, 7, 0, subgroup=False, device=dev)
recs = torch.stack([eng.msm_partial(S[g * m:(g + 1) * m], P[g * m:(g + 1) * m]) for g in range(G)])
comm = LoopbackComm(0, G); comm.add_round(recs); eng.set_comm(comm)
s, p = S[:m], P[:m]
pend = []
for _ in range(24):
    pend.append(eng.msm_allgather_begin(s, p))
    if len(pend) == 4:
        eng.msm_finish(pend.pop(0))
for j in pend:
    eng.msm_finish(j)
torch.cuda.synchronize()
eng.set_comm(None); comm.close(); eng.close()
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $D -o ag -- python /tmp/jj_allgather_stream.py > $D.log 2>&1)
python3 - "$LOG2N" "$G" "$D" <<'PY' | tee gpurun_out/msm_allgather_timeline.txt
import glob, os, sqlite3, sys
lg, G, d = sys.argv[1:4]
db = sqlite3.connect(glob.glob(os.path.join(d, "**", "*results.db"), recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = [r for r in db.execute("select name, start, end, %s from kernels order by start" % q).fetchall() if "jj::" in r[0]]
folds = [i for i, r in enumerate(rows) if "k_msm_fold_records" in r[0]]
# three MSMs out of the steady state: from the first kernel after the fold of job 16 to the fold of job 19
lo, hi = folds[15] + 1, folds[18] + 1
sel = rows[lo:hi]
t0 = sel[0][1]
print("# one rank of %s, 2^%s-term MSM cut by terms, four jj_msm_allgather_begin jobs in flight, steady state: kernel, queue, start us, end us, duration us" % (G, lg))
print("# (the all-gather itself is two device copies of the loopback communicator between a job's last reduce kernel and its k_msm_fold_records: not in the kernel trace)")
for name, s, e, qid in sel:
    print("%-46s q%-4s %9.1f %9.1f %8.1f" % (name.split("(")[0][:46], qid, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
span = (rows[folds[18]][2] - rows[folds[15]][2]) / 1e3
print("# fold of job 16 -> fold of job 19: %.1f us = %.1f us per MSM; sum of kernel durations in between %.1f us (kernels of two lanes overlap)" % (span, span / 3, sum(e - s for _, s, e, _ in sel) / 1e3))
PY
rm -rf $D /tmp/jj_allgather_stream.py
