mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "msm" 2>&1 | tail -3
for l in 17 20 24; do timeout 300 python bench.py --workload msm --log2n $l --steps 10 --warmup 2 2>&1 | tail -1 > gpurun_out/msm_$l.json; python - <<PY
import json; d=json.load(open("gpurun_out/msm_$l.json")); print("log2n $l", round(d["value"]/1e6,1), "M/s", round(d["ms_per_step"],3), "ms")
PY
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_msm5 -o msm -- python $GRAFT_REPO_ROOT/bench.py --workload msm --steps 6 --warmup 2 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_msm5 | head; python tools/rocpd_summary.py gpurun_out/prof_msm5/*.db 2>/dev/null | head -30
