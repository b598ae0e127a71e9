mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "msm" 2>&1 | tail -3
for l in 20 21 22 24; do timeout 300 python bench.py --workload msm --log2n $l --steps 10 --warmup 2 2>&1 | tail -1 > gpurun_out/msm_$l.json; python - <<PY
import json; d=json.load(open("gpurun_out/msm_$l.json")); print("log2n $l", round(d["value"]/1e6,1), "M/s", round(d["ms_per_step"],3), "ms")
PY
done
JJ_MSM_PASS_LOG2=22 timeout 300 python bench.py --workload msm --log2n 24 --steps 10 --warmup 2 2>&1 | tail -1 | cut -c1-150
JJ_MSM_WINDOW=15 timeout 300 python bench.py --workload msm --log2n 22 --steps 10 --warmup 2 2>&1 | tail -1 | cut -c1-150
