for l in 17 20; do for c in 13 14 15 16; do JJ_MSM_WINDOW=$c timeout 300 python bench.py --workload msm --log2n $l --steps 10 --warmup 2 2>&1 | tail -1 > /tmp/m.json; python - <<PY
import json; d=json.load(open("/tmp/m.json")); print("log2n $l c $c", round(d["value"]/1e6,1), "M/s", round(d["ms_per_step"],3), "ms")
PY
done; done
