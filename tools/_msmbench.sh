timeout 900 python -m pytest tests -m gpu -x -q -k "msm" 2>&1 | tail -3
for l in 17 20 22; do timeout 300 python bench.py --workload msm --log2n $l --steps 10 --warmup 2 2>&1 | tail -1 > /tmp/m.json; python - <<PY
import json; d=json.load(open("/tmp/m.json")); print("log2n $l", round(d["value"]/1e6,1), "M/s", round(d["ms_per_step"],3), "ms")
PY
done
