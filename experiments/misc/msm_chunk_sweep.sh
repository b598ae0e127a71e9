#!/bin/bash
# Runs ON THE GPU BOX: accumulation chunk length x input size for the chunked Pippenger path (time per MSM, ms; every line verified).
cd "$(dirname "$0")/../.."
for lg in 15 16 17; do
  for ch in 0 12 16 20 24 32; do
    r=$(JJ_MSM_CHUNK=$ch python bench.py --workload msm --log2n $lg --steps 200 --warmup 20 --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.4f %s' % (d['config']['ms_per_pass'], d['verified']))")
    echo "2^$lg chunk=$ch : $r"
  done
done
for lg in 16 17; do
  for L in 1 2 4 8; do
    r=$(JJ_MSM_REDUCE_CHUNK=$L python bench.py --workload msm --log2n $lg --steps 200 --warmup 20 --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.4f %s' % (d['config']['ms_per_pass'], d['verified']))")
    echo "2^$lg reduce chunk L=$L : $r"
  done
done
