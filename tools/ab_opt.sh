#!/bin/bash
# Runs ON THE GPU BOX: alternating bench.py runs of one MSM size with option settings in turn (each argument after the size: "key=value[,key=value]").
#   bash tools/ab_opt.sh 17 msm_fixup_quad=1 msm_fixup_quad=0      -> one line per run: the settings, ms per synchronous call, verified
LOG2N=$1; shift
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for o in "$@"; do
    args=""; for kv in ${o//,/ }; do args="$args --opt $kv"; done
    python bench.py $args --workload msm --log2n $LOG2N --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('2^$LOG2N  %-34s %.4f ms per call (%.4f without the events), verified %s' % ('$o', c['ms_per_pass'], c.get('ms_per_call_without_events', {}).get('median', float('nan')), d['verified']))"
  done
done
