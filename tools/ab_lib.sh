#!/bin/bash
# Runs ON THE GPU BOX: alternating bench.py runs of one MSM size with the shipped library and a variant build (JJ_LIB_PATH).
#   bash tools/ab_lib.sh <log2n> <path of the variant .so> [bench.py arguments]
LOG2N=$1; VARIANT=$2; shift; shift
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for which in shipped variant; do
    if [ $which = variant ]; then export JJ_LIB_PATH=$PWD/$VARIANT; else unset JJ_LIB_PATH; fi
    python bench.py --workload msm --log2n $LOG2N --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('2^$LOG2N  %-8s %.4f ms per call (%.4f without the events), verified %s' % ('$which', c['ms_per_pass'], c.get('ms_per_call_without_events', {}).get('median', float('nan')), d['verified']))"
  done
done
unset JJ_LIB_PATH
