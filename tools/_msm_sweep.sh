cd $GRAFT_REPO_ROOT
run() { echo "== LOG2N=$LOG2N $*"; env "$@" timeout 600 python bench.py --workload msm --log2n ${LOG2N:-20} --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', round(d['value']/1e6,1), 'M terms/s', round(d['ms_per_step']/d['config'].get('passes_per_step',32),4), 'ms/MSM', d['verified'])"; }
for l in 17 16 15 14 12; do
LOG2N=$l run JJ_MSM_ACCUM=chunks
LOG2N=$l run JJ_MSM_ACCUM=segments
LOG2N=$l run JJ_MSM_ACCUM=segments JJ_MSM_SEG_LEN=16
LOG2N=$l run JJ_MSM_ACCUM=segments JJ_MSM_SEG_LEN=64
done
