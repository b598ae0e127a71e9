cd $GRAFT_REPO_ROOT
run() { echo "== LOG2N=$LOG2N $*"; env "$@" timeout 600 python bench.py --workload msm --log2n ${LOG2N:-20} --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', round(d['value']/1e6,1), 'M terms/s', round(d['ms_per_step']/d['config'].get('passes_per_step',32),4), 'ms/MSM', d['verified'])"; }
for l in 17 16 15 13; do for c in 9 10 11 12; do LOG2N=$l run JJ_MSM_WINDOW=$c; done; done
for l in 11 10; do for c in 8 9 10 11; do LOG2N=$l run JJ_MSM_WINDOW=$c; done; done
