cd $GRAFT_REPO_ROOT
run() { echo "== LOG2N=$LOG2N $*"; env "$@" timeout 600 python bench.py --workload msm --log2n ${LOG2N:-20} --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', round(d['value']/1e6,1), 'M terms/s', round(d['ms_per_step']/d['config'].get('passes_per_step',32),4), 'ms/MSM', d['verified'])"; }
for L in 64 32 16 8 4; do LOG2N=20 run JJ_MSM_REDUCE_CHUNK=$L; done
for L in 32 16 8; do LOG2N=18 run JJ_MSM_REDUCE_CHUNK=$L; done
for L in 16 8 4 2; do LOG2N=17 run JJ_MSM_REDUCE_CHUNK=$L; done
for L in 8 4 2; do LOG2N=14 run JJ_MSM_REDUCE_CHUNK=$L; done
