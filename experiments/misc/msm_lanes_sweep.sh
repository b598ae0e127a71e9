# MSM lanes / jobs in flight at large sizes (one context, one host thread): is a third / fourth lane worth it when the host thread is not the bound?
cd $GRAFT_REPO_ROOT
for l in 20 19 18; do
  for cfg in "1 1" "2 2" "2 4" "3 3" "3 6" "4 4" "4 8"; do
    set -- $cfg
    JJ_MSM_LANES=$1 python bench.py --workload msm --log2n $l --msm-async $2 --no-cpu-baseline --steps 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('2^$l lanes $1 jobs $2:', round(d['config']['ms_per_pass'],4), 'ms/MSM', round(d['value']/1e6,1), 'M terms/s frac', round(d['roofline']['frac'],3), d['verified'])"
  done
done
