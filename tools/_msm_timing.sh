cd $GRAFT_REPO_ROOT
for l in 20 17; do
JJ_MSM_TIMING=1 timeout 600 python bench.py --workload msm --log2n $l --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-verify 2>&1 | grep "jj msm" | tail -4
done
