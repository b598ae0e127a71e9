mkdir -p gpurun_out
for L in 8 32; do
cd /tmp && export TMPDIR=/tmp && JJ_MSM_REDUCE_CHUNK=$L JJ_MSM_FOLD=4 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_msm7_$L -o msm -- python $GRAFT_REPO_ROOT/bench.py --workload msm --log2n 20 --steps 6 --warmup 2 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; echo "== L $L"; python tools/rocpd_summary.py gpurun_out/prof_msm7_$L/*.db 2>/dev/null | grep "k_msm\|k_sum\|k_scan\|k_soa" | cut -c1-150
done
