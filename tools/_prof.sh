mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1g_msm -o msm -- python $GRAFT_REPO_ROOT/bench.py --workload msm --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/bench_msm_prof.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1g_dec -o dec -- python $GRAFT_REPO_ROOT/bench.py --workload decompress --decompress-flags 15 --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/bench_dec_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof_r1g_msm/*.db > gpurun_out/r1_msm_kernel_stats.txt
python tools/rocpd_summary.py gpurun_out/prof_r1g_dec/*.db > gpurun_out/r1_decompress_kernel_stats.txt
python bench.py > gpurun_out/r1_bench_default.json 2>/dev/null
python bench.py --workload msm > gpurun_out/r1_bench_msm.json 2>/dev/null
python bench.py --workload decompress > gpurun_out/r1_bench_decompress.json 2>/dev/null
python bench.py --workload decompress --decompress-flags 3 > gpurun_out/r1_bench_decompress_subgroup.json 2>/dev/null
python bench.py --workload fixedbase > gpurun_out/r1_bench_fixedbase.json 2>/dev/null
tail -c 600 gpurun_out/r1_bench_default.json
