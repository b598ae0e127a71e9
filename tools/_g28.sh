bash tools/refresh_round.sh r4 2>&1 | tail -3
python bench.py --workload msm > gpurun_out/r4_bench_msm20_cpu.json 2>/dev/null
timeout 300 python tests/soak_host.py 120 > gpurun_out/r4_soak_host.txt 2>&1; tail -1 gpurun_out/r4_soak_host.txt
