JJ_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --workload msm --msm-exchange c --no-cpu-baseline > gpurun_out/r4_bench_msm20_rccl1.json 2> gpurun_out/rccl1.err; python -c "
import json; d=json.load(open('gpurun_out/r4_bench_msm20_rccl1.json')); print('rccl1:', d['rccl_world_size'], '%.1f M terms/s' % (d['value']/1e6), d['verified'], d['config']['parallelism'][-90:])"
grep -c "RCCL version" gpurun_out/rccl1.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras | wc -l
timeout 2400 python -m pytest tests/test_gpu_dist.py tests/test_gpu_host_path.py -x -q -m gpu > gpurun_out/t_dist.txt 2>&1; tail -3 gpurun_out/t_dist.txt
