set -x
timeout 900 python -m pytest tests/test_gpu_host_path.py -x -q 2>&1 | tail -5
for i in 1 2; do
for lib in "" experiments/probe_lib/libjj_fbc768.so; do
  JJ_LIB_PATH=$lib timeout 300 python bench.py --workload fixedbase --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('fixedbase lib=[$lib]: %.1f M/s kernel_ms %.2f frac %.3f verified %s' % (d['value']/1e6, r['kernel_ms'], r['frac'], d.get('verified')))" >> gpurun_out/fbc_ab.txt
done; done
cat gpurun_out/fbc_ab.txt
