rm -f gpurun_out/fb6_single.txt
b() { # label lib
  JJ_LIB_PATH=$2 timeout 600 python bench.py --workload fixedbase --fb-window 6 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: %.1f M/s kernel_ms %.3f frac %.3f verified %s block %s' % (d['value']/1e6, r['kernel_ms'], r['frac'], d.get('verified'), d.get('verified_block',{}).get('ok')))" >> gpurun_out/fb6_single.txt
}
for i in 1 2; do b fb6_512_double ""; b fb6_768_single experiments/probe_lib/libjj_fb6_768_1.so; done
JJ_LIB_PATH=experiments/probe_lib/libjj_fb6_768_1.so python tools/composite_bench.py 22 2>&1 | grep -v amdgpu | tail -4 >> gpurun_out/fb6_single.txt
python tools/composite_bench.py 22 2>&1 | grep -v amdgpu | tail -4 >> gpurun_out/fb6_single.txt
cat gpurun_out/fb6_single.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "fixedbase or composite or comb" 2>&1 | tail -2
