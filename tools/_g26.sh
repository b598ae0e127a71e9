timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "msm" > gpurun_out/t_msm.txt 2>&1; tail -3 gpurun_out/t_msm.txt
rm -f gpurun_out/reduce_ab.txt
b() { # label lib log2n
  JJ_LIB_PATH=$2 timeout 600 python bench.py --workload msm --log2n $3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 2^$3: %.1f M/s  ms/pass %.4f frac %.3f verified %s' % (d['value']/1e6, d['config']['ms_per_pass'], r['frac'], d.get('verified')))" >> gpurun_out/reduce_ab.txt
}
for i in 1 2; do for l in 17 20 22 15 18; do b old experiments/probe_lib/libjj_fbc_512_1.so $l; b new "" $l; done; done
cat gpurun_out/reduce_ab.txt
