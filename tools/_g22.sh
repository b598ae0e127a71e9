rm -f gpurun_out/fbc_single.txt
b() { # label lib
  JJ_LIB_PATH=$2 timeout 600 python bench.py --workload fixedbase --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1: %.1f M/s kernel_ms %.3f frac %.3f verified %s block %s' % (d['value']/1e6, r['kernel_ms'], r['frac'], d.get('verified'), d.get('verified_block',{}).get('ok')))" >> gpurun_out/fbc_single.txt
}
for i in 1 2; do
  b default_512_double ""
  b 768_single experiments/probe_lib/libjj_fbc_768_1.so
  b 512_single experiments/probe_lib/libjj_fbc_512_1.so
  b 1024_single experiments/probe_lib/libjj_fbc_1024_1.so
done
cat gpurun_out/fbc_single.txt
