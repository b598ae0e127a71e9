rm -f gpurun_out/cmid.txt
for c in 16 8 4; do
  JJ_DEC_C_MID=$c JJ_NORM_C_MID=$c timeout 300 python bench.py --workload decompress --log2n 20 --steps 5 --warmup 2 --passes 8 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('decompress 2^20 C=$c: %.1f M/s  ms/pass %.3f kernel_ms %.3f' % (d['value']/1e6, d['config']['ms_per_pass'], r['kernel_ms']))" >> gpurun_out/cmid.txt
  JJ_NORM_C_MID=$c timeout 300 python bench.py --workload fixedbase --log2n 20 --steps 5 --warmup 2 --passes 8 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('fixedbase 2^20 normC=$c: %.1f M/s  ms/pass %.3f kernel_ms %.3f tail_ms %.3f' % (d['value']/1e6, d['config']['ms_per_pass'], r['kernel_ms'], r['tail_ms']))" >> gpurun_out/cmid.txt
done
cat gpurun_out/cmid.txt
