rm -f gpurun_out/size_eff.txt
for wl in fixedbase decompress; do for l in 19 20 21 22 23; do
  timeout 300 python bench.py --workload $wl --log2n $l --steps 5 --warmup 2 --passes 8 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl 2^$l device-resident: %.1f M/s  ms/pass %.3f kernel_ms %.3f tail_ms %.3f' % (d['value']/1e6, d['config']['ms_per_pass'], r['kernel_ms'], r['tail_ms']))" >> gpurun_out/size_eff.txt
done; done
cat gpurun_out/size_eff.txt
