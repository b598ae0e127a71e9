rm -f gpurun_out/chunk_ramp.txt
run() { # workload chunk
  JJ_PIPE_CHUNK_LOG2=$2 timeout 300 python bench.py --workload $1 --host-buffers pinned --steps 5 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 chunk=$2 (ramped): %.1f M/s  ratio %.3f  ms/pass %.2f' % (d['value']/1e6, d['host_over_device_resident'], d['roofline']['pcie']['ms_per_pass']))" >> gpurun_out/chunk_ramp.txt
}
for i in 1 2; do for ch in 20 21 22; do run fixedbase $ch; run decompress $ch; done; for ch in 17 18 19; do run varbase $ch; done; done
cat gpurun_out/chunk_ramp.txt
