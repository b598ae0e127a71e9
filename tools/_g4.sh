timeout 900 python -m pytest tests/test_gpu_host_path.py -x -q > gpurun_out/t_host.txt 2>&1; tail -3 gpurun_out/t_host.txt
rm -f gpurun_out/pipe2.txt
run() { # workload streams chunk
  JJ_PIPE_STREAMS=$2 JJ_PIPE_CHUNK_LOG2=$3 timeout 300 python bench.py --workload $1 --host-buffers pinned --steps 3 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 streams=$2 chunk=$3: %.1f M/s  ratio %.3f  ms/pass %.2f' % (d['value']/1e6, d['host_over_device_resident'], d['roofline']['pcie']['ms_per_pass']))" >> gpurun_out/pipe2.txt
}
for st in 1 2; do
  for ch in 19 20; do run fixedbase $st $ch; run decompress $st $ch; done
  for ch in 16 17; do run varbase $st $ch; done
done
run fixedbase 2 18; run decompress 2 18; run decompress 2 21
cat gpurun_out/pipe2.txt
