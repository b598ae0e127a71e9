rm -f gpurun_out/quantum.txt
run() { # workload extra-label env
  env $3 timeout 300 python bench.py --workload $1 --host-buffers pinned --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2: %.1f M/s  ratio %.3f  ms/pass %.2f verified %s' % (d['value']/1e6, d['host_over_device_resident'], d['roofline']['pcie']['ms_per_pass'], d.get('verified')))" >> gpurun_out/quantum.txt
}
for i in 1 2; do run fixedbase "quantised chunk (default)" X=1; run fixedbase "2^21 chunk" JJ_PIPE_CHUNK_LOG2=21; run "fixedbase --compressed" "quantised" X=1; done
run "fixedbase --fb-window 6" "quantised" X=1
run "fixedbase --fb-window 16" "quantised" X=1
cat gpurun_out/quantum.txt
timeout 900 python -m pytest tests/test_gpu_host_path.py -x -q 2>&1 | tail -2
