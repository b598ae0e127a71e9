rm -f gpurun_out/ramp.txt
run() { # workload ramp
  JJ_PIPE_RAMP=$2 timeout 300 python bench.py --workload $1 --host-buffers pinned --steps 5 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 ramp=$2: %.1f M/s  ratio %.3f  ms/pass %.2f' % (d['value']/1e6, d['host_over_device_resident'], d['roofline']['pcie']['ms_per_pass']))" >> gpurun_out/ramp.txt
}
for i in 1 2; do for wl in varbase fixedbase decompress; do run $wl 0; run $wl 1; done; done
cat gpurun_out/ramp.txt
timeout 900 python -m pytest tests/test_gpu_host_path.py -x -q 2>&1 | tail -2
