set -x
timeout 900 python -m pytest tests/test_gpu_host_path.py -x -q 2>&1 | tail -15
for wl in fixedbase decompress; do for ch in 19 20 21 22; do
  JJ_PIPE_CHUNK_LOG2=$ch timeout 300 python bench.py --workload $wl --host-buffers pinned --steps 3 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl chunk $ch: %.1f M/s  ratio %.3f  ms/pass %.2f' % (d['value']/1e6, d['host_over_device_resident'], d['roofline']['pcie']['ms_per_pass']))" >> gpurun_out/chunk_sweep.txt
done; done
for ch in 15 16 17 18; do
  JJ_PIPE_CHUNK_LOG2=$ch timeout 300 python bench.py --workload varbase --host-buffers pinned --steps 3 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('varbase chunk $ch: %.1f M/s  ratio %.3f  ms/pass %.2f' % (d['value']/1e6, d['host_over_device_resident'], d['roofline']['pcie']['ms_per_pass']))" >> gpurun_out/chunk_sweep.txt
done
cat gpurun_out/chunk_sweep.txt
