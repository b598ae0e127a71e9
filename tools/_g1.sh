set -x
./tools/pcie_probe > gpurun_out/r4_pcie_probe_raw.txt 2>&1
for hb in pinned pageable; do
  for wl in fixedbase decompress varbase; do
    JJ_PIPE_DEBUG=1 timeout 600 python bench.py --workload $wl --host-buffers $hb --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/hb_${wl}_${hb}.json 2> gpurun_out/hb_${wl}_${hb}.err
    tail -3 gpurun_out/hb_${wl}_${hb}.err
  done
done
