./experiments/lds_probe/energy_probe 2>&1 | tee gpurun_out/energy_probe.txt
