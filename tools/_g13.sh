./experiments/lds_probe/probe > gpurun_out/lds_probe.txt 2>&1; cat gpurun_out/lds_probe.txt
rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -v "^$" | head -30 > gpurun_out/smi_idle.txt
(for i in $(seq 1 40); do rocm-smi --showpower --showclocks -t 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ' '; echo; sleep 0.5; done) > gpurun_out/smi_load.txt &
python bench.py --workload fixedbase --steps 60 --warmup 3 --no-cpu-baseline --no-verify > /dev/null 2>&1
python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-verify --no-extras > /dev/null 2>&1
wait
cat gpurun_out/smi_idle.txt; cat gpurun_out/smi_load.txt | head -50
