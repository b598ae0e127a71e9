bash tools/fixedbase_select_pmc.sh > gpurun_out/r4_fixedbase_select_pmc.txt 2>&1; cat gpurun_out/r4_fixedbase_select_pmc.txt
(./experiments/lds_probe/energy_probe; ./experiments/lds_probe/probe) > gpurun_out/r4_issue_energy_probe.txt 2>&1; tail -6 gpurun_out/r4_issue_energy_probe.txt
timeout 400 python tests/soak_host.py 150 > gpurun_out/r4_soak_host.txt 2>&1; grep -v amdgpu gpurun_out/r4_soak_host.txt | tail -4
for m in fresh; do
  python bench.py --workload fixedbase --host-buffers fresh --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['roofline']['pcie']; print('fixedbase fresh (bounce): %.2f ms/pass, inside the call %.2f ms, caller alloc+free %.2f ms' % (p['ms_per_pass'], p['call_ms_per_pass'], p['caller_alloc_free_ms_per_pass']))"
  JJ_PIPE_PAGEABLE=register python bench.py --workload fixedbase --host-buffers fresh --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['roofline']['pcie']; print('fixedbase fresh (register): %.2f ms/pass, inside the call %.2f ms, caller alloc+free %.2f ms' % (p['ms_per_pass'], p['call_ms_per_pass'], p['caller_alloc_free_ms_per_pass']))"
done
