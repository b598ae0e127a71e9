timeout 1200 python -m pytest tests/test_gpu_host_path.py -x -q > gpurun_out/t_host.txt 2>&1; tail -3 gpurun_out/t_host.txt
python - <<'PY' 2>&1 | grep -v amdgpu
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from jubjub_amd import Engine
eng = Engine(0); dev = torch.device("cuda", 0)
n = 1 << 20
s = eng.synth_scalars(n, 5, 0, device=dev); p = eng.random_points(n, 6, 0, subgroup=False, device=dev)
hs, hp = s.cpu().numpy().copy(), p.cpu().numpy().copy()
ps, pp = eng.host_alloc((n, 32)), eng.host_alloc((n, 64)); ps[...] = hs; pp[...] = hp
def med(fn):
    fn(); ts = []
    for _ in range(9):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[4] * 1e3
print("msm 2^20: device-resident %.2f ms, page-locked host arrays %.2f ms, pageable host arrays %.2f ms" % (med(lambda: eng.msm(s, p)), med(lambda: eng.msm(ps, pp)), med(lambda: eng.msm(hs, hp))))
PY
