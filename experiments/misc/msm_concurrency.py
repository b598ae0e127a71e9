import sys, time, threading
sys.path.insert(0, '.')
import torch
from jubjub_amd import Engine
dev = torch.device('cuda', 0)
n = 1 << 20
engs = [Engine(0) for _ in range(3)]
data = []
for e in engs:
    s = e.synth_scalars(n, 7, 0, device=dev); p = e.random_points(n, 9, 0, device=dev); data.append((s, p))
torch.cuda.synchronize()
def loop(e, s, p, iters, streams=None):
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        for _ in range(iters):
            e.msm(s, p)
    st.synchronize()
for k in (1, 2, 3):
    for w in range(2):
        th = [threading.Thread(target=loop, args=(engs[i], data[i][0], data[i][1], 20)) for i in range(k)]
        t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("concurrent contexts: %d  -> %.3f ms per MSM (aggregate %.1f M terms/s)" % (k, dt / (20 * k) * 1e3, 20 * k * n / dt / 1e6))
