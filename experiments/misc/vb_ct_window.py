import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from jubjub_amd import Engine
n = 1 << 20
base = Engine(0)
S = base.synth_scalars(n, 7, 0, device="cuda:0"); P = base.random_points(n, 7, 0, subgroup=False, device="cuda:0")
want = base.varbase_mul_vartime(S, P)
def t(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts)//2]
d = t(lambda: base.varbase_mul_vartime(S, P))
print("table ladder (jj_varbase_mul_vartime)  %.3f ms  %.1f M/s" % (d*1e3, n/d/1e6))
for w in (2, 3):
    e = Engine(0, options={"vb_ct_window": w})          # round 6: a context option (was JJ_VB_CT_WINDOW)
    got = e.varbase_mul_ct(S, P)
    c = t(lambda: e.varbase_mul_ct(S, P))
    print("ct window %s     %.3f ms  %.1f M/s  ratio %.3f  equal %s" % (w, c*1e3, n/c/1e6, d/c, bool(torch.equal(got, want))))
    e.close()
