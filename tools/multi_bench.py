#!/usr/bin/env python3
"""jj_multi_* (one process, all listed devices, HOST buffers; include/jubjub_hip.h) timed on one GPU listed once and twice:
   python tools/multi_bench.py            -> devices [0] and [0, 0]  (profiles/r4_multi_bench.txt)
   python tools/multi_bench.py 0 1 2 3    -> the devices given
Batches live in page-locked host memory (jj_host_alloc), result buffers are reused; every line is the median of 5 calls and is
checked against the single-context entry point on the same inputs."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jubjub_amd import Engine, MultiEngine  # noqa: E402

GEN_U = 0x62EDCBB8BF3787C88B0F03DDD60A8187CAF55D1B29BF81AFE4B3D35DF1A7ADFE        # generator, reference src/lib.rs:1380-1396
SEED = 0x4A55424A5542


def med(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def main():
    import torch

    lists = [[int(x) for x in sys.argv[1:]]] if len(sys.argv) > 1 else [[0], [0, 0]]
    eng = Engine(0)
    dev = torch.device("cuda", 0)
    base = np.frombuffer(GEN_U.to_bytes(32, "little") + (11).to_bytes(32, "little"), dtype=np.uint8)

    def pinned(t):
        h = eng.host_alloc(tuple(t.shape)); h[...] = t.cpu().numpy(); return h

    n_vb, n_fb, n_dec, n_msm = 1 << 20, 1 << 24, 1 << 23, 1 << 20
    s_fb = pinned(eng.synth_scalars(n_fb, SEED, 0, device=dev))
    p_vb = pinned(eng.random_points(n_vb, SEED ^ 1, 0, subgroup=False, device=dev))
    enc = pinned(eng.compress(eng.random_points(n_dec, SEED ^ 2, 0, subgroup=False, device=dev)))
    o64, ok = eng.host_alloc((n_fb, 64)), eng.host_alloc((n_dec,))
    ref = {}
    for devs in lists:
        m = MultiEngine(devs)
        tab = m.fixedbase_table(base)
        rows = [("varbase 2^20", n_vb, lambda: m.varbase_mul(s_fb[:n_vb], p_vb, out=o64[:n_vb])),
                ("fixedbase 2^24", n_fb, lambda: m.fixedbase_mul(tab, s_fb, out=o64)),
                ("decompress 2^23 (flags 13)", n_dec, lambda: m.decompress(enc, 13, out=(o64[:n_dec], ok))),
                ("msm 2^20", n_msm, lambda: m.msm(s_fb[:n_msm], p_vb))]
        for name, n, fn in rows:
            dt = med(fn)
            got = np.array(fn()[0] if name.startswith("decompress") else fn()).copy()
            if name not in ref:
                ref[name] = got
            same = bool((got == ref[name]).all())
            print("devices %-8s %-28s %8.2f ms  %8.1f M units/s   equal to the first device list: %s" % (devs, name, dt * 1e3, n / dt / 1e6, same), flush=True)
        m.close()
    # the single-context entry points on the same host buffers, for reference
    tab1 = eng.fixedbase_table(base)
    for name, n, fn in [("varbase 2^20", n_vb, lambda: eng.varbase_mul(s_fb[:n_vb], p_vb, out=o64[:n_vb])),
                        ("fixedbase 2^24", n_fb, lambda: eng.fixedbase_mul(tab1, s_fb, out=o64)),
                        ("decompress 2^23 (flags 13)", n_dec, lambda: eng.decompress(enc, 13, out=(o64[:n_dec], ok))),
                        ("msm 2^20", n_msm, lambda: eng.msm(s_fb[:n_msm], p_vb))]:
        dt = med(fn)
        got = np.array(fn()[0] if name.startswith("decompress") else fn()).copy()
        print("single context   %-28s %8.2f ms  %8.1f M units/s   equal: %s" % (name, dt * 1e3, n / dt / 1e6, bool((got == ref[name]).all())), flush=True)


if __name__ == "__main__":
    main()
