#!/usr/bin/env python3
"""Adversarial MSM inputs at full size: all scalars equal (one bucket per window holds every term), half equal, and scalars
that differ only in their low 16 bits.  Prints the time per MSM next to the uniform case."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from jubjub_amd import Engine  # noqa: E402

eng = Engine(0)
n = 1 << 20
S = eng.synth_scalars(n, 7, 0, device="cuda:0")
P = eng.random_points(n, 7, 0, subgroup=False, device="cuda:0")


def timed(name, s):
    eng.msm(s, P)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = eng.msm(s, P)
    torch.cuda.synchronize()
    print("%-46s %8.3f ms per MSM" % (name, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    return out


timed("uniform scalars", S)
eq = S[:1].repeat(n, 1).contiguous()
timed("all scalars equal", eq)
half = S.clone(); half[: n // 2] = S[0]
timed("half of the scalars equal", half)
low = S[:1].repeat(n, 1).contiguous(); low[:, :2] = S[:, :2]
timed("scalars differ in their low 16 bits only", low)
zero = torch.zeros_like(S)
timed("all scalars zero", zero)
