#!/usr/bin/env python3
"""
Generates tests/golden/oracle_vectors.json: small committed input/output fixtures produced by the pinned
Python oracle (oracle/jubjub_ref.py) — random variable-base / fixed-base scalar-muls, decompression cases
(valid, invalid, non-canonical, small-order, off-subgroup) and MSM instances.  Data only.
Run: python tests/golden/make_oracle_vectors.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import jubjub_ref as J  # noqa: E402

Q, R = J.Q, J.R_MOD
hx = lambda b: bytes(b).hex()
i32 = lambda x: int(x).to_bytes(32, "little")
pt = lambda p: i32(p[0]) + i32(p[1])


def main():
    rng = random.Random(0x4A55424A5542)
    out = {"source": "oracle/jubjub_ref.py (pinned by tests/test_oracle_golden.py)", "format": "hex of little-endian wire bytes"}
    pts = [J.scalar_mul_fast(J.GENERATOR, rng.randrange(1, 8 * R)) for _ in range(24)]
    tors = [J.ext_to_affine(J.ext_multiply(J.affine_to_extended(J.scalar_mul_fast(J.GENERATOR, k)), J.FR_MODULUS_BYTES)) for k in range(1, 9)]
    scal = [0, 1, R - 1, R, (1 << 252) - 1, (0xF << 252) | 7] + [rng.randrange(1 << 256) for _ in range(18)]
    vb = []
    for k, p in zip(scal, pts):
        e = J.ext_multiply(J.affine_to_extended(p), i32(k))
        vb.append({"scalar": hx(i32(k)), "point": hx(pt(p)), "out": hx(pt(J.ext_to_affine(e))),
                   "ext": hx(b"".join(i32(c) for c in e))})
    for k, p in zip(scal[:8], tors):
        e = J.ext_multiply(J.affine_to_extended(p), i32(k))
        vb.append({"scalar": hx(i32(k)), "point": hx(pt(p)), "out": hx(pt(J.ext_to_affine(e))),
                   "ext": hx(b"".join(i32(c) for c in e))})
    out["varbase"] = vb
    base = pts[0]
    out["fixedbase"] = {"base": hx(pt(base)), "cases": [
        {"scalar": hx(i32(k)), "out": hx(pt(J.ext_to_affine(J.affine_niels_multiply(J.affine_to_niels(base), i32(k)))))} for k in scal]}
    encs = [J.affine_to_bytes(p) for p in pts[:10]] + [J.affine_to_bytes(t) for t in tors]
    encs += [bytes(rng.randrange(256) for _ in range(32)) for _ in range(24)]
    encs += [i32(Q), i32(Q - 1), i32((Q - 1) | (1 << 255)), i32(1 | (1 << 255)), i32(0), i32((1 << 256) - 1)]
    dec = []
    for e in encs:
        row = {"in": hx(e)}
        for flags in (0, 1, 3, 5, 9, 15):
            p, ok = J.affine_from_bytes(e, zip216=bool(flags & 1))
            if ok:
                ep = J.affine_to_extended(p)
                if (flags & 2) and not J.ext_is_torsion_free(ep):
                    ok = 0
                if (flags & 4) and J.ext_is_small_order(ep):
                    ok = 0
                if ok and (flags & 8):
                    p = J.ext_to_affine(J.ext_mul_by_cofactor(ep))
            row["f%d" % flags] = {"ok": ok, "out": hx(pt(p if ok else (0, 0)))}
        dec.append(row)
    out["decompress"] = dec
    msm = []
    for n in (1, 4, 16):
        ks = [rng.randrange(1 << 256) for _ in range(n)]
        ps = [pts[rng.randrange(len(pts))] for _ in range(n)]
        msm.append({"scalars": [hx(i32(k)) for k in ks], "points": [hx(pt(p)) for p in ps],
                    "out": hx(pt(J.ext_to_affine(J.msm([i32(k) for k in ks], ps))))})
    out["msm"] = msm
    path = os.path.join(HERE, "oracle_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
