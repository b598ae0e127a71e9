#!/usr/bin/env python3
"""
Transcribes the known-answer DATA (numbers only) held by the reference's own tests into
tests/golden/reference_vectors.json.  Run in the build container where /root/reference
exists:   python tests/golden/make_golden.py

It parses numeric literals out of fixed line ranges of the reference's .rs files (cited
per entry) — no reference source text is stored, only the integers/bytes.  The GPU box has
no /root/reference; tests read the committed JSON.

A second part (oracle-generated vectors: random scalar-mul / decompress / MSM cases) is
produced by tests/golden/make_oracle_vectors.py from the pinned oracle.
"""
import json
import os
import re
import sys

REF = os.environ.get("JJ_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

NUM = re.compile(r"0x[0-9a-fA-F_]+|\b\d[\d_]*\b")


def lines(path, lo, hi):
    with open(os.path.join(REF, path)) as f:
        all_lines = f.readlines()
    return "".join(all_lines[lo - 1 : hi])


def strip_comments(text):
    return re.sub(r"//.*", "", text)


def nums(path, lo, hi):
    """All integer literals in the line range (comments stripped), as Python ints."""
    out = []
    for tok in NUM.findall(strip_comments(lines(path, lo, hi))):
        tok = tok.replace("_", "")
        out.append(int(tok, 16) if tok.startswith("0x") else int(tok))
    return out


def hexnums(path, lo, hi):
    out = []
    for tok in re.findall(r"0x[0-9a-fA-F_]+", strip_comments(lines(path, lo, hi))):
        out.append(int(tok.replace("_", ""), 16))
    return out


def limbs_hex(l):
    return ["0x%016x" % x for x in l]


def chunks(l, n):
    assert len(l) % n == 0, (len(l), n)
    return [l[i : i + n] for i in range(0, len(l), n)]


def main():
    g = {}
    L = "src/lib.rs"
    F = "src/fr.rs"

    # ---- curve constants -------------------------------------------------------------
    g["FR_MODULUS_BYTES"] = {"src": "src/lib.rs:73-76", "bytes": nums(L, 74, 75)}
    g["EDWARDS_D_raw"] = {"src": "src/lib.rs:399-404", "limbs": limbs_hex(hexnums(L, 400, 403))}
    g["EDWARDS_D2_raw"] = {"src": "src/lib.rs:407-412", "limbs": limbs_hex(hexnums(L, 408, 411))}
    g["GENERATOR_raw"] = {
        "src": "src/lib.rs:1383-1394",
        "u": limbs_hex(hexnums(L, 1384, 1387)),
        "v": limbs_hex(hexnums(L, 1390, 1393)),
    }
    fg = hexnums(L, 1579, 1585)
    g["FULL_GENERATOR_raw"] = {"src": "src/lib.rs:1578-1586", "u": limbs_hex(fg[:4]), "v": limbs_hex(fg[4:8])}

    # test point used by test_assoc / test_batch_normalize / test_mul_consistency
    tp = hexnums(L, 1507, 1518)
    g["TEST_POINT_raw"] = {"src": "src/lib.rs:1506-1519", "u": limbs_hex(tp[:4]), "v": limbs_hex(tp[4:8])}
    assert hexnums(L, 1532, 1543) == tp and hexnums(L, 1778, 1789) == tp
    g["test_assoc_scalars"] = {"src": "src/lib.rs:1523-1526", "a": 1000, "b": 3938}

    # EIGHT_TORSION: 8 affine points, from_raw canonical limbs (some written as 0x0 shorthand)
    et_text = strip_comments(lines(L, 1589, 1677))
    pts = []
    # split on AffinePoint::from_raw_unchecked( ... ),
    for blk in et_text.split("AffinePoint::from_raw_unchecked(")[1:]:
        raws = blk.split("Fq::from_raw(")[1:3]
        coords = []
        for r in raws:
            inner = r[: r.index("])") + 1]
            vals = [int(t.replace("_", ""), 16) for t in re.findall(r"0x[0-9a-fA-F_]+", inner)]
            assert len(vals) == 4, vals
            coords.append(limbs_hex(vals))
        pts.append({"u": coords[0], "v": coords[1]})
    assert len(pts) == 8
    g["EIGHT_TORSION_raw"] = {"src": "src/lib.rs:1589-1677", "points": pts}

    # test_mul_consistency Fr triple (Montgomery limbs) a*b = c
    tri = hexnums(L, 1758, 1775)
    g["fr_mul_consistency_mont"] = {
        "src": "src/lib.rs:1758-1776",
        "a": limbs_hex(tri[0:4]),
        "b": limbs_hex(tri[4:8]),
        "c": limbs_hex(tri[8:12]),
    }

    # test_serialization_consistency: 16 encodings of k*(8G), k=1..16
    enc = nums(L, 1811, 1876)
    g["serialization_16"] = {"src": "src/lib.rs:1811-1876", "encodings": chunks(enc, 32)}
    assert len(g["serialization_16"]["encodings"]) == 16

    # test_zip_216 non-canonical encodings
    z = hexnums(L, 1896, 1906)
    g["zip216_noncanonical"] = {"src": "src/lib.rs:1894-1907", "encodings": chunks(z, 32)}
    assert len(g["zip216_noncanonical"]["encodings"]) == 2

    # WnafGroup recommendations
    g["wnaf_recommendations"] = {"src": "src/lib.rs:1322-1323", "table": nums(L, 1323, 1323)}

    # ---- Fr constants and vectors ---------------------------------------------------------
    def fr4(lo, hi):
        v = hexnums(F, lo, hi)
        assert len(v) == 4, (lo, hi, v)
        return limbs_hex(v)

    g["fr"] = {
        "MODULUS": {"src": "src/fr.rs:77-82", "limbs": fr4(78, 81)},
        "MODULUS_LIMBS_32": {"src": "src/fr.rs:86-95", "limbs": ["0x%08x" % x for x in hexnums(F, 87, 94)]},
        "MODULUS_BITS": {"src": "src/fr.rs:98", "value": 252},
        "TWO_INV_mont": {"src": "src/fr.rs:101-106", "limbs": fr4(102, 105)},
        "GENERATOR_mont": {"src": "src/fr.rs:109-114", "limbs": fr4(110, 113)},
        "S": {"src": "src/fr.rs:117", "value": nums(F, 117, 117)[-1]},
        "ROOT_OF_UNITY_mont": {"src": "src/fr.rs:120-125", "limbs": fr4(121, 124)},
        "DELTA_mont": {"src": "src/fr.rs:132-137", "limbs": fr4(133, 136)},
        "INV": {"src": "src/fr.rs:214", "value": "0x%016x" % hexnums(F, 214, 214)[0]},
        "R_mont": {"src": "src/fr.rs:217-222", "limbs": fr4(218, 221)},
        "R2_mont": {"src": "src/fr.rs:225-230", "limbs": fr4(226, 229)},
        "R3_mont": {"src": "src/fr.rs:233-238", "limbs": fr4(234, 237)},
        "SQRT_EXP": {"src": "src/fr.rs:388-393", "limbs": fr4(389, 392)},
        "DELTA_T_EXP": {"src": "src/fr.rs:803-808", "limbs": fr4(804, 807)},
        "LARGEST_mont": {"src": "src/fr.rs:1045-1050", "limbs": fr4(1046, 1049)},
        "R_MINUS_2": {"src": "src/fr.rs:1179-1184", "limbs": fr4(1180, 1183)},
    }
    # test_to_bytes / test_from_bytes golden byte strings
    g["fr"]["to_bytes"] = {
        "src": "src/fr.rs:856-888",
        "zero": nums(F, 860, 861),
        "one": nums(F, 868, 869),
        "R2": nums(F, 876, 877),
        "neg_one": nums(F, 884, 885),
    }
    g["fr"]["from_bytes_invalid"] = {
        "src": "src/fr.rs:928-960",
        "cases": [nums(F, 931, 932), nums(F, 940, 941), nums(F, 948, 949), nums(F, 956, 957)],
    }
    g["fr"]["from_bytes_wide"] = {
        "src": "src/fr.rs:1000-1034",
        "r2_input": nums(F, 1004, 1006),
        "neg_one_input": nums(F, 1016, 1018),
        "max_output_mont": fr4(1027, 1030),
    }
    g["fr"]["test_addition"] = {"src": "src/fr.rs:1053-1071", "largest_plus_largest_mont": fr4(1060, 1063)}
    g["fr"]["test_sqrt"] = {"src": "src/fr.rs:1205-1227", "start_mont": fr4(1208, 1211), "none_count": 47, "iters": 100}
    g["fr"]["test_from_raw"] = {"src": "src/fr.rs:1230-1244", "expect_mont_of_all_ones": fr4(1233, 1236)}
    g["fr"]["debug_R2"] = {"src": "src/fr.rs:838-841", "hex": re.findall(r'"(0x[0-9a-f]+)"', lines(F, 838, 841))[0]}

    # ---- SafeCurves evidence parameter files (independent cross-check) -------------------------
    ev = {}
    for name in ["p", "l", "d", "a", "x0", "y0", "x1", "y1"]:
        with open(os.path.join(REF, "doc/evidence", name)) as f:
            ev[name] = f.read().strip()
    g["evidence"] = {"src": "doc/evidence/{p,l,d,a,x0,y0,x1,y1}", **ev}

    # README modulus strings
    readme = open(os.path.join(REF, "README.md")).read()
    g["readme_hex"] = {"src": "README.md", "values": sorted(set(re.findall(r"0x[0-9a-f]{60,64}", readme)))}

    # black-box test parameters (tests/common.rs)
    g["blackbox"] = {"src": "tests/common.rs:5-9", "checks": 2000, "xorshift_seed": list(range(16))}

    out = os.path.join(HERE, "reference_vectors.json")
    with open(out, "w") as f:
        json.dump(g, f, indent=1)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    sys.exit(main())
