"""
Runs the C++ replay of the reference crate's tests (tests/cpp/test_reference_suite.cpp, built by
__graft_entry__.build()) through include/jubjub_hip.hpp on the GPU.
"""
import json
import os
import subprocess

import pytest

from conftest import GOLDEN_DIR, ROOT, limbs

BIN = os.path.join(ROOT, "tests", "cpp", "test_reference_suite")


def write_vectors(path):
    from oracle import jubjub_ref as J   # only for Montgomery -> canonical conversion of the golden Fr limbs

    g = json.load(open(os.path.join(GOLDEN_DIR, "reference_vectors.json")))
    hx = lambda b: bytes(b).hex()
    i32 = lambda x: int(x).to_bytes(32, "little")
    pt = lambda d: i32(limbs(d["u"]) % J.Q) + i32(limbs(d["v"]) % J.Q)
    fr = lambda l: i32(J.FR.from_mont_limbs([int(x, 16) for x in l]))
    lines = {
        "serialization_16": [hx(e) for e in g["serialization_16"]["encodings"]],
        "zip216_noncanonical": [hx(e) for e in g["zip216_noncanonical"]["encodings"]],
        "FR_MODULUS_BYTES": [hx(g["FR_MODULUS_BYTES"]["bytes"])],
        "EIGHT_TORSION": [hx(pt(p)) for p in g["EIGHT_TORSION_raw"]["points"]],
        "TEST_POINT": [hx(pt(g["TEST_POINT_raw"]))],
        "fr_mul_a": [hx(fr(g["fr_mul_consistency_mont"]["a"]))],
        "fr_mul_b": [hx(fr(g["fr_mul_consistency_mont"]["b"]))],
        "fr_mul_c": [hx(fr(g["fr_mul_consistency_mont"]["c"]))],
        "fr_neg_one": [hx(g["fr"]["to_bytes"]["neg_one"])],
        "fr_from_bytes_invalid": [hx(e) for e in g["fr"]["from_bytes_invalid"]["cases"]],
        "fr_sqrt_start": [hx(fr(g["fr"]["test_sqrt"]["start_mont"]))],
        "fr_wide_max_in": ["ff" * 64],
        "fr_wide_max_out": [hx(fr(g["fr"]["from_bytes_wide"]["max_output_mont"]))],
    }
    with open(path, "w") as f:
        for k, v in lines.items():
            f.write(k + " " + " ".join(v) + "\n")


def test_cpp_binary_is_built():
    """CPU-side check: the C++ host mirror compiles against the C ABI (done by __graft_entry__.build())."""
    import __graft_entry__ as ge

    ge.build_cpp_tests()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_reference_suite_through_cpp_host(tmp_path):
    import __graft_entry__ as ge

    ge.build_cpp_tests()
    vec = tmp_path / "vectors.txt"
    write_vectors(str(vec))
    r = subprocess.run([BIN, str(vec)], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    print(r.stderr)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "ALL PASSED" in r.stdout


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    """examples/scalar_mul.c prints the 16 golden encodings of the reference's test_serialization_consistency."""
    lib = os.path.join(ROOT, "jubjub_amd", "lib")
    out = tmp_path / "scalar_mul"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "scalar_mul.c"),
                           "-L", lib, "-ljubjub_hip", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-o", str(out)])
    r = subprocess.run([str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    g = json.load(open(os.path.join(GOLDEN_DIR, "reference_vectors.json")))
    want = [bytes(e).hex() for e in g["serialization_16"]["encodings"]]
    got = [line.split("= ")[1].strip() for line in r.stdout.splitlines() if "*(8G) =" in line]
    assert got == want
    assert "msm ok, begin/finish ok, partial+combine ok" in r.stdout, r.stdout
