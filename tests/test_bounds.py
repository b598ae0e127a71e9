"""Machine-checks the lazy-reduction bounds of the device field arithmetic (tools/bounds_check.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import bounds_check  # noqa: E402
import gen_constants  # noqa: E402


def test_curve_formula_bounds():
    _F, _ops, inv, _ld = bounds_check.check_curve(verbose=False)
    # accumulator coordinates are product-class values: limbs 0..7 in [0, 2^29), value inside (-1.2q, q)
    for k in ("u", "v", "z"):
        assert min(inv[k].lo[:-1]) >= 0 and max(inv[k].hi[:-1]) < (1 << 29)
        assert -1.2 * gen_constants.Q < inv[k].vlo and inv[k].vhi < gen_constants.Q


def test_kernel_formula_bounds():
    """normalise, decode, square root, pairing and the quad-lane point operations of jj_kernels.h"""
    bounds_check.check_kernel_formulas(verbose=False)


def test_field_helper_bounds():
    bounds_check.check_field_misc(gen_constants.Q, "Fq", verbose=False)
    bounds_check.check_field_misc(gen_constants.R, "Fr", verbose=False)


def test_constants_header_is_current():
    """jj_constants.h must be exactly what tools/gen_constants.py generates."""
    path = os.path.join(ROOT, "jubjub_amd", "csrc", "jj_constants.h")
    before = open(path).read()
    gen_constants.main()
    after = open(path).read()
    assert before == after


def test_sqrt_dlog_key_is_collision_free():
    """fq_sqrt_fast (jj_kernels.h) looks the 8-bit discrete logs up by 16 bits of the canonical value: limb SQRT_KEY_LIMB."""
    import re
    src = open(os.path.join(ROOT, "jubjub_amd", "csrc", "jj_kernels.h")).read()
    limb = int(re.search(r"constexpr int SQRT_KEY_LIMB = (\d+);", src).group(1))
    q = gen_constants.Q
    gam = pow(pow(7, (q - 1) >> 32, q), 1 << 24, q)
    keys = {(pow(gam, k, q) >> (29 * limb)) & 0xFFFF for k in range(256)}
    assert len(keys) == 256
