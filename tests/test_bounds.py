"""Machine-checks the lazy-reduction bounds of the device field arithmetic (tools/bounds_check.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import bounds_check  # noqa: E402
import gen_constants  # noqa: E402


def test_curve_formula_bounds():
    inv = bounds_check.check_curve(verbose=False)
    assert inv["u"][1] <= 2 * gen_constants.Q


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
