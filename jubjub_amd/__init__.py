"""
jubjub_amd — MI355X-native batched Jubjub scalar-multiplication engine.

The product is the C-ABI library jubjub_amd/lib/libjubjub_hip.so (sources in jubjub_amd/csrc, interface in
include/jubjub_hip.h).  This package is the Python host side: `Engine` (typed wrapper over the C ABI) and
`group` (batch mirrors of the reference crate's public types and method names).
"""
from .engine import (Engine, MultiEngine, FixedBaseTable, JubjubError, FLAG_ZIP216, FLAG_TORSION_FREE, FLAG_NOT_SMALL_ORDER,
                     FLAG_CLEAR_COFACTOR)

__all__ = ["Engine", "MultiEngine", "FixedBaseTable", "JubjubError", "FLAG_ZIP216", "FLAG_TORSION_FREE", "FLAG_NOT_SMALL_ORDER",
           "FLAG_CLEAR_COFACTOR"]
