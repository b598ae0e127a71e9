"""
Host-side mirror of the zkcrypto/jubjub public interface (names, argument meaning, error behaviour), lifted
from one element to a batch and backed by the MI355X engine through the C ABI.

    reference (src/lib.rs, src/fr.rs)                          here
    AffinePoint / ExtendedPoint / SubgroupPoint                 Points      (batch of affine encodings, 64 B each)
    Fr / Fq                                                     Fr / Fq     (batch of canonical 32-byte encodings)
    p * k, p + q, p - q, -p, p.double(), p.mul_by_cofactor()    same operators / method names
    AffinePoint::from_bytes / batch_from_bytes / to_bytes       Points.from_bytes / batch_from_bytes / to_bytes
    is_identity / is_small_order / is_torsion_free / ...        same names, return a uint8 Choice per element
    batch_normalize, iter.sum()                                 batch_normalize, Points.sum()

CtOption<T> becomes (value, is_some) with is_some a uint8 array; None entries are zeroed like the C ABI.
"""
import numpy as np

from .engine import (Engine, FLAG_CLEAR_COFACTOR, FLAG_NOT_SMALL_ORDER, FLAG_TORSION_FREE, FLAG_ZIP216)

_GEN_U = 0x62EDCBB8BF3787C88B0F03DDD60A8187CAF55D1B29BF81AFE4B3D35DF1A7ADFE   # reference src/lib.rs:1383-1388
FR_MODULUS_BYTES = bytes([183, 44, 247, 214, 94, 14, 151, 208, 130, 16, 200, 204, 147, 32, 104, 166, 0, 59, 52, 1, 1,
                          59, 103, 6, 169, 175, 51, 101, 234, 180, 125, 14])                      # reference src/lib.rs:73-76


_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def random_stream_seeds(seed):
    """seeds of the two 32-byte halves Field::random feeds to from_bytes_wide: splitmix64(seed ^ tag), tags "jjFrndLo" / "jjFrndHi"
    (little-endian ASCII).  include/jubjub_hip.hpp FieldBatch::random uses the same two streams."""
    return _splitmix64((seed ^ 0x6F4C646E72466A6A) & _M64), _splitmix64((seed ^ 0x6948646E72466A6A) & _M64)


def _as_rows(x, width):
    if type(x).__module__.startswith("torch"):
        return x.reshape(-1, width)
    return np.ascontiguousarray(x, dtype=np.uint8).reshape(-1, width)


class _Field:
    _name = None

    def __init__(self, engine: Engine, data):
        self.engine = engine
        self.data = _as_rows(data, 32)

    def __len__(self):
        return self.data.shape[0]

    @classmethod
    def from_u64(cls, engine, values):                      # From<u64> (src/fr.rs:42-46)
        a = np.zeros((len(values), 32), np.uint8)
        for i, v in enumerate(values):
            a[i, :8] = np.frombuffer(int(v).to_bytes(8, "little"), np.uint8)
        return cls(engine, a)

    @classmethod
    def from_bytes(cls, engine, data):                      # src/fr.rs:268-292 -> (value, is_some)
        out, ok = engine.field_unary_ok(cls._name, "from_bytes", _as_rows(data, 32))
        return cls(engine, out), ok

    @classmethod
    def from_bytes_wide(cls, engine, data):                 # src/fr.rs:312-343
        return cls(engine, engine.from_bytes_wide(cls._name, _as_rows(data, 64)))

    def to_bytes(self):                                     # src/fr.rs:296-308
        return self.data

    def _bin(self, op, other):
        if len(other) != len(self):
            raise ValueError("length mismatch")
        return type(self)(self.engine, self.engine.field_binary(self._name, op, self.data, other.data))

    def __add__(self, o): return self._bin("add", o)
    def __sub__(self, o): return self._bin("sub", o)
    def __mul__(self, o):
        if isinstance(o, Points):
            return o * self
        return self._bin("mul", o)
    def __neg__(self): return type(self)(self.engine, self.engine.field_unary(self._name, "neg", self.data))
    def square(self): return type(self)(self.engine, self.engine.field_unary(self._name, "square", self.data))
    def double(self): return type(self)(self.engine, self.engine.field_unary(self._name, "double", self.data))

    def invert(self):                                       # src/fr.rs:438-540 -> (value, is_some)
        out, ok = self.engine.field_unary_ok(self._name, "invert", self.data)
        return type(self)(self.engine, out), ok

    def sqrt(self):                                         # src/fr.rs:384-399 -> (value, is_some)
        out, ok = self.engine.field_unary_ok(self._name, "sqrt", self.data)
        return type(self)(self.engine, out), ok

    def to_le_bits(self):                                   # PrimeFieldBits::to_le_bits (src/fr.rs:746-773): n x 256 bytes of 0/1
        return self.engine.to_le_bits(self._name, self.data)

    @classmethod
    def random(cls, engine, n, seed, first_index=0):
        """Field::random (src/fr.rs:684-688; Fq = bls12_381::Scalar does the same): 64 PRNG bytes through from_bytes_wide, for
        BOTH fields (no truncate-and-subtract bias).  The two 32-byte halves of unit i come from two decorrelated counter
        streams of the library's splitmix64 generator (the seed hashed with a domain tag per half: random_stream_seeds).
        The generator is counter-based and reproducible: it makes TEST DATA, it is not a source of secret scalars."""
        lo_seed, hi_seed = random_stream_seeds(seed)
        wide = np.concatenate([engine.synth_bytes32(n, lo_seed, first_index), engine.synth_bytes32(n, hi_seed, first_index)], axis=1)
        return cls(engine, engine.from_bytes_wide(cls._name, wide))

    def __eq__(self, o):
        return bool((self.data == o.data).all())


class Fr(_Field):
    _name = "fr"


class Fq(_Field):
    _name = "fq"


class Points:
    """A batch of curve points in affine (u, v) wire form — plays AffinePoint, ExtendedPoint and SubgroupPoint."""

    def __init__(self, engine: Engine, data):
        self.engine = engine
        self.data = _as_rows(data, 64)

    def __len__(self):
        return self.data.shape[0]

    @classmethod
    def identity(cls, engine, n=1):                         # src/lib.rs:416-421
        a = np.zeros((n, 64), np.uint8)
        a[:, 32] = 1
        return cls(engine, a)

    @classmethod
    def generator(cls, engine, n=1):                        # src/lib.rs:1380-1396
        g = np.frombuffer(_GEN_U.to_bytes(32, "little") + (11).to_bytes(32, "little"), np.uint8)
        return cls(engine, np.tile(g, (n, 1)))

    @classmethod
    def random(cls, engine, n, seed, first_index=0, subgroup=False):   # Group::random (src/lib.rs:1244-1267; SubgroupPoint: 1290-1298)
        return cls(engine, engine.random_points(n, seed, first_index, subgroup=subgroup))

    @classmethod
    def from_raw_unchecked(cls, engine, uv):                # src/lib.rs:662-664
        return cls(engine, uv)

    @classmethod
    def from_bytes(cls, engine, enc, zip216=True, subgroup=False, not_small_order=False, clear_cofactor=False):
        """AffinePoint::from_bytes (src/lib.rs:469-471); zip216=False: from_bytes_pre_zip216_compatibility (488-490);
        subgroup=True: SubgroupPoint::from_bytes (1427-1429).  Returns (points, is_some)."""
        flags = (FLAG_ZIP216 if zip216 else 0) | (FLAG_TORSION_FREE if subgroup else 0) | \
                (FLAG_NOT_SMALL_ORDER if not_small_order else 0) | (FLAG_CLEAR_COFACTOR if clear_cofactor else 0)
        out, ok = engine.decompress(_as_rows(enc, 32), flags)
        return cls(engine, out), ok

    @classmethod
    def batch_from_bytes(cls, engine, enc):                 # src/lib.rs:541-627
        return cls.from_bytes(engine, enc)

    def to_bytes(self): return self.engine.compress(self.data)                               # src/lib.rs:455-464
    def get_u(self): return Fq(self.engine, self.data[:, :32])                               # src/lib.rs:630-632
    def get_v(self): return Fq(self.engine, self.data[:, 32:])                               # src/lib.rs:635-637
    def to_niels(self): return self.engine.to_niels(self.data)                               # src/lib.rs:652-658
    def double(self): return Points(self.engine, self.engine.point_double(self.data))        # src/lib.rs:739-828
    def mul_by_cofactor(self): return Points(self.engine, self.engine.mul_by_cofactor(self.data))  # src/lib.rs:722-724
    clear_cofactor = mul_by_cofactor                                                         # src/lib.rs:1343-1345
    def __neg__(self): return Points(self.engine, self.engine.point_neg(self.data))          # src/lib.rs:92-104
    def __add__(self, o): return Points(self.engine, self.engine.point_add(self.data, o.data))   # src/lib.rs:1012-1019
    def __sub__(self, o): return Points(self.engine, self.engine.point_sub(self.data, o.data))   # src/lib.rs:1021-1028
    def is_identity(self): return self.engine.predicate("is_identity", self.data)            # src/lib.rs:691-696
    def is_small_order(self): return self.engine.predicate("is_small_order", self.data)      # src/lib.rs:699-705
    def is_torsion_free(self): return self.engine.predicate("is_torsion_free", self.data)    # src/lib.rs:709-711
    def is_prime_order(self): return self.engine.predicate("is_prime_order", self.data)      # src/lib.rs:717-719
    def is_on_curve(self): return self.engine.predicate("is_on_curve", self.data)            # src/lib.rs:670-675

    def multiply_bits(self, by):                            # src/lib.rs:357-385, 831-833: raw 32-byte patterns
        by = _as_rows(by, 32)
        if by.shape[0] != len(self):
            raise ValueError("length mismatch")            # cf. the assert at src/lib.rs:841
        return Points(self.engine, self.engine.varbase_mul(by, self.data))

    def __mul__(self, k):                                   # `&ExtendedPoint * &Fr` src/lib.rs:873-879
        return self.multiply_bits(k.to_bytes())

    def sum(self):                                          # Sum (src/lib.rs:183-193)
        return Points(self.engine, self.engine.point_sum(self.data))

    def __eq__(self, o):
        return bool((self.data == o.data).all())


class FixedBase:
    """`AffineNielsPoint * Fr` / multiply_bits for one base (src/lib.rs:272-310) with a device-resident window table."""

    def __init__(self, engine, base64, window_bits=0):
        self.engine = engine
        self.table = engine.fixedbase_table(base64, window_bits)

    def multiply_bits(self, by):
        return Points(self.engine, self.engine.fixedbase_mul(self.table, _as_rows(by, 32)))

    def __mul__(self, k):
        return self.multiply_bits(k.to_bytes())


def batch_normalize(engine, ext160):                       # src/lib.rs:1084-1107
    return Points(engine, engine.batch_normalize(_as_rows(ext160, 160)))


def msm(points: Points, scalars: Fr):                       # sum of p * k  (src/lib.rs:183-193 + 873-879)
    if len(points) != len(scalars):
        raise ValueError("length mismatch")
    return Points(points.engine, points.engine.msm(scalars.to_bytes(), points.data))
