"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Pure-Python big-integer restatement of the zkcrypto/jubjub reference algorithms
(crate jubjub 0.10.0 at /root/reference).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this; the product path (jubjub_amd/) never does.

Every function cites the reference file:line it restates.  Field elements are plain
Python ints in canonical (non-Montgomery) form in [0, p); helpers convert to/from the
reference's in-memory Montgomery limbs (4 x u64, R = 2^256) so that golden-limb vectors
from the reference tests can be checked directly.

Points use the *exact* reference formulas (extended coordinates with T split in T1,T2),
so even projective coordinates match what the Rust code computes.

Parity pinning: this oracle is checked (tests/test_oracle_golden.py) against every
known-answer vector in the reference's own tests (src/lib.rs:1456-1935, src/fr.rs:787-1244)
transcribed into tests/golden/reference_vectors.json by tests/golden/make_golden.py.

Third-party arithmetic not on disk: Fq = bls12_381::Scalar 0.8.0 (Cargo.lock:50-53) and
ff 0.13.1 helpers (Cargo.lock:191-194).  Every Fq op except sqrt returns the unique
canonical residue, fixed by the modulus q (README.md:28, doc/evidence/p).  Fq.sqrt is
restated from the published ff::helpers::sqrt_tonelli_shanks algorithm with
ROOT_OF_UNITY = 7^t; its root-sign choice is "parity unpinned" by the reference tests
(decompression removes the ambiguity with the sign bit, lib.rs:518-520).
"""

# --------------------------------------------------------------------------------------
# Constants
# --------------------------------------------------------------------------------------

# Base field modulus q (README.md:28; doc/evidence/p)
Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
# Scalar field modulus r (src/fr.rs:76-82; doc/evidence/l)
R_MOD = 0x0E7DB4EA6533AFA906673B0101343B00A6682093CCC81082D0970E5ED6F72CB7

MONT_R = 1 << 256  # Montgomery radix used by both fields (src/fr.rs:19-21)

# d = -(10240/10241) mod q  (src/lib.rs:398-404)
EDWARDS_D = (-10240 * pow(10241, -1, Q)) % Q
EDWARDS_D2 = (2 * EDWARDS_D) % Q

# FR_MODULUS_BYTES (src/lib.rs:73-76) = r little-endian
FR_MODULUS_BYTES = R_MOD.to_bytes(32, "little")

# Fq two-adicity (bls12_381::Scalar): q - 1 = 2^32 * t, multiplicative generator 7
FQ_S = 32
FQ_T = (Q - 1) >> FQ_S
FQ_ROOT_OF_UNITY = pow(7, FQ_T, Q)
FQ_T_MINUS1_OVER2 = (FQ_T - 1) // 2

# Full-group generator (src/lib.rs:1380-1396): v = 11
GEN_U = 0x62EDCBB8BF3787C88B0F03DDD60A8187CAF55D1B29BF81AFE4B3D35DF1A7ADFE
GEN_V = 11


def limbs_to_int(limbs):
    """[u64;4] little-endian limbs -> int."""
    return sum(int(l) << (64 * i) for i, l in enumerate(limbs))


def int_to_limbs(x, n=4):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


# --------------------------------------------------------------------------------------
# Prime fields (src/fr.rs:246-665 template; identical structure for Fq in bls12_381)
# --------------------------------------------------------------------------------------


class Field:
    """A prime field with the reference's byte/Montgomery conventions."""

    def __init__(self, p, name):
        self.p = p
        self.name = name
        self.R = MONT_R % p  # src/fr.rs:216-222
        self.R2 = (MONT_R * MONT_R) % p  # src/fr.rs:224-230
        self.R3 = (MONT_R * MONT_R * MONT_R) % p  # src/fr.rs:232-238
        self.Rinv = pow(MONT_R, -1, p)
        # INV = -(p^{-1} mod 2^64) mod 2^64  (src/fr.rs:213-214)
        self.INV = (-pow(p, -1, 1 << 64)) % (1 << 64)

    # -- Montgomery <-> canonical (what the reference keeps in memory) --
    def to_mont_limbs(self, a):
        return int_to_limbs((a * MONT_R) % self.p)

    def from_mont_limbs(self, limbs):
        return (limbs_to_int(limbs) * self.Rinv) % self.p

    # -- arithmetic: src/fr.rs:592-665 (results are the unique canonical residue) --
    def add(self, a, b):
        return (a + b) % self.p

    def sub(self, a, b):
        return (a - b) % self.p

    def neg(self, a):
        return (-a) % self.p

    def double(self, a):
        return (2 * a) % self.p

    def mul(self, a, b):
        return (a * b) % self.p

    def square(self, a):
        return (a * a) % self.p

    def pow(self, a, e):
        return pow(a, e, self.p)

    def invert(self, a):
        """src/fr.rs:438-540: a^(p-2); CtOption is None iff a == 0. Returns (value, ok)."""
        if a % self.p == 0:
            return 0, 0
        return pow(a, self.p - 2, self.p), 1

    # -- encodings --
    def from_bytes(self, b):
        """src/fr.rs:268-292: 32 LE bytes, reject if >= p. Returns (value, ok)."""
        assert len(b) == 32
        x = int.from_bytes(bytes(b), "little")
        if x >= self.p:
            # the reference still computes tmp*R2 on the unreduced limbs; the CtOption is None
            return x % self.p, 0
        return x, 1

    def to_bytes(self, a):
        """src/fr.rs:296-308: canonical little-endian."""
        return (a % self.p).to_bytes(32, "little")

    def from_bytes_wide(self, b):
        """src/fr.rs:312-343: 64 LE bytes reduced mod p (d0*R2 + d1*R3 in Montgomery form)."""
        assert len(b) == 64
        return int.from_bytes(bytes(b), "little") % self.p

    def from_raw(self, limbs):
        """src/fr.rs:347-349: integer limbs -> element (reduces mod p)."""
        return limbs_to_int(limbs) % self.p


FQ = Field(Q, "Fq")
FR = Field(R_MOD, "Fr")


def fr_sqrt(a):
    """src/fr.rs:384-399: r = 3 mod 4 so sqrt = a^((r+1)/4); Some iff it squares back."""
    s = pow(a, (R_MOD + 1) // 4, R_MOD)
    return s, int((s * s) % R_MOD == a % R_MOD)


def fq_sqrt(a):
    """Fq::sqrt = ff::helpers::sqrt_tonelli_shanks(self, (t-1)/2) (bls12_381 0.8.0, ff 0.13.1;
    call sites src/lib.rs:515,610,1253).  Constant-time Tonelli-Shanks with S = 32,
    ROOT_OF_UNITY = 7^t.  Returns (x, ok) with ok = (x^2 == a).  Root sign: parity unpinned."""
    p = Q
    a %= p
    w = pow(a, FQ_T_MINUS1_OVER2, p)
    v = FQ_S
    x = (a * w) % p
    b = (x * w) % p
    z = FQ_ROOT_OF_UNITY
    for max_v in range(FQ_S, 0, -1):
        k = 1
        tmp = (b * b) % p
        j_less_than_v = 1
        for j in range(2, max_v):
            tmp_is_one = int(tmp == 1)
            squared = (z if tmp_is_one else tmp)
            squared = (squared * squared) % p
            tmp = tmp if tmp_is_one else squared  # conditional_select(&squared, &tmp, tmp_is_one)
            new_z = squared if tmp_is_one else z  # conditional_select(&z, &squared, tmp_is_one)
            j_less_than_v &= int(j != v)
            k = k if tmp_is_one else j  # conditional_select(&j, &k, tmp_is_one)
            z = new_z if j_less_than_v else z  # conditional_select(&z, &new_z, j_less_than_v)
        result = (x * z) % p
        x = x if b == 1 else result  # conditional_select(&result, &x, b == 1)
        z = (z * z) % p
        b = (b * z) % p
        v = k
    return x, int((x * x) % p == a)


# --------------------------------------------------------------------------------------
# Points.  AffinePoint = (u, v); ExtendedPoint = (U, V, Z, T1, T2);
# AffineNiels = (v_plus_u, v_minus_u, t2d); ExtendedNiels = (v_plus_u, v_minus_u, z, t2d)
# --------------------------------------------------------------------------------------

AFFINE_IDENTITY = (0, 1)  # src/lib.rs:416-421
EXT_IDENTITY = (0, 1, 1, 0, 0)  # src/lib.rs:680-688
AFFINE_NIELS_IDENTITY = (1, 1, 0)  # src/lib.rs:263-269
EXT_NIELS_IDENTITY = (1, 1, 1, 0)  # src/lib.rs:347-354
GENERATOR = (GEN_U, GEN_V)


def affine_to_extended(p):
    """src/lib.rs:213-225, 640-648."""
    u, v = p
    return (u, v, 1, u, v)


def affine_neg(p):
    """src/lib.rs:92-104."""
    return ((-p[0]) % Q, p[1])


def ext_neg(p):
    """src/lib.rs:195-211."""
    u, v, z, t1, t2 = p
    return ((-u) % Q, v, z, (-t1) % Q, t2)


def affine_to_niels(p):
    """src/lib.rs:652-658."""
    u, v = p
    return ((v + u) % Q, (v - u) % Q, (u * v * EDWARDS_D2) % Q)


def ext_to_niels(p):
    """src/lib.rs:728-735."""
    u, v, z, t1, t2 = p
    return ((v + u) % Q, (v - u) % Q, z, (t1 * t2 * EDWARDS_D2) % Q)


def _completed_into_extended(cu, cv, cz, ct):
    """src/lib.rs:1052-1060."""
    return ((cu * ct) % Q, (cv * cz) % Q, (cz * ct) % Q, cu % Q, cv % Q)


def ext_double(p):
    """src/lib.rs:739-828."""
    u, v, z, _, _ = p
    uu = (u * u) % Q
    vv = (v * v) % Q
    zz2 = (2 * z * z) % Q
    uv2 = ((u + v) * (u + v)) % Q
    vv_plus_uu = (vv + uu) % Q
    vv_minus_uu = (vv - uu) % Q
    return _completed_into_extended(
        (uv2 - vv_plus_uu) % Q, vv_plus_uu, vv_minus_uu, (zz2 - vv_minus_uu) % Q
    )


def ext_add_ext_niels(p, n):
    """src/lib.rs:883-920."""
    u, v, z, t1, t2 = p
    vpu, vmu, nz, t2d = n
    a = ((v - u) * vmu) % Q
    b = ((v + u) * vpu) % Q
    c = (t1 * t2 * t2d) % Q
    d = (2 * z * nz) % Q
    return _completed_into_extended((b - a) % Q, (b + a) % Q, (d + c) % Q, (d - c) % Q)


def ext_sub_ext_niels(p, n):
    """src/lib.rs:922-940."""
    u, v, z, t1, t2 = p
    vpu, vmu, nz, t2d = n
    a = ((v - u) * vpu) % Q
    b = ((v + u) * vmu) % Q
    c = (t1 * t2 * t2d) % Q
    d = (2 * z * nz) % Q
    return _completed_into_extended((b - a) % Q, (b + a) % Q, (d - c) % Q, (d + c) % Q)


def ext_add_affine_niels(p, n):
    """src/lib.rs:944-968."""
    u, v, z, t1, t2 = p
    vpu, vmu, t2d = n
    a = ((v - u) * vmu) % Q
    b = ((v + u) * vpu) % Q
    c = (t1 * t2 * t2d) % Q
    d = (2 * z) % Q
    return _completed_into_extended((b - a) % Q, (b + a) % Q, (d + c) % Q, (d - c) % Q)


def ext_sub_affine_niels(p, n):
    """src/lib.rs:970-988."""
    u, v, z, t1, t2 = p
    vpu, vmu, t2d = n
    a = ((v - u) * vpu) % Q
    b = ((v + u) * vmu) % Q
    c = (t1 * t2 * t2d) % Q
    d = (2 * z) % Q
    return _completed_into_extended((b - a) % Q, (b + a) % Q, (d - c) % Q, (d + c) % Q)


def ext_add(p, q):
    """src/lib.rs:992-999: self + other.to_niels()."""
    return ext_add_ext_niels(p, ext_to_niels(q))


def ext_sub(p, q):
    """src/lib.rs:1001-1008."""
    return ext_sub_ext_niels(p, ext_to_niels(q))


def ext_add_affine(p, a):
    """src/lib.rs:1012-1019."""
    return ext_add_affine_niels(p, affine_to_niels(a))


def ext_sub_affine(p, a):
    """src/lib.rs:1021-1028."""
    return ext_sub_affine_niels(p, affine_to_niels(a))


def ext_to_affine(p):
    """src/lib.rs:227-243."""
    u, v, z, _, _ = p
    zinv = pow(z, Q - 2, Q)
    return ((u * zinv) % Q, (v * zinv) % Q)


def ext_eq(p, q):
    """src/lib.rs:153-163."""
    return (p[0] * q[2] - q[0] * p[2]) % Q == 0 and (p[1] * q[2] - q[1] * p[2]) % Q == 0


def ext_is_identity(p):
    """src/lib.rs:691-696."""
    return int(p[0] % Q == 0 and (p[1] - p[2]) % Q == 0)


def ext_is_small_order(p):
    """src/lib.rs:699-705."""
    return int(ext_double(ext_double(p))[0] == 0)


def ext_mul_by_cofactor(p):
    """src/lib.rs:722-724."""
    return ext_double(ext_double(ext_double(p)))


def _ladder_bits(by):
    """Bit order of the ladders: bytes reversed, each byte MSB->LSB, skip the first 4
    (src/lib.rs:283-288, 368-372) -> bits 251..0 of the little-endian integer."""
    assert len(by) == 32
    k = int.from_bytes(bytes(by), "little")
    return [(k >> i) & 1 for i in range(251, -1, -1)]


def ext_niels_multiply(n, by):
    """src/lib.rs:357-379."""
    acc = EXT_IDENTITY
    for bit in _ladder_bits(by):
        acc = ext_double(acc)
        acc = ext_add_ext_niels(acc, n if bit else EXT_NIELS_IDENTITY)
    return acc


def affine_niels_multiply(n, by):
    """src/lib.rs:272-295."""
    acc = EXT_IDENTITY
    for bit in _ladder_bits(by):
        acc = ext_double(acc)
        acc = ext_add_affine_niels(acc, n if bit else AFFINE_NIELS_IDENTITY)
    return acc


def ext_multiply(p, by):
    """src/lib.rs:831-833."""
    return ext_niels_multiply(ext_to_niels(p), by)


def ext_mul_scalar(p, k):
    """Mul<&Fr> for &ExtendedPoint (src/lib.rs:873-879): k is a canonical Fr int."""
    return ext_multiply(p, FR.to_bytes(k))


def affine_mul_scalar(p, k):
    """Mul<&Fr> for &AffinePoint (src/lib.rs:1109-1115)."""
    return affine_niels_multiply(affine_to_niels(p), FR.to_bytes(k))


def ext_is_torsion_free(p):
    """src/lib.rs:709-711."""
    return ext_is_identity(ext_multiply(p, FR_MODULUS_BYTES))


def ext_is_prime_order(p):
    """src/lib.rs:717-719."""
    return int(ext_is_torsion_free(p) and not ext_is_identity(p))


def affine_is_on_curve(p):
    """src/lib.rs:670-675 (test-only in the reference)."""
    u, v = p
    u2, v2 = (u * u) % Q, (v * v) % Q
    return (v2 - u2) % Q == (1 + EDWARDS_D * u2 * v2) % Q


def ext_is_on_curve(p):
    """src/lib.rs:864-870."""
    u, v, z, t1, t2 = p
    if z % Q == 0:
        return False
    a = ext_to_affine(p)
    return affine_is_on_curve(a) and (a[0] * a[1] * z - t1 * t2) % Q == 0


def affine_to_bytes(p):
    """src/lib.rs:455-464."""
    u, v = p
    tmp = bytearray(FQ.to_bytes(v))
    tmp[31] |= (FQ.to_bytes(u)[0] << 7) & 0xFF
    return bytes(tmp)


def affine_from_bytes(b, zip216=True):
    """src/lib.rs:469-534 (from_bytes / from_bytes_pre_zip216_compatibility).
    Returns ((u, v), ok); on failure the point is (0, 0) like the C ABI's zeroed output."""
    b = bytearray(b)
    assert len(b) == 32
    sign = b[31] >> 7
    b[31] &= 0x7F
    v, ok = FQ.from_bytes(b)
    if not ok:
        return (0, 0), 0
    v2 = (v * v) % Q
    den, _ = FQ.invert((1 + EDWARDS_D * v2) % Q)
    u, ok = fq_sqrt(((v2 - 1) * den) % Q)
    if not ok:
        return (0, 0), 0
    flip_sign = (FQ.to_bytes(u)[0] ^ sign) & 1
    final_u = (-u) % Q if flip_sign else u
    if zip216 and u == 0 and flip_sign:
        return (0, 0), 0
    return (final_u, v), 1


def batch_from_bytes(items):
    """src/lib.rs:541-627: same results as from_bytes with ZIP-216 always enabled."""
    return [affine_from_bytes(b, True) for b in items]


def batch_normalize(points):
    """src/lib.rs:1084-1107: returns (normalized extended points, affine points)."""
    out_ext, out_aff = [], []
    for p in points:
        u, v = ext_to_affine(p)
        out_ext.append((u, v, 1, u, v))
        out_aff.append((u, v))
    return out_ext, out_aff


def ext_sum(points):
    """Sum for ExtendedPoint (src/lib.rs:183-193): left fold of + from the identity."""
    acc = EXT_IDENTITY
    for p in points:
        acc = ext_add(acc, p)
    return acc


def msm(scalars, points):
    """MSM oracle semantics: sum_i (P_i * k_i), P_i affine, k_i 32-byte LE
    bit patterns (low 252 bits used)."""
    acc = EXT_IDENTITY
    for k, p in zip(scalars, points):
        acc = ext_add(acc, ext_multiply(affine_to_extended(p), k))
    return acc


def recommended_wnaf_for_num_scalars(n):
    """src/lib.rs:1320-1335."""
    rec = [1, 3, 7, 20, 43, 120, 273, 563, 1630, 3128, 7933, 62569]
    ret = 4
    for r in rec:
        if n > r:
            ret += 1
        else:
            break
    return ret


# --------------------------------------------------------------------------------------
# Fast (non-reference-structured) helpers used only to build large expected values quickly.
# They compute the same group element; tests compare canonical affine output.
# --------------------------------------------------------------------------------------


def affine_add_fast(p, q):
    """Unified affine twisted-Edwards addition (a = -1); same group law as lib.rs:883-920."""
    u1, v1 = p
    u2, v2 = q
    t = (EDWARDS_D * u1 * u2 * v1 * v2) % Q
    u3 = ((u1 * v2 + v1 * u2) * pow(1 + t, -1, Q)) % Q
    v3 = ((v1 * v2 + u1 * u2) * pow(1 - t, -1, Q)) % Q
    return (u3, v3)


def scalar_mul_fast(p, k):
    """k * p for an int k, via extended coordinates double-and-add (no fixed length)."""
    acc = EXT_IDENTITY
    n = ext_to_niels(affine_to_extended(p))
    for i in range(k.bit_length() - 1, -1, -1):
        acc = ext_double(acc)
        if (k >> i) & 1:
            acc = ext_add_ext_niels(acc, n)
    return ext_to_affine(acc)


# --------------------------------------------------------------------------------------
# Synthetic input generator shared with bench/tests: counter-based splitmix64
# --------------------------------------------------------------------------------------

SEED = 0x4A55424A5542
_M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def synth_scalar(i, seed=SEED):
    """scalar_i = 4 PRNG u64, top 4 bits cleared, minus r if >= r."""
    k = 0
    for j in range(4):
        k |= splitmix64((seed + i * 4 + j) & _M64) << (64 * j)
    k &= (1 << 252) - 1
    if k >= R_MOD:
        k -= R_MOD
    return k


def synth_bytes32(i, seed=SEED):
    """the four PRNG words of unit i as 32 raw little-endian bytes (jj_synth_bytes32)"""
    k = 0
    for j in range(4):
        k |= splitmix64((seed + i * 4 + j) & _M64) << (64 * j)
    return k.to_bytes(32, "little")


def synth_point(i, seed=SEED, subgroup=False):
    """ExtendedPoint::random / SubgroupPoint::random (src/lib.rs:1244-1267, 1290-1298) over the counter-based stream of
    unit i: attempt t reads splitmix64(seed + (i << 16) + 16 t + j), j = 0..7 for the 64 bytes of Fq::random
    (from_bytes_wide, src/fr.rs:684-688 pattern), j = 8 for `next_u32() % 2`.  Returns (affine point, attempts)."""
    base = (seed + (i << 16)) & _M64
    t = 0
    while True:
        wide = 0
        for j in range(8):
            wide |= splitmix64((base + 16 * t + j) & _M64) << (64 * j)
        flip = splitmix64((base + 16 * t + 8) & _M64) & 1
        t += 1
        v = wide % Q
        v2 = v * v % Q
        den = (1 + EDWARDS_D * v2) % Q
        u2 = (v2 - 1) * (pow(den, -1, Q) if den else 0) % Q
        u, ok = fq_sqrt(u2)
        if not ok:
            continue
        if flip:
            u = -u % Q
        if (u, v) == AFFINE_IDENTITY:
            continue
        if subgroup:
            e = ext_mul_by_cofactor(affine_to_extended((u, v)))
            if ext_is_identity(e):
                continue
            return ext_to_affine(e), t
        return (u, v), t
