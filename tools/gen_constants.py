#!/usr/bin/env python3
"""
Generates jubjub_amd/csrc/jj_constants.h: every field/curve constant the HIP kernels use, in the
device representation (9 limbs x 29 bits, Montgomery radix R = 2^261).

All values are DERIVED here from the two moduli and d = -10240/10241 (reference: README.md:28,
src/fr.rs:76-82, src/lib.rs:398-412, 1380-1396); nothing is copied from the oracle.  The generated header
is committed so the build needs no Python.
"""
import os

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = 0x0E7DB4EA6533AFA906673B0101343B00A6682093CCC81082D0970E5ED6F72CB7
LB, NL = 29, 9
MASK = (1 << LB) - 1
MONT = 1 << (LB * NL)  # 2^261


def limbs(x, n=NL):
    assert 0 <= x < (1 << (LB * n))
    return [(x >> (LB * i)) & MASK for i in range(n)]


def arr(name, vals, ty="u32"):
    body = ", ".join("0x%08xu" % v for v in vals)
    return "  static constexpr %s %s[%d] = {%s};\n" % (ty, name, len(vals), body)


def signed_digits(x):
    """x (any sign) as 9 digits: limbs 0..7 in [0, 2^29), top limb signed (two's complement u32)."""
    out = []
    for _ in range(NL - 1):
        out.append(x & MASK)
        x >>= LB          # floor (arithmetic) shift
    assert -(1 << 31) <= x < (1 << 31)
    out.append(x & 0xFFFFFFFF)
    return out


def bias(p, K, s):
    """K*p written with limbs 0..7 in [2^s, 2^s + 2^29) (borrowing 2^(s-29) from the next limb):
    same VALUE K*p, but every low limb is >= 2^s so that a + bias - b never underflows limb-wise
    for b with limbs < 2^s."""
    l = limbs(K * p)
    lend = 1 << (s - LB)
    for i in range(NL - 1):
        l[i] += 1 << s
        l[i + 1] -= lend
    assert all(v >= 0 for v in l), (K, s, l)
    assert sum(v << (LB * i) for i, v in enumerate(l)) == K * p
    assert all((1 << s) <= v < (1 << s) + (1 << LB) for v in l[:-1])
    return l


def bits_msb_first(e):
    return [int(c) for c in bin(e)[2:]]


def words32(e, n):
    return [(e >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def field_block(name, p, extra=""):
    ninv = (-pow(p, -1, 1 << LB)) % (1 << LB)
    s = "struct %s {\n" % name
    s += "  // modulus p, 9 x 29-bit limbs, little-endian\n"
    s += arr("P", limbs(p))
    s += "  static constexpr u32 NINV = 0x%08xu;  // -p^-1 mod 2^29\n" % ninv
    s += arr("ONE", limbs(MONT % p))  # R mod p  (Montgomery 1)
    s += arr("R2", limbs((MONT * MONT) % p))  # to-Montgomery multiplier
    s += arr("R2_256", limbs(((1 << 256) * MONT * MONT) % p))  # converts hi half of a 512-bit value: x*2^256 -> Montgomery
    s += arr("HOST_R", limbs((1 << 256) % p))   # 2^256 mod p as a PLAIN integer: mul(a, HOST_R) = value(a) * 2^256, the 4 x 64-bit Montgomery form of the host tail
    s += arr("FROM_HOST", limbs(pow(2, 2 * LB * NL - 256, p)))   # 2^266 mod p as a PLAIN integer: mul(x, FROM_HOST) with x = value * 2^256 (the host tail's form) = value * 2^261 (Montgomery)
    s += arr("NEGP_DIGITS", signed_digits(-p))   # -p as the digits a Montgomery product emits (limbs 0..7 in [0, 2^29), signed top limb)
    s += arr("PM2", words32(p - 2, 8))  # exponent for inversion
    s += "  static constexpr int PBITS = %d;\n" % p.bit_length()
    s += extra
    s += "};\n\n"
    return s


def sqrt_mod(a, p):
    """Tonelli-Shanks; returns the smaller root or None."""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    s, t = 0, p - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    z = 2
    while pow(z, (p - 1) // 2, p) == 1:
        z += 1
    c, x, b, m = pow(z, t, p), pow(a, (t + 1) // 2, p), pow(a, t, p), s
    while b != 1:
        i, b2 = 0, b
        while b2 != 1:
            b2, i = b2 * b2 % p, i + 1
        g = pow(c, 1 << (m - i - 1), p)
        x, c = x * g % p, g * g % p
        b, m = b * c % p, i
    return min(x, p - x)


def torsion_pairing_constants(d):
    """Constants of the Tate-pairing subgroup test (jj_curve.h Curve::is_torsion_free).

    E(Fq) is cyclic of order 8r, so P is in the prime-order subgroup iff the order-8 Tate pairing
    t_8(T, P) = f_{8,T}(P)^((q-1)/8) is 1 for a generator T of the 8-torsion.  On the Montgomery model
    B y^2 = x^3 + A x^2 + x (x = (1+v)/(1-v), y = x/u) three Miller doubling steps give
        f_{8,T} = L1^4 L2^2 / (V1^4 X Z) / B,   L_i = Y - l_i X - n_i Z  (tangents at T and 2T),  V1 = X - Z,
    with (X:Y:Z) = ((1+v)u : 1+v : (1-v)u).  Modulo 8th powers that is C g^4 L2^2 k^7 with g = L1 u v,
    k = u^2 (1-v^2), C = 16 B^7, and L_i = (1+v) - u (a_i v + b_i), a_i = l_i - n_i, b_i = l_i + n_i."""
    q = Q
    inv = lambda x: pow(x, -1, q)

    def add(P, S):
        (u1, v1), (u2, v2) = P, S
        k = d * u1 * u2 * v1 * v2 % q
        return ((u1 * v2 + v1 * u2) * inv(1 + k) % q, (v1 * v2 + u1 * u2) * inv(1 - k) % q)

    def mul(P, k):
        acc = (0, 1)
        while k:
            if k & 1:
                acc = add(acc, P)
            P, k = add(P, P), k >> 1
        return acc

    v = 2
    while True:  # smallest v whose point has an order-8 component
        u = sqrt_mod((v * v - 1) * inv(1 + d * v * v), q)
        if u:
            T = mul((u, v), R)
            if mul(T, 4) != (0, 1):
                break
        v += 1
    assert mul(T, 8) == (0, 1)
    A = 2 * (d - 1) * inv(-1 - d) % q
    B = 4 * inv(-1 - d) % q

    def to_mont(P):
        x = (1 + P[1]) * inv(1 - P[1]) % q
        return x, x * inv(P[0]) % q

    def tangent(M):
        lam = (3 * M[0] * M[0] + 2 * A * M[0] + 1) * inv(2 * B * M[1]) % q
        return lam, (M[1] - lam * M[0]) % q

    T2 = add(T, T)
    assert to_mont(T2)[0] == 1 and add(T2, T2) == (0, q - 1)
    (l1, n1), (l2, n2) = tangent(to_mont(T)), tangent(to_mont(T2))
    return {"TP_A1": (l1 - n1) % q, "TP_B1": (l1 + n1) % q, "TP_A2": (l2 - n2) % q, "TP_B2": (l2 + n2) % q,
            "TP_C": 16 * pow(B, 7, q) % q}


def main():
    d = (-10240 * pow(10241, -1, Q)) % Q
    d2 = (2 * d) % Q
    t = (Q - 1) >> 32
    root = pow(7, t, Q)
    gen_u = 0x62EDCBB8BF3787C88B0F03DDD60A8187CAF55D1B29BF81AFE4B3D35DF1A7ADFE
    assert (11 * 11 - gen_u * gen_u) % Q == (1 + d * gen_u * gen_u * 121) % Q
    mont = lambda x, p: (x * MONT) % p

    fq_extra = ""
    fq_extra += arr("D", limbs(mont(d, Q)))
    fq_extra += arr("D2", limbs(mont(d2, Q)))
    fq_extra += arr("ROOT_OF_UNITY", limbs(mont(root, Q)))  # 7^t, 2^32-th root of unity
    fq_extra += arr("ROOT_OF_UNITY_INV", limbs(mont(pow(root, -1, Q), Q)))
    fq_extra += arr("TM1D2", words32((t - 1) // 2, 8))  # (t-1)/2, Tonelli-Shanks exponent
    fq_extra += "  static constexpr int TWO_ADICITY = 32;\n"
    # sqrt via 2-adic discrete log needs powers of the root: ROOT^(2^i) -- generated as table too
    pw = [mont(pow(root, 1 << i, Q), Q) for i in range(32)]
    fq_extra += "  static constexpr u32 ROOT_POW2[32][9] = {\n" + ",\n".join(
        "    {" + ", ".join("0x%08xu" % v for v in limbs(x)) + "}" for x in pw) + "};\n"

    fq_extra += "  // Tate-pairing subgroup test (see torsion_pairing_constants in tools/gen_constants.py)\n"
    for nm, val in torsion_pairing_constants(d).items():
        fq_extra += arr(nm, limbs(mont(val, Q)))

    fr_extra = ""
    fr_extra += arr("SQRT_EXP", words32((R + 1) // 4, 8))  # src/fr.rs:388-393

    out = "// GENERATED by tools/gen_constants.py -- do not edit.\n"
    out += "// Device representation: 9 limbs x 29 bits, Montgomery radix 2^261.\n#pragma once\n#include <stdint.h>\n"
    out += "namespace jj {\ntypedef uint32_t u32;\ntypedef uint64_t u64;\n"
    out += "constexpr int NL = 9;\nconstexpr int LB = 29;\nconstexpr u32 LMASK = 0x1fffffffu;\n\n"
    out += field_block("FqP", Q, fq_extra)
    out += field_block("FrP", R, fr_extra)
    out += "// r (scalar field modulus) as 32 little-endian bytes = FR_MODULUS_BYTES (reference src/lib.rs:73-76)\n"
    out += "constexpr uint8_t FR_MODULUS_BYTES[32] = {" + ", ".join(str(b) for b in R.to_bytes(32, "little")) + "};\n"
    out += "constexpr u32 FR_MODULUS_W[8] = {" + ", ".join("0x%08xu" % v for v in words32(R, 8)) + "};\n"
    out += "// generator (reference src/lib.rs:1380-1396), canonical 32-bit words\n"
    out += "constexpr u32 GEN_U_W[8] = {" + ", ".join("0x%08xu" % v for v in words32(gen_u, 8)) + "};\n"
    out += "constexpr u32 GEN_V_W[8] = {11u, 0, 0, 0, 0, 0, 0, 0};\n"
    # signed-window recoding offsets: sum_i 2^(w-1) * 2^(w*i) over the windows below bit 252
    rec4 = sum(8 << (4 * i) for i in range(63))
    rec6 = sum(32 << (6 * i) for i in range(42))
    out += "constexpr u32 RECODE4[8] = {" + ", ".join("0x%08xu" % v for v in words32(rec4, 8)) + "};\n"
    out += "constexpr u32 RECODE6[8] = {" + ", ".join("0x%08xu" % v for v in words32(rec6, 8)) + "};\n"
    out += "}  // namespace jj\n"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "jubjub_amd", "csrc", "jj_constants.h")
    with open(path, "w") as f:
        f.write(out)
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
