// Jubjub prime-field arithmetic for CDNA4 (gfx950) — device code.
//
// Representation (MI355X-first, NOT the reference's 4x64 layout): one field element per lane, held in
// registers as 9 limbs x 29 bits ("reduced radix"), Montgomery form with R = 2^261.
//
// Why: measured on MI355X (experiments/ubench, profiles/ubench_r1.txt) v_mad_u64_u32 issues at the same
// 4-cycle/wave64 rate as every other VOP3 instruction, but any multi-word carry chain goes through an SGPR
// carry (v_add_co/v_addc_co), which on gfx940/950 costs an extra 2 wait states per link ("VALU writes SGPR ->
// VALU reads it").  With 29-bit limbs a 9x9 schoolbook column sum (<= 18 products of < 2^61) never overflows a
// 64-bit VGPR pair, so a Montgomery product is 153 back-to-back v_mad_u64_u32 (81 a*b + 72 m*p) plus ~75 plain
// shift/mask/add instructions and NO carry flags at all.  Additions are 9 independent v_add_u32 (lazy, no
// carry); subtraction adds a limb-lifted multiple of p (value K*p) so no limb underflows.
//
// What it restates: results are the same canonical residues as the reference's Fq/Fr (reference
// src/fr.rs:246-665 is the template for both fields; Fq = bls12_381::Scalar).  Only canonical little-endian
// 32-byte encodings cross the kernel boundary (reference Fr::to_bytes src/fr.rs:296-308, from_bytes 268-292).
//
// Bounds contract (checked by tools/bounds_check.py and by tests at extreme values):
//   "N"   : limbs < 2^29 (top limb small), value < 2p            -- output of mul/sqr/sub/norm
//   "L"   : limbs < 2^30 + 2^8,            value < 4p            -- output of add(N, N)
//   mul/sqr inputs: per-limb bound product A*B < 2^60.6, value product alpha*beta <= 64 (in units of p)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "jj_constants.h"

namespace jj {

#define JJ_DEV __device__ __forceinline__

struct Fe {
  u32 l[NL];
};

// d = a*b + c as ONE v_mad_u64_u32 with a fixed association: c is the accumulator coming in, so a carry from the
// previous column rides in as the addend of the next column's first multiply-add (hipcc would otherwise
// re-associate every column into an independent chain and spend an extra v_lshl_add_u64 per column to merge).
// Not volatile: the scheduler may still interleave independent field operations.
static JJ_DEV u64 mad_vv(u32 a, u32 b, u64 c) {
  u64 d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
  return d;
}
// same with a wave-uniform (compile-time constant) multiplier held in an SGPR
static JJ_DEV u64 mad_vs(u32 a, u32 k, u64 c) {
  u64 d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c) : "vcc");
  return d;
}

// JJ_MUL_PIN (default of Field's PIN parameter): how mul/sqr keep LLVM's reassociation from moving the column carry to
// the END of each column's sum, which costs one 64-bit add per column (17 per product):
//   0 = let it;  3 = give every partial sum a second use in an empty, non-volatile asm statement chained through a
//   dummy SGPR token (no code, no hazard padding, ordered only inside one product so that independent products still
//   interleave): 205 instead of 223 instructions per product.  Measured +5..7 % on the ladders and the decoder.  The
//   pins lengthen live ranges, so kernels that are already register-bound (k_msm_accumulate) instantiate PIN = 0.
//   4 = as 3, plus the column shift amount routed through the token chain, which retires a column's pins before the
//   next column starts: fits k_msm_accumulate in 97 VGPRs, but measured 1-3 % slower than 3 everywhere (kept for reference).
#ifndef JJ_MUL_PIN
#define JJ_MUL_PIN 3
#endif
template <class P, int PIN = JJ_MUL_PIN>
struct Field {
  // ---------------------------------------------------------------- constants
  static JJ_DEV Fe one() { Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = P::ONE[i]; return r; }
  static JJ_DEV Fe zero() { Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = 0; return r; }
  template <int N>
  static JJ_DEV Fe konst(const u32 (&c)[N]) { Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = c[i]; return r; }

  // ---------------------------------------------------------------- Montgomery reduction of 17 columns
  // c[0..16] hold the column sums of a 9x9 limb product (each < ~2^63); on return r = (sum c[k] 2^(29k)) / 2^261 mod p
  // with limbs < 2^29.  Mirrors the role of reference montgomery_reduce (src/fr.rs:544-588) with 29-bit digits.
  static JJ_DEV Fe reduce(u64 (&c)[2 * NL]) {
    _Pragma("unroll") for (int k = 0; k < NL; k++) {
      u32 m;
      if constexpr (P::NINV == LMASK) m = (0u - (u32)c[k]) & LMASK;   // p = 1 mod 2^29  (Fq): m = -c
      else m = ((u32)c[k] * P::NINV) & LMASK;                          // generic (Fr)
      if constexpr (P::P[0] == 1u) c[k] += m;
      else c[k] += (u64)m * P::P[0];
      c[k + 1] += c[k] >> LB;                                           // low 29 bits of c[k] are now zero
      _Pragma("unroll") for (int j = 1; j < NL; j++) c[k + j] += (u64)m * P::P[j];
    }
    Fe r;
    _Pragma("unroll") for (int k = NL; k < 2 * NL - 1; k++) {
      r.l[k - NL] = (u32)c[k] & LMASK;
      c[k + 1] += c[k] >> LB;
    }
    r.l[NL - 1] = (u32)c[2 * NL - 1];
    return r;
  }

#ifndef JJ_MUL_VARIANT
#define JJ_MUL_VARIANT 1
#endif
#if JJ_MUL_VARIANT == 2
  // Variant 2: finely-integrated product scanning with pinned association (mad_vv / mad_vs).
  template <bool SQUARE>
  static JJ_DEV Fe mul_fips(const Fe& a, const Fe& b) {
    u32 m[NL];
    u32 b2[NL];
    if constexpr (SQUARE) { _Pragma("unroll") for (int i = 0; i < NL; i++) b2[i] = a.l[i] << 1; }
    Fe r;
    u64 acc = 0;
    _Pragma("unroll") for (int k = 0; k < 2 * NL - 1; k++) {
      _Pragma("unroll") for (int i = 0; i < NL; i++) {
        const int j = k - i;
        if (j < 0 || j >= NL) continue;
        if constexpr (SQUARE) {
          if (j > i) acc = mad_vv(a.l[i], b2[j], acc);
          else if (j == i) acc = mad_vv(a.l[i], a.l[i], acc);
        } else {
          acc = mad_vv(a.l[i], b.l[j], acc);
        }
      }
      _Pragma("unroll") for (int i = 0; i < NL; i++) {
        const int j = k - i;
        if (i >= k || j < 1 || j >= NL) continue;
        acc = mad_vs(m[i], P::P[j], acc);
      }
      if (k < NL) {
        u32 mk;
        if constexpr (P::NINV == LMASK) mk = (0u - (u32)acc) & LMASK;
        else mk = ((u32)acc * P::NINV) & LMASK;
        m[k] = mk;
        if constexpr (P::P[0] == 1u) acc += mk; else acc = mad_vs(mk, P::P[0], acc);
      } else {
        r.l[k - NL] = (u32)acc & LMASK;
      }
      acc >>= LB;
    }
    r.l[NL - 1] = (u32)acc;
    return r;
  }
  static JJ_DEV Fe mul(const Fe& a, const Fe& b) { return mul_fips<false>(a, b); }
  static JJ_DEV Fe sqr(const Fe& a) { return mul_fips<true>(a, a); }
#elif JJ_MUL_VARIANT == 0
  // Variant 0: all product columns first, then a separate reduction sweep (reduce()).
  static JJ_DEV Fe mul(const Fe& a, const Fe& b) {
    u64 c[2 * NL];
    _Pragma("unroll") for (int k = 0; k < 2 * NL - 1; k++) {
      u64 s = 0;
      _Pragma("unroll") for (int i = 0; i < NL; i++) {
        const int j = k - i;
        if (j >= 0 && j < NL) s += (u64)a.l[i] * b.l[j];
      }
      c[k] = s;
    }
    c[2 * NL - 1] = 0;
    return reduce(c);
  }
  static JJ_DEV Fe sqr(const Fe& a) {
    u32 a2[NL];
    _Pragma("unroll") for (int i = 0; i < NL; i++) a2[i] = a.l[i] << 1;
    u64 c[2 * NL];
    _Pragma("unroll") for (int k = 0; k < 2 * NL - 1; k++) {
      u64 s = 0;
      _Pragma("unroll") for (int i = 0; i < NL; i++) {
        const int j = k - i;
        if (j > i && j < NL) s += (u64)a.l[i] * a2[j];
      }
      if ((k & 1) == 0) s += (u64)a.l[k / 2] * a.l[k / 2];
      c[k] = s;
    }
    c[2 * NL - 1] = 0;
    return reduce(c);
  }
#else
  // Variant 1 (default): finely-integrated product scanning.  One running 64-bit accumulator walks the 18
  // columns; the carry out of column k is simply the addend of the first v_mad_u64_u32 of column k+1, so carry
  // propagation costs no instruction.  Per column: the a*b terms, the m_i*p_j terms, then (k < 9) the Montgomery
  // digit m_k = -acc mod 2^29 and acc = (acc + m_k) >> 29, or (k >= 9) emit a limb and shift.
  // r = a*b/R mod p.  reference Fr::mul src/fr.rs:592-616 + montgomery_reduce 544-588.
  template <bool SQUARE>
  static JJ_DEV Fe mul_fips(const Fe& a, const Fe& b) {
    u32 m[NL];
    u32 b2[NL];
    if constexpr (SQUARE) { _Pragma("unroll") for (int i = 0; i < NL; i++) b2[i] = a.l[i] << 1; }
    Fe r;
    u64 acc = 0;
    u32 p0 = P::P[0];
    asm("" : "+s"(p0));   // opaque to the optimiser
    [[maybe_unused]] u32 pin_tok = 0;
    [[maybe_unused]] u32 pin_sh = LB;     // PIN == 4: the column shift amount, routed through the token chain
#define JJ_PIN(x) do { if constexpr (PIN == 3 || PIN == 4) asm("" : "+s"(pin_tok) : "v"(x)); } while (0)
    _Pragma("unroll") for (int k = 0; k < 2 * NL - 1; k++) {
      _Pragma("unroll") for (int i = 0; i < NL; i++) {
        const int j = k - i;
        if (j < 0 || j >= NL) continue;
        if constexpr (SQUARE) {
          if (j > i) { acc += (u64)a.l[i] * b2[j]; JJ_PIN(acc); }
          else if (j == i) { acc += (u64)a.l[i] * a.l[i]; JJ_PIN(acc); }
        } else {
          acc += (u64)a.l[i] * b.l[j]; JJ_PIN(acc);
        }
      }
      _Pragma("unroll") for (int i = 0; i < NL; i++) {
        const int j = k - i;
        if (i >= k || i >= NL || j < 1 || j >= NL) continue;   // m_i exists for i < min(k, 9); p_0 handled below
        acc += (u64)m[i] * P::P[j]; JJ_PIN(acc);
      }
      if (k < NL) {
        u32 mk;
        if constexpr (P::NINV == LMASK) mk = (0u - (u32)acc) & LMASK;
        else mk = ((u32)acc * P::NINV) & LMASK;
        m[k] = mk;
        // acc += mk * p_0.  For Fq p_0 = 1: multiplying by an opaque 1 keeps this a single v_mad_u64_u32 instead
        // of zero-extending mk into a register pair (v_mov) and a 64-bit add.
        acc += (u64)mk * p0; JJ_PIN(acc);
      } else {
        r.l[k - NL] = (u32)acc & LMASK;
      }
      if constexpr (PIN == 4) { asm("" : "+s"(pin_sh), "+s"(pin_tok)); acc >>= pin_sh; }   // all pins of this column retire before the next one starts
      else acc >>= LB;
    }
    r.l[NL - 1] = (u32)acc;
    if constexpr (PIN == 3 || PIN == 4) asm volatile("" ::"s"(pin_tok));
#undef JJ_PIN
    return r;
  }
  static JJ_DEV Fe mul(const Fe& a, const Fe& b) { return mul_fips<false>(a, b); }
  // r = a*a/R mod p.  reference Fr::square src/fr.rs:353-381 (same cross-term doubling idea).
  static JJ_DEV Fe sqr(const Fe& a) { return mul_fips<true>(a, a); }
#endif

  // ---------------------------------------------------------------- additive ops (lazy, carry-free)
  // r = a + b, no carry.  reference Fr::add src/fr.rs:638-647 (which reduces; we defer).
  static JJ_DEV Fe add(const Fe& a, const Fe& b) {
    Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i]; return r;
  }
  // one parallel carry step: limbs < 2^32 in, limbs <= 2^29 + 7 out, value unchanged.
  static JJ_DEV Fe carry(const Fe& a) {
    Fe r;
    r.l[0] = a.l[0] & LMASK;
    _Pragma("unroll") for (int i = 1; i < NL - 1; i++) r.l[i] = (a.l[i] & LMASK) + (a.l[i - 1] >> LB);
    r.l[NL - 1] = a.l[NL - 1] + (a.l[NL - 2] >> LB);
    return r;
  }
  // r = a - b (+ 3p), b with limbs <= 2^30, value(b) < 3p.  Output carried ("N"-like limbs <= 2^29+7).
  // reference Fr::sub src/fr.rs:620-634.
  static JJ_DEV Fe sub(const Fe& a, const Fe& b) {
    Fe t; _Pragma("unroll") for (int i = 0; i < NL; i++) t.l[i] = a.l[i] + P::BIAS_N[i] - b.l[i];
    return carry(t);
  }
  // same without the carry step: limbs < 2^31 (a N-like), for operands that meet a carried partner in their next multiply
  static JJ_DEV Fe sub_lazy(const Fe& a, const Fe& b) {
    Fe t; _Pragma("unroll") for (int i = 0; i < NL; i++) t.l[i] = a.l[i] + P::BIAS_N[i] - b.l[i];
    return t;
  }
  // r = a - b (+ 5p), b with limbs <= 2^31, value(b) < 5p.
  static JJ_DEV Fe sub_wide(const Fe& a, const Fe& b) {
    Fe t; _Pragma("unroll") for (int i = 0; i < NL; i++) t.l[i] = a.l[i] + P::BIAS_L[i] - b.l[i];
    return carry(t);
  }
  // r = 2a - b (+ 5p) in one pass (the doubling's 2Z^2 - (VV - UU)); same bounds as sub_wide(add(a, a), b)
  static JJ_DEV Fe dbl_sub_wide(const Fe& a, const Fe& b) {
    Fe t; _Pragma("unroll") for (int i = 0; i < NL; i++) t.l[i] = (a.l[i] << 1) + P::BIAS_L[i] - b.l[i];
    return carry(t);
  }
  // r = -a (+3p).  reference Fr::neg src/fr.rs:651-665.
  static JJ_DEV Fe neg(const Fe& a) {
    Fe t; _Pragma("unroll") for (int i = 0; i < NL; i++) t.l[i] = P::BIAS_N[i] - a.l[i];
    return carry(t);
  }
  static JJ_DEV Fe dbl(const Fe& a) { return add(a, a); }  // reference Fr::double src/fr.rs:261-263

  // ---------------------------------------------------------------- canonical form
  // exact sequential carry: limbs < 2^32 in, limbs < 2^29 out (top limb takes the rest)
  static JJ_DEV Fe carry_full(const Fe& a) {
    Fe r; u32 c = 0;
    _Pragma("unroll") for (int i = 0; i < NL - 1; i++) { u32 t = a.l[i] + c; r.l[i] = t & LMASK; c = t >> LB; }
    r.l[NL - 1] = a.l[NL - 1] + c;
    return r;
  }
  // a normalized (limbs < 2^29), value < 2p  ->  value mod p in [0, p)
  static JJ_DEV Fe cond_sub_p(const Fe& a) {
    Fe d; int32_t borrow = 0;
    _Pragma("unroll") for (int i = 0; i < NL; i++) {
      int32_t t = (int32_t)a.l[i] - (int32_t)P::P[i] + borrow;
      d.l[i] = (u32)t & LMASK;
      borrow = t >> LB;  // arithmetic: 0 or -1
    }
    // borrow == -1  <=> a < p : keep a
    const u32 keep = (u32)borrow;  // all-ones or zero
    Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = (a.l[i] & keep) | (d.l[i] & ~keep);
    return r;
  }
  // Any in-contract element (value < 64p/.., limbs <= 2^31) -> the unique Montgomery representative in [0,p), limbs < 2^29.
  static JJ_DEV Fe canon(const Fe& a) { return cond_sub_p(mul(a, one())); }
  static JJ_DEV bool is_zero_canon(const Fe& a) {
    u32 o = 0; _Pragma("unroll") for (int i = 0; i < NL; i++) o |= a.l[i]; return o == 0;
  }
  static JJ_DEV bool eq_canon(const Fe& a, const Fe& b) {
    u32 o = 0; _Pragma("unroll") for (int i = 0; i < NL; i++) o |= a.l[i] ^ b.l[i]; return o == 0;
  }
  static JJ_DEV bool is_zero(const Fe& a) { return is_zero_canon(canon(a)); }          // reference ct_eq(&zero)
  static JJ_DEV bool eq(const Fe& a, const Fe& b) { return eq_canon(canon(a), canon(b)); }  // reference Fr::ct_eq src/fr.rs:48-55
  // select: mask all-ones -> b, zero -> a (reference conditional_select src/fr.rs:64-73); bit-masking, not v_cndmask
  // The mask is made opaque so hipcc emits one v_bfi_b32 per limb: it would otherwise rebuild a v_cmp +
  // v_cndmask_b32_e32 (VCC) sequence, and on gfx950 a VOP2 v_cndmask that re-reads a VCC written several
  // instructions earlier issues at ~22 cycles instead of 4 (measured: experiments/ubench/ubench2.hip).
  static JJ_DEV Fe select(const Fe& a, const Fe& b, u32 mask) {
    asm("" : "+v"(mask));
    Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = (b.l[i] & mask) | (a.l[i] & ~mask); return r;
  }

  // ---------------------------------------------------------------- wire format: 8 x u32 little-endian canonical words
  // plain 256-bit integer -> 9x29 limbs (no reduction)
  static JJ_DEV Fe unpack(const u32 (&w)[8]) {
    Fe r;
    _Pragma("unroll") for (int i = 0; i < NL; i++) {
      const int bit = LB * i, wi = bit >> 5, sh = bit & 31;
      u32 v = w[wi] >> sh;
      if (sh > 32 - LB && wi + 1 < 8) v |= w[wi + 1] << (32 - sh);
      r.l[i] = v & LMASK;
    }
    return r;
  }
  // limbs < 2^29 (value < 2^256) -> 8 words
  static JJ_DEV void pack(u32 (&w)[8], const Fe& a) {
    _Pragma("unroll") for (int wi = 0; wi < 8; wi++) {
      const int bit = 32 * wi, li = bit / LB, sh = bit % LB;   // word starts inside limb li at offset sh
      u32 v = a.l[li] >> sh;
      const int got = LB - sh;
      if (got < 32 && li + 1 < NL) v |= a.l[li + 1] << got;
      if (got + LB < 32 && li + 2 < NL) v |= a.l[li + 2] << (got + LB);
      w[wi] = v;
    }
  }
  // reference Fr::from_raw src/fr.rs:347-349 : any 256-bit integer -> element (reduced mod p), Montgomery form
  static JJ_DEV Fe from_words(const u32 (&w)[8]) { return mul(unpack(w), konst(P::R2)); }
  // reference Fr::from_bytes src/fr.rs:268-292 : ok iff integer < p
  static JJ_DEV Fe from_words_checked(const u32 (&w)[8], bool& ok) {
    Fe x = unpack(w);
    int32_t borrow = 0;
    _Pragma("unroll") for (int i = 0; i < NL; i++) { int32_t t = (int32_t)x.l[i] - (int32_t)P::P[i] + borrow; borrow = t >> LB; }
    ok = (borrow != 0);
    return mul(x, konst(P::R2));
  }
  // reference Fr::to_bytes src/fr.rs:296-308 : Montgomery -> canonical integer words
  static JJ_DEV void to_words(u32 (&w)[8], const Fe& a) {
    Fe plain; _Pragma("unroll") for (int i = 0; i < NL; i++) plain.l[i] = (i == 0);
    pack(w, cond_sub_p(mul(a, plain)));
  }
  // reference Fr::from_bytes_wide / from_u512 src/fr.rs:312-343 : 512-bit integer mod p
  static JJ_DEV Fe from_words_wide(const u32 (&lo)[8], const u32 (&hi)[8]) {
    return add(mul(unpack(lo), konst(P::R2)), mul(unpack(hi), konst(P::R2_256)));
  }

  // ---------------------------------------------------------------- exponentiation
  // a^e for a public fixed exponent given as 8 x 32-bit words (4-bit fixed windows).
  // reference Fr::pow_vartime src/fr.rs:422-434 (same value; exponent is public so no select needed).
  // a^E for a public compile-time exponent E (32-bit words, little-endian): sliding 4-bit windows over odd powers
  // a, a^3 .. a^15; the window program (squarings before each multiplication, table index) is built at compile time
  // so the run-time loop reads two bytes per window.  Wave-uniform control flow.
  struct PowProg { uint8_t nsq[80]; uint8_t idx[80]; int len; int first; };
  template <int NW>
  static constexpr PowProg make_prog(const u32 (&e)[NW]) {
    PowProg p{};
    bool started = false;
    int pend = 0, i = 32 * NW - 1;
    while (i >= 0) {
      if (!((e[i >> 5] >> (i & 31)) & 1u)) { if (started) pend++; i--; continue; }
      int j = i >= 3 ? i - 3 : 0;
      while (!((e[j >> 5] >> (j & 31)) & 1u)) j++;          // window e[i..j] ends in a set bit
      u32 val = 0;
      for (int b = i; b >= j; b--) val = (val << 1) | ((e[b >> 5] >> (b & 31)) & 1u);
      if (started) { p.nsq[p.len] = (uint8_t)(pend + (i - j + 1)); p.idx[p.len] = (uint8_t)(val >> 1); p.len++; }
      else { p.first = (int)(val >> 1); started = true; }
      pend = 0;
      i = j - 1;
    }
    if (pend) { p.nsq[p.len] = (uint8_t)pend; p.idx[p.len] = 255; p.len++; }
    return p;
  }
  template <int NW, const u32 (&E)[NW]>
  static JJ_DEV Fe pow_const(const Fe& a) {
    static constexpr PowProg prog = make_prog(E);
    Fe tab[8];
    const Fe a2 = sqr(a);
    tab[0] = a;
    for (int i = 1; i < 8; i++) tab[i] = mul(tab[i - 1], a2);
    Fe r = tab[prog.first];
    #pragma unroll 1
    for (int s = 0; s < prog.len; s++) {
      const int nsq = prog.nsq[s], idx = prog.idx[s];
      #pragma unroll 1
      for (int q = 0; q < nsq; q++) r = sqr(r);
      if (idx != 255) r = mul(r, tab[idx]);
    }
    return r;
  }
  // reference Fr::invert src/fr.rs:438-540 : a^(p-2); ok = (a != 0); returns 0 when a == 0
  static JJ_DEV Fe invert(const Fe& a) { return pow_const<8, P::PM2>(a); }
};

typedef Field<FqP> Fq;
typedef Field<FrP> Fr;

}  // namespace jj
