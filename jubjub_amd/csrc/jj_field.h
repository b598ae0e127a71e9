// Jubjub prime-field arithmetic for CDNA4 (gfx950) — device code.
//
// Representation (MI355X-first, NOT the reference's 4x64 layout): one field element per lane, held in registers as
// 9 limbs x 29 bits with SIGNED 32-bit limbs ("reduced radix"), Montgomery form with R = 2^261, and a signed,
// subtractive Montgomery reduction.
//
// Why 29-bit limbs: measured on MI355X (experiments/ubench, profiles/ubench_r1.txt) v_mad_{u,i}64_{u,i}32 issues at the
// same ~4.3-cycle/wave64 rate as every other VOP3 instruction, but any multi-word carry chain goes through an SGPR
// carry (v_add_co/v_addc_co), which on gfx940/950 costs extra wait states per link.  With 29-bit limbs a 9x9
// schoolbook column (<= 9 a*b + 9 m*p products of < 2^59) never overflows a 64-bit VGPR pair, so a product needs no
// carry flag at all.
//
// Why signed (round 2): the round-1 additive form (c + m*p, m = -c mod 2^29) needed per reduction column a negate, a
// mask, a multiply-add by p_0 = 1 and a 64-bit shift.  Subtracting instead (c - m*p with m = c mod 2^29: a mask only)
// leaves the low 29 bits of the column zero by construction, so the column step is {v_and_b32, v_ashrrev_i64} and the
// m*p_0 multiply-add disappears: a product is 153 v_mad_i64_i32 (81 a*b + 72 m*(-p_j)) + 34 other VALU instructions
// (round 1: 162 + 43), a square 117 + 8 + 34.  The price is that values and limbs are signed: a Montgomery output is
// v = (a*b - M*p)/R in (a*b/R - p, a*b/R], limbs 0..7 in [0, 2^29) and a small signed top limb.  That also makes
// subtraction a plain limb-wise v_sub_u32 (no bias constant, no carry step): the additive part of a point doubling drops
// from ~135 to ~70 VALU instructions.  experiments/signed_mont/probe.hip holds the instruction-count probe.
//
// What it restates: results are the same canonical residues as the reference's Fq/Fr (reference src/fr.rs:246-665 is
// the template for both fields; Fq = bls12_381::Scalar).  Only canonical little-endian 32-byte encodings cross the
// kernel boundary (reference Fr::to_bytes src/fr.rs:296-308, from_bytes 268-292).
//
// Bounds contract (tools/bounds_check.py replays every formula with interval arithmetic; tests/test_emu_field.py runs
// the same header on the host with a 128-bit shadow accumulator):
//   "N" : limbs 0..7 in [0, 2^29), top limb signed and small, value in (-1.2p, 0.2p)     -- output of mul/sqr
//   lazy: limb-wise sums/differences of a few N's; |limb| < 2^31 always
//   mul/sqr: every 64-bit column accumulator stays inside (-2^63, 2^63):  9 * max|a_i| * max|b_j| + 9 * 2^58 + carry
#pragma once
// Tuning and probe macros (window widths, workgroup sizes, register caps, buffering schemes, the probe builds that compute WRONG results on purpose)
// belong to experiments (experiments/, tools/fixedbase_floor.py): the shipped library is built with the defaults, and a build that overrides one
// must say so with -DJJ_EXPERIMENTS.
#if !defined(JJ_EXPERIMENTS) && (defined(JJ_MUL_PIN) || defined(JJ_OPAQUE_MODE) || defined(JJ_VB_MINWAVES) || defined(JJ_VB_W) || defined(JJ_VB_PROBE_SHARED_READS) || \
    defined(JJ_FB_THREADS) || defined(JJ_FB_SINGLE_BUFFER) || defined(JJ_FBC_THREADS) || defined(JJ_FBC_SINGLE_BUFFER) || defined(JJ_FBC_PROBE) || \
    defined(JJ_MSM_SORT_UNROLL) || defined(JJ_MSM_P2_THREADS) || defined(JJ_MSM_ACC_MINBLOCKS))
#error "tuning / probe macros (JJ_VB_W, JJ_FB_THREADS, JJ_FBC_PROBE, ...) need -DJJ_EXPERIMENTS: the shipped library is built with the defaults"
#endif
#ifndef JJ_HOST_EMU
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include "jj_constants.h"

namespace jj {

#ifdef JJ_HOST_EMU
// Host emulation of the device arithmetic for CPU-side tests (tests/cpp/emu_field.cpp): same code, no inline asm,
// and every accumulator update is mirrored in 128 bits to catch a 64-bit overflow that the static checker missed.
#define JJ_DEV inline
extern "C" void jj_emu_overflow(const char* what);
#define JJ_EMU_ACC_DECL __int128 acc_shadow = 0
#define JJ_EMU_ACC_MAD(x, y) do { acc_shadow += (__int128)(x) * (__int128)(y); if (acc_shadow != (__int128)acc) jj_emu_overflow("column accumulator"); } while (0)
#define JJ_EMU_ACC_SHIFT() do { acc_shadow >>= LB; } while (0)
#else
#define JJ_DEV __device__ __forceinline__
#define JJ_EMU_ACC_DECL
#define JJ_EMU_ACC_MAD(x, y)
#define JJ_EMU_ACC_SHIFT()
#endif

typedef int32_t i32;
typedef int64_t i64;

struct Fe {
  u32 l[NL];   // two's-complement signed limbs; value = sum (i32)l[i] * 2^(29 i)
};

// JJ_MUL_PIN (default of Field's PIN parameter): how mul/sqr keep LLVM's reassociation pass from moving the column carry
// to the END of each column's sum, which costs one 64-bit add per column (17 per product):
//   0 = let it;  1 = pass every partial sum through __builtin_annotation (llvm.annotation): an identity intrinsic that the
//   middle-end treats as an opaque call (so the add tree is never linearised and every multiply-add keeps its running
//   sum as the addend) and that instruction selection drops without a trace -- no code, no scheduling artefact.
//   (Round 1 pinned with empty asm statements reading the sum; the scheduler was free to sink those reads, which kept
//   every partial sum alive and drove some kernels to 512 VGPRs + scratch.)
#ifndef JJ_MUL_PIN
#define JJ_MUL_PIN 1
#endif
// JJ_OPAQUE_MODE: how the operands of a product are hidden from value-range reasoning (see mul_fips):
//   1 = an empty asm statement per limb ("+v": no code; hipcc adds an s_nop only when the very next instruction reads
//   the register, 0-6 per loop body in the shipped kernels);  0 = llvm.annotation, which hides the range from the
//   middle-end only -- instruction selection still sees it through live-out information and expands mixed products.
#ifndef JJ_OPAQUE_MODE
#define JJ_OPAQUE_MODE 1
#endif
template <class P, int PIN = JJ_MUL_PIN>
struct Field {
  static constexpr u32 PINV = (0u - P::NINV) & LMASK;   // p^-1 mod 2^29 (1 for Fq: p = 1 mod 2^32)

  // ---------------------------------------------------------------- constants
  static JJ_DEV Fe one() { Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = P::ONE[i]; return r; }
  static JJ_DEV Fe zero() { Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = 0; return r; }
  template <int N>
  static JJ_DEV Fe konst(const u32 (&c)[N]) { Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = c[i]; return r; }

  // ---------------------------------------------------------------- Montgomery product
  // Finely-integrated product scanning with signed accumulation.  One running 64-bit accumulator walks the 17 columns;
  // the carry out of column k is the addend of the first multiply-add of column k+1, so carry propagation costs no
  // instruction.  Per column: the a_i*b_j terms, the -m_i*p_j terms, then (k < 9) the Montgomery digit
  // m_k = acc mod 2^29 (after which acc - m_k*p_0 has 29 zero low bits, p_0 * p^-1 = 1) or (k >= 9) emit a limb; shift.
  // r = (a*b - M*p)/R, M = sum m_k 2^(29k) in [0, R):  r = a*b/R mod p, value in (a*b/R - p, a*b/R].
  // DOUBLE: returns 2*a*b/R-class value of 2*a^2 (squares only): cross terms use 4a_j, the diagonal 2a_i.
  // reference Fr::mul src/fr.rs:592-616 + montgomery_reduce 544-588 (same residue, different digit set).
#ifdef JJ_HOST_EMU
#define JJ_OPAQUE(x) (x)
#else
#if JJ_OPAQUE_MODE == 1
static __device__ __forceinline__ u32 jj_opaque_asm(u32 x) { asm("" : "+v"(x)); return x; }
#define JJ_OPAQUE(x) jj_opaque_asm(x)
#else
#define JJ_OPAQUE(x) ((u32)__builtin_annotation((u32)(x), "jj"))
#endif
#endif
  template <bool SQUARE, bool DOUBLE, bool OPQ = true>
  static JJ_DEV Fe mul_fips(const Fe& a_in, const Fe& b_in) {
    i32 m[NL];
    i32 b2[NL], b4[NL];
    // The operands go through the same identity intrinsic: what the middle-end may know about their sign (masked limbs
    // are non-negative) must not reach the multiplications, or it rewrites sext(x) as zext(x) and, where instruction
    // selection cannot re-derive the range (values carried around a loop, selects), a signed x unsigned 64-bit product
    // is expanded into two v_mad_u64_u32 and two moves instead of one v_mad_i64_i32.
    Fe a, b;
    _Pragma("unroll") for (int i = 0; i < NL; i++) {
      if constexpr (OPQ) { a.l[i] = JJ_OPAQUE(a_in.l[i]); b.l[i] = SQUARE ? a.l[i] : JJ_OPAQUE(b_in.l[i]); }
      else { a.l[i] = a_in.l[i]; b.l[i] = SQUARE ? a.l[i] : b_in.l[i]; }
    }
    if constexpr (SQUARE) {
      _Pragma("unroll") for (int i = 0; i < NL; i++) b2[i] = (i32)(a.l[i] << 1);
      if constexpr (DOUBLE) { _Pragma("unroll") for (int i = 0; i < NL; i++) b4[i] = (i32)(a.l[i] << 2); }
    }
    Fe r;
    i64 acc = 0;
    JJ_EMU_ACC_DECL;
#ifdef JJ_HOST_EMU
#define JJ_PIN(x)
#else
#define JJ_PIN(x) do { if constexpr (PIN == 1) x = __builtin_annotation(x, "jj"); } while (0)
#endif
#define JJ_MAD(x, y) do { acc += (i64)(i32)(x) * (i64)(i32)(y); JJ_EMU_ACC_MAD((i32)(x), (i32)(y)); JJ_PIN(acc); } while (0)
    _Pragma("unroll") for (int k = 0; k < 2 * NL - 1; k++) {
      _Pragma("unroll") for (int i = 0; i < NL; i++) {
        const int j = k - i;
        if (j < 0 || j >= NL) continue;
        if constexpr (SQUARE) {
          if (j > i) { if constexpr (DOUBLE) JJ_MAD(a.l[i], b4[j]); else JJ_MAD(a.l[i], b2[j]); }
          else if (j == i) { if constexpr (DOUBLE) JJ_MAD(a.l[i], b2[i]); else JJ_MAD(a.l[i], a.l[i]); }
        } else {
          JJ_MAD(a.l[i], b.l[j]);
        }
      }
      _Pragma("unroll") for (int i = 0; i < NL; i++) {
        const int j = k - i;
        if (i >= k || i >= NL || j < 1 || j >= NL) continue;   // m_i exists for i < min(k, 9); p_0 handled below
        JJ_MAD(m[i], -(i32)P::P[j]);
      }
      if (k < NL) {
        u32 mk;
        if constexpr (PINV == 1u) mk = (u32)acc & LMASK;
        else mk = ((u32)acc * PINV) & LMASK;
        m[k] = (i32)mk;
        if constexpr (P::P[0] != 1u) JJ_MAD(mk, -(i32)P::P[0]);   // Fq: p_0 = 1, acc - m_k only clears the bits the shift drops
      } else {
        r.l[k - NL] = (u32)acc & LMASK;
      }
      acc >>= LB;   // arithmetic
      JJ_EMU_ACC_SHIFT();
    }
    r.l[NL - 1] = (u32)acc;
#ifdef JJ_HOST_EMU
    if (acc != (i64)(i32)acc) jj_emu_overflow("top limb");
#endif
#undef JJ_MAD
#undef JJ_PIN
    return r;
  }
  static JJ_DEV Fe mul(const Fe& a, const Fe& b) { return mul_fips<false, false>(a, b); }
  // A value that enters SEVERAL products is hidden once (opaque) and then multiplied as it is (mul_hidden): the "+v" constraint of the
  // empty asm ties input and output to one register, so hiding a value that has another use left costs a v_mov per limb
  // (36 per point operation when the four completed coordinates were hidden once per product).
  static JJ_DEV Fe opaque(const Fe& a) { Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = JJ_OPAQUE(a.l[i]); return r; }
  static JJ_DEV Fe mul_hidden(const Fe& a, const Fe& b) { return mul_fips<false, false, false>(a, b); }
  static JJ_DEV Fe sqr_hidden(const Fe& a) { return mul_fips<true, false, false>(a, a); }
  // r = a*a/R mod p.  reference Fr::square src/fr.rs:353-381 (same cross-term doubling idea).
  static JJ_DEV Fe sqr(const Fe& a) { return mul_fips<true, false>(a, a); }
  // r = 2*a*a/R mod p in one product (the doubling's 2Z^2)
  static JJ_DEV Fe sqr2(const Fe& a) { return mul_fips<true, true>(a, a); }

  // ---------------------------------------------------------------- additive ops (lazy, carry-free, signed limbs)
  // reference Fr::add src/fr.rs:638-647 / sub 620-634 / neg 651-665 / double 261-263 (which reduce; we defer).
  static JJ_DEV Fe add(const Fe& a, const Fe& b) {
    Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i]; return r;
  }
  static JJ_DEV Fe sub(const Fe& a, const Fe& b) {
    Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = a.l[i] - b.l[i]; return r;
  }
  static JJ_DEV Fe neg(const Fe& a) {
    Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = 0u - a.l[i]; return r;
  }
  static JJ_DEV Fe dbl(const Fe& a) { return add(a, a); }
  // r = 2a - b in one pass
  static JJ_DEV Fe dbl_sub(const Fe& a, const Fe& b) {
    Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = (a.l[i] << 1) - b.l[i]; return r;
  }
  // conditional negation: mask all-ones -> -a, zero -> a   ((a ^ mask) - mask, two VOP2 per limb)
  static JJ_DEV Fe cneg(const Fe& a, u32 mask) {
#ifndef JJ_HOST_EMU
    asm("" : "+v"(mask));
#endif
    Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = (a.l[i] ^ mask) - mask; return r;
  }
  // one parallel carry step: |limb| < 2^31 in; limbs 1..7 in [-4, 2^29 + 4), limb 0 in [0, 2^29) out; value unchanged.
  static JJ_DEV Fe carry(const Fe& a) {
    Fe r;
    r.l[0] = a.l[0] & LMASK;
    _Pragma("unroll") for (int i = 1; i < NL - 1; i++) r.l[i] = (a.l[i] & LMASK) + (u32)((i32)a.l[i - 1] >> LB);
    r.l[NL - 1] = a.l[NL - 1] + (u32)((i32)a.l[NL - 2] >> LB);
    return r;
  }
  // exact sequential carry: limbs 0..7 in [0, 2^29) out, the top limb takes the rest (sign of the value = sign of the top limb)
  static JJ_DEV Fe carry_full(const Fe& a) {
    Fe r; i32 c = 0;
    _Pragma("unroll") for (int i = 0; i < NL - 1; i++) { const i32 t = (i32)a.l[i] + c; r.l[i] = (u32)t & LMASK; c = t >> LB; }
    r.l[NL - 1] = a.l[NL - 1] + (u32)c;
    return r;
  }
  // select: mask all-ones -> b, zero -> a (reference conditional_select src/fr.rs:64-73); bit-masking, not v_cndmask.
  // The mask is made opaque so hipcc emits one v_bfi_b32 per limb: it would otherwise rebuild a v_cmp +
  // v_cndmask_b32_e32 (VCC) sequence, and on gfx950 a VOP2 v_cndmask that re-reads a VCC written several
  // instructions earlier issues at ~22 cycles instead of 4 (measured: experiments/ubench/ubench2.hip).
  static JJ_DEV Fe select(const Fe& a, const Fe& b, u32 mask) {
#ifndef JJ_HOST_EMU
    asm("" : "+v"(mask));
#endif
    Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = (b.l[i] & mask) | (a.l[i] & ~mask); return r;
  }

  // ---------------------------------------------------------------- canonical forms and predicates
  // Montgomery form -> the canonical plain integer in [0, p), limbs in [0, 2^29).  a*1/R = a/R lies in (-1, 1) for any
  // in-contract a, so the product lands in [-p, 0]: add p when negative (reference Fr::to_bytes src/fr.rs:296-308).
  static JJ_DEV Fe to_plain(const Fe& a) {
    Fe one_plain; _Pragma("unroll") for (int i = 0; i < NL; i++) one_plain.l[i] = (i == 0);
    const Fe t = mul(a, one_plain);                       // digits: limbs 0..7 in [0, 2^29), top limb signed
    const u32 negm = (u32)((i32)t.l[NL - 1] >> 31);       // all-ones iff value < 0
    Fe s; _Pragma("unroll") for (int i = 0; i < NL; i++) s.l[i] = t.l[i] + (P::P[i] & negm);
    return carry_full(s);
  }
  // a == 0 mod p, for any in-contract a (reference ct_eq(&zero)): a/R-product is 0 or -p
  static JJ_DEV bool is_zero(const Fe& a) {
    Fe one_plain; _Pragma("unroll") for (int i = 0; i < NL; i++) one_plain.l[i] = (i == 0);
    const Fe t = mul(a, one_plain);
    u32 o0 = 0, o1 = 0;
    _Pragma("unroll") for (int i = 0; i < NL; i++) { o0 |= t.l[i]; o1 |= t.l[i] ^ P::NEGP_DIGITS[i]; }
    return o0 == 0 || o1 == 0;
  }
  static JJ_DEV bool eq(const Fe& a, const Fe& b) { return is_zero(sub(a, b)); }   // reference Fr::ct_eq src/fr.rs:48-55
  // The same test without the product, for a value that is already in product form (a mul/sqr output or a canonical
  // constant: limbs 0..7 in [0, 2^29), value in (-2p, p)): its digits are unique, so it is 0 mod p iff they are 0 or -p.
  static JJ_DEV bool is_zero_product(const Fe& t) {
    u32 o0 = 0, o1 = 0;
    _Pragma("unroll") for (int i = 0; i < NL; i++) { o0 |= t.l[i]; o1 |= t.l[i] ^ P::NEGP_DIGITS[i]; }
    return o0 == 0 || o1 == 0;
  }
  // canonical integer in [0, p) of a PLAIN (non-Montgomery) value held in product form, value in (-2p, p): two conditional
  // additions of p (a product by a plain-form operand lands there directly, which saves the extra product of to_plain)
  static JJ_DEV Fe canon_plain_product(const Fe& a) {
    Fe w = a;
    _Pragma("unroll") for (int rep = 0; rep < 2; rep++) {
      const u32 negm = (u32)((i32)w.l[NL - 1] >> 31);
      Fe s; _Pragma("unroll") for (int i = 0; i < NL; i++) s.l[i] = w.l[i] + (P::P[i] & negm);
      w = carry_full(s);
    }
    return w;
  }
  static JJ_DEV Fe plain_one() { Fe r; _Pragma("unroll") for (int i = 0; i < NL; i++) r.l[i] = (i == 0); return r; }
  // Montgomery form -> the Montgomery-form representative in [0, p) with limbs in [0, 2^29) (table construction only)
  static JJ_DEV Fe canon(const Fe& a) {
    Fe w = mul(a, one());                                   // value in (-1.2p, 0.2p)
    _Pragma("unroll") for (int rep = 0; rep < 2; rep++) {
      const u32 negm = (u32)((i32)w.l[NL - 1] >> 31);
      Fe s; _Pragma("unroll") for (int i = 0; i < NL; i++) s.l[i] = w.l[i] + (P::P[i] & negm);
      w = carry_full(s);
    }
    return w;
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
  // limbs in [0, 2^29) (value < 2^256) -> 8 words
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
    i32 borrow = 0;
    _Pragma("unroll") for (int i = 0; i < NL; i++) { i32 t = (i32)x.l[i] - (i32)P::P[i] + borrow; borrow = t >> LB; }
    ok = (borrow != 0);
    return mul(x, konst(P::R2));
  }
  // reference Fr::to_bytes src/fr.rs:296-308 : Montgomery -> canonical integer words
  static JJ_DEV void to_words(u32 (&w)[8], const Fe& a) { pack(w, to_plain(a)); }
  // reference Fr::from_bytes_wide / from_u512 src/fr.rs:312-343 : 512-bit integer mod p
  static JJ_DEV Fe from_words_wide(const u32 (&lo)[8], const u32 (&hi)[8]) {
    return add(mul(unpack(lo), konst(P::R2)), mul(unpack(hi), konst(P::R2_256)));
  }

  // ---------------------------------------------------------------- exponentiation
  // a^E for a public compile-time exponent E (32-bit words, little-endian): sliding 4-bit windows over odd powers
  // a, a^3 .. a^15; the window program (squarings before each multiplication, table index) is built at compile time
  // so the run-time loop reads two bytes per window.  Wave-uniform control flow.
  // reference Fr::pow_vartime src/fr.rs:422-434 (same value; the exponent is public so no select is needed).
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
  // The eight odd powers stay in named registers (72 VGPRs) and the wave-uniform window index picks one through a
  // uniform switch: an indexed `Fe tab[8]` would be placed in scratch memory (round 1: 304 B/lane of scratch traffic
  // in the decoder, the normaliser and the pairing kernel).
  template <int NW, const u32 (&E)[NW]>
  static JJ_DEV Fe pow_const(const Fe& a) {
    static constexpr PowProg prog = make_prog(E);
    const Fe a2 = sqr(a);
    const Fe t0 = a, t1 = mul(t0, a2), t2 = mul(t1, a2), t3 = mul(t2, a2), t4 = mul(t3, a2), t5 = mul(t4, a2), t6 = mul(t5, a2), t7 = mul(t6, a2);
    auto pick = [&](int idx) -> Fe {
      switch (idx) {
        case 0: return t0; case 1: return t1; case 2: return t2; case 3: return t3;
        case 4: return t4; case 5: return t5; case 6: return t6; default: return t7;
      }
    };
    Fe r = pick(prog.first);
    #pragma unroll 1
    for (int s = 0; s < prog.len; s++) {
      const int nsq = prog.nsq[s], idx = prog.idx[s];
      #pragma unroll 1
      for (int q = 0; q < nsq; q++) r = sqr(r);
      if (idx != 255) r = mul(r, pick(idx));
    }
    return r;
  }
  // reference Fr::invert src/fr.rs:438-540 : a^(p-2); returns 0 when a == 0
  static JJ_DEV Fe invert(const Fe& a) { return pow_const<8, P::PM2>(a); }
};

typedef Field<FqP> Fq;
typedef Field<FrP> Fr;

}  // namespace jj
