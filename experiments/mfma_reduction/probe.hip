// VERDICT r2 item 9 (time-boxed probe): could the CONSTANT-operand halves of the Montgomery product -- M = T_low * q' mod R and
// M * q -- run on the idle matrix pipe as i8 contractions (v_mfma_i32_16x16x64_i8: a Toeplitz(q) x [lanes] product), while a * b
// stays on the VALU?  The matrix instructions themselves would be nearly free (36 per product and wave, on another pipe); what
// decides is the VALU work AROUND them, which this probe compiles and counts (hipcc -S; tools: count.py):
//   k_valu_reduction   the reduction half as shipped: from the 17 column sums of a * b to the 9 output limbs
//                      (72 v_mad_i64_i32 by the limbs of -q, 9 masks, 17 shifts)
//   k_mfma_overheads   everything the MFMA route needs on the VALU besides the matrix instructions:
//                        (1) T_low (9 signed 29-bit limbs, lazy) -> canonical digits -> 33 bytes packed in 9 words (the B operand)
//                        (2) 48 i32 column sums of the first contraction -> 33 byte digits of M (a carry chain: the sums overlap by 8 bits)
//                        (3) the MFMA output layout (lane holds 4 rows of one column of a 16 x 16 tile) -> one lane per field element:
//                            12 values per lane leave through ds_bpermute (LDS pipe) + 12 selects
//                        (4) the same two steps for the 66 column sums of M * q (the top 33 digits and the carry out of the low 33)
//                        (5) bytes -> 9 limbs of 29 bits, added to the high half of a * b
// The matrix instructions are stubbed by opaque moves: the probe counts the VALU side only.  Result: profiles/r3_mfma_reduction_experiment.txt
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32; typedef int32_t i32; typedef int64_t i64; typedef uint64_t u64;
constexpr int NL = 9, LB = 29; constexpr u32 LMASK = (1u << LB) - 1u;
__device__ __forceinline__ u32 opq(u32 x) { asm volatile("" : "+v"(x)); return x; }
__device__ const i32 NEGQ[NL] = {-1, -0x1fffffff, -0x1ffe5bfe, -0x0dea4020, -0x09a1d805, -0x0ce76020, -0x099d7d48, -0x1da9ca65, -0x0073eda7};   // stand-in constants (any 29-bit values: the instruction count does not depend on them)

// (A) the shipped reduction half: column sums c[0..16] of a*b (as 64-bit accumulators) -> r[0..8]
__global__ void k_valu_reduction(const i64* in, u32* out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  i64 c[17];
  #pragma unroll
  for (int k = 0; k < 17; k++) c[k] = in[k * 4096 + t];
  i32 m[NL]; u32 r[NL];
  i64 acc = 0;
  #pragma unroll
  for (int k = 0; k < 17; k++) {
    acc += c[k];
    #pragma unroll
    for (int i = 0; i < NL; i++) { const int j = k - i; if (i >= k || j < 1 || j >= NL) continue; acc += (i64)m[i] * (i64)NEGQ[j]; }
    if (k < NL) m[k] = (i32)((u32)acc & LMASK); else r[k - NL] = (u32)acc & LMASK;
    acc >>= LB;
  }
  r[NL - 1] = (u32)acc;
  #pragma unroll
  for (int i = 0; i < NL; i++) out[i * 4096 + t] = r[i];
}

// (B) the VALU work around the matrix instructions
__device__ __forceinline__ void limbs_to_words(const u32 (&l)[NL], u32 (&w)[9]) {       // canonicalise (carry chain) + pack 261 bits into 9 words
  u32 d[NL]; i32 cy = 0;
  #pragma unroll
  for (int i = 0; i < NL; i++) { const i32 t = (i32)l[i] + cy; d[i] = (u32)t & LMASK; cy = t >> LB; }
  #pragma unroll
  for (int wi = 0; wi < 9; wi++) {
    const int bit = 32 * wi, li = bit / LB, sh = bit % LB;
    u32 v = li < NL ? d[li] >> sh : 0u;
    const int got = LB - sh;
    if (got < 32 && li + 1 < NL) v |= d[li + 1] << got;
    if (got + LB < 32 && li + 2 < NL) v |= d[li + 2] << (got + LB);
    w[wi] = v;
  }
}
template <int N>
__device__ __forceinline__ void sums_to_bytes(const u32 (&s)[N], u32 (&w)[(N + 3) / 4]) {   // column sums (overlapping by 8 bits) -> packed byte digits
  u32 cy = 0;
  #pragma unroll
  for (int q = 0; q < (N + 3) / 4; q++) {
    u64 acc = cy;
    #pragma unroll
    for (int b = 0; b < 4; b++) if (4 * q + b < N) acc += (u64)s[4 * q + b] << (8 * b);
    w[q] = (u32)acc; cy = (u32)(acc >> 32);
  }
}
template <int NT>   // NT tiles of 16 rows: a lane holds 4 rows of its tile column; gather the 4 * NT * 4 values of "its" field element from 4 lanes
__device__ __forceinline__ void regroup(const u32 (&mine)[4 * NT], u32 (&all)[16 * NT], u32 lane) {
  #pragma unroll
  for (int g = 0; g < 4; g++) {
    const int src = (int)(((lane & 15u) + 16u * g) << 2);
    #pragma unroll
    for (int v = 0; v < 4 * NT; v++) all[(v / 4) * 16 + 4 * g + (v & 3)] = (u32)__builtin_amdgcn_ds_bpermute(src, (int)mine[v]);
  }
}
__global__ void k_mfma_overheads(const u32* in, u32* out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 lane = threadIdx.x & 63u;
  u32 tl[NL], th[NL];
  #pragma unroll
  for (int i = 0; i < NL; i++) { tl[i] = in[i * 4096 + t]; th[i] = in[(9 + i) * 4096 + t]; }
  u32 bop[9];
  limbs_to_words(tl, bop);                                   // (1)
  // ---- 12 x v_mfma_i32_16x16x64_i8 (Toeplitz(q') x bytes): stubbed; each lane receives 12 column sums of the tile layout
  u32 o1[12];
  #pragma unroll
  for (int v = 0; v < 12; v++) o1[v] = opq(bop[v % 9] + v);
  u32 s1[48];
  regroup<3>(o1, s1, lane);                                  // (3)
  u32 s1u[33];
  #pragma unroll
  for (int k = 0; k < 33; k++) s1u[k] = s1[k];
  u32 mw[9];
  sums_to_bytes<33>(s1u, mw);                                // (2)  M as 33 byte digits = the B operand of the second contraction
  mw[8] &= 0x1fu;                                            // mod R = 2^261
  // ---- 24 x v_mfma (Toeplitz(q) x bytes of M, 66 output digits): stubbed
  u32 o2[20];
  #pragma unroll
  for (int v = 0; v < 20; v++) o2[v] = opq(mw[v % 9] + v);
  u32 s2[80];
  regroup<5>(o2, s2, lane);                                  // (4)
  u32 s2u[66];
  #pragma unroll
  for (int k = 0; k < 66; k++) s2u[k] = s2[k];
  u32 pw[17];
  sums_to_bytes<66>(s2u, pw);
  // (5) high half of M q (bits 261 ..) -> 9 limbs, added to the high half of a b
  u32 r[NL];
  #pragma unroll
  for (int i = 0; i < NL; i++) {
    const int bit = 261 + LB * i, wi = bit >> 5, sh = bit & 31;
    u32 v = pw[wi] >> sh;
    if (sh > 32 - LB && wi + 1 < 17) v |= pw[wi + 1] << (32 - sh);
    r[i] = (v & LMASK) + th[i];
  }
  #pragma unroll
  for (int i = 0; i < NL; i++) out[i * 4096 + t] = r[i];
}
