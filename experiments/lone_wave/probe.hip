// Lone-wave latency probe (round 3): how long does one Fq product take when a SIMD holds ONE wave (the regime of the MSM's
// bucket-reduce / fix-up / fold chains and of the small-batch ladders), and does instruction-level parallelism inside the
// product help?  Variants:
//   fips      the shipped product: one running 64-bit accumulator walks the 17 columns (187 instructions, one dependent chain)
//   fips x2   two independent shipped products per lane (what ILP the compiler + hardware extract from two chains)
//   cols      operand scanning into 18 independent column accumulators; only the Montgomery digit -> p_1 term -> carry step of
//             each column is serial (+17 64-bit additions)
// Build: hipcc --offload-arch=gfx950 -O3 -I jubjub_amd/csrc -o experiments/lone_wave/probe experiments/lone_wave/probe.hip
#include "jj_field.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
using namespace jj;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static __device__ __forceinline__ u32 opq(u32 x) { asm("" : "+v"(x)); return x; }

template <bool SQUARE>
static __device__ __forceinline__ Fe mul_cols(const Fe& a_in, const Fe& b_in) {
  Fe a, b;
  #pragma unroll
  for (int i = 0; i < NL; i++) { a.l[i] = opq(a_in.l[i]); b.l[i] = SQUARE ? a.l[i] : opq(b_in.l[i]); }
  i64 t[2 * NL];
  #pragma unroll
  for (int k = 0; k < 2 * NL; k++) t[k] = 0;
  if constexpr (SQUARE) {
    i32 a2[NL];
    #pragma unroll
    for (int i = 0; i < NL; i++) a2[i] = (i32)(a.l[i] << 1);
    #pragma unroll
    for (int i = 0; i < NL; i++) {
      #pragma unroll
      for (int j = i; j < NL; j++) t[i + j] += (i64)(i32)a.l[i] * (i64)(j == i ? (i32)a.l[j] : a2[j]);
    }
  } else {
    #pragma unroll
    for (int i = 0; i < NL; i++) {
      #pragma unroll
      for (int j = 0; j < NL; j++) t[i + j] += (i64)(i32)a.l[i] * (i64)(i32)b.l[j];
    }
  }
  Fe r;
  #pragma unroll
  for (int k = 0; k < NL; k++) {
    const i32 m = (i32)((u32)t[k] & LMASK);
    #pragma unroll
    for (int j = 1; j < NL; j++) t[k + j] += (i64)m * (i64)(-(i32)FqP::P[j]);
    t[k + 1] += t[k] >> LB;
  }
  #pragma unroll
  for (int k = NL; k < 2 * NL - 1; k++) { r.l[k - NL] = (u32)t[k] & LMASK; t[k + 1] += t[k] >> LB; }
  r.l[NL - 1] = (u32)t[2 * NL - 1];
  return r;
}

template <int V>
__global__ void __launch_bounds__(256) k_chain(u32* out, const u32* in, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  Fe x, y, z;
  for (int i = 0; i < NL; i++) { x.l[i] = in[i] + (i == 0 ? (tid & 7) : 0); y.l[i] = in[NL + i]; z.l[i] = in[2 * NL + i]; }
  #pragma unroll 1
  for (int it = 0; it < iters; it++) {
    if constexpr (V == 0) { x = Fq::mul(x, y); }
    if constexpr (V == 1) { x = Fq::mul(x, y); z = Fq::mul(z, y); }
    if constexpr (V == 2) { x = mul_cols<false>(x, y); }
    if constexpr (V == 3) { x = Fq::sqr(x); }
    if constexpr (V == 4) { x = mul_cols<true>(x, x); }
    if constexpr (V == 5) { x = mul_cols<false>(x, y); z = mul_cols<false>(z, y); }
  }
  for (int i = 0; i < NL; i++) out[(size_t)tid * 2 * NL + i] = x.l[i], out[(size_t)tid * 2 * NL + NL + i] = z.l[i];
}

typedef void (*kern_t)(u32*, const u32*, int);
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  u32 *out, *in; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 2 * NL * 4)); CK(hipMalloc(&in, 3 * NL * 4));
  u32 h[3 * NL];
  for (int i = 0; i < 3 * NL; i++) h[i] = (0x12345u * (i + 3) + 0x9e3779u * i) & LMASK;
  h[NL - 1] &= 0xffff; h[2 * NL - 1] &= 0xffff; h[3 * NL - 1] &= 0xffff;
  CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct { const char* name; kern_t fn; int prods; int ref; } v[] = {
    {"mul  fips (shipped)", k_chain<0>, 1, -1}, {"mul  fips x2 chains", k_chain<1>, 2, -1}, {"mul  cols", k_chain<2>, 1, 0},
    {"mul  cols x2 chains", k_chain<5>, 2, 1}, {"sqr  fips (shipped)", k_chain<3>, 1, -1}, {"sqr  cols", k_chain<4>, 1, 4}};
  const int iters = 4000;
  const size_t words = (size_t)cus * 256 * 2 * NL;
  u32* ref[6] = {0}; u32* got = (u32*)malloc(words * 4);
  printf("ns per product per wave (dependent chains of %d products; 256-thread blocks, W = waves per SIMD)\n", iters);
  printf("%-22s %10s %10s %10s %10s   check\n", "variant", "W=1", "W=2", "W=3", "W=4");
  for (int q = 0; q < 6; q++) {
    printf("%-22s", v[q].name);
    for (int wps = 1; wps <= 4; wps++) {
      const int blocks = cus * wps;
      v[q].fn<<<blocks, 256>>>(out, in, 10); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      v[q].fn<<<blocks, 256>>>(out, in, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf(" %10.1f", ms * 1e6 / iters / v[q].prods);
      if (wps == 1) { CK(hipMemcpy(got, out, words * 4, hipMemcpyDeviceToHost)); ref[q] = (u32*)malloc(words * 4); memcpy(ref[q], got, words * 4); }
    }
    // the cols variants must give the same residue class; limbs may differ only if the digit sets differ -- they do not (same m_k)
    if (v[q].ref >= 0) printf("   %s", memcmp(ref[q], ref[v[q].ref], words * 4) == 0 ? "same limbs as fips" : "DIFFERENT");
    printf("\n");
  }
  return 0;
}
