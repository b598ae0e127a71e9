// Probe: instruction count of a signed ("subtractive") Montgomery product on 9 x 29-bit limbs vs the round-1 additive one.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps -c probe.hip ; count instructions in the .s
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32; typedef uint64_t u64; typedef int64_t i64; typedef int32_t i32;
constexpr int NL = 9, LB = 29; constexpr u32 LMASK = 0x1fffffffu;
__device__ constexpr u32 P[9] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
struct Fe { u32 l[NL]; };
#define DEV __device__ __forceinline__
template <bool SQUARE, int PIN>
static DEV Fe mul_s(const Fe& a, const Fe& b) {
  i32 m[NL]; i32 b2[NL];
  if constexpr (SQUARE) { _Pragma("unroll") for (int i = 0; i < NL; i++) b2[i] = (i32)a.l[i] << 1; }
  Fe r; i64 acc = 0; u32 tok = 0;
#define PINA(x) do { if constexpr (PIN) asm("" : "+s"(tok) : "v"(x)); } while (0)
  _Pragma("unroll") for (int k = 0; k < 2 * NL - 1; k++) {
    _Pragma("unroll") for (int i = 0; i < NL; i++) {
      const int j = k - i; if (j < 0 || j >= NL) continue;
      if constexpr (SQUARE) {
        if (j > i) { acc += (i64)(i32)a.l[i] * b2[j]; PINA(acc); }
        else if (j == i) { acc += (i64)(i32)a.l[i] * (i32)a.l[i]; PINA(acc); }
      } else { acc += (i64)(i32)a.l[i] * (i32)b.l[j]; PINA(acc); }
    }
    _Pragma("unroll") for (int i = 0; i < NL; i++) {
      const int j = k - i; if (i >= k || j < 1 || j >= NL) continue;
      acc += (i64)m[i] * (-(i32)P[j]); PINA(acc);
    }
    if (k < NL) m[k] = (i32)((u32)acc & LMASK);
    else r.l[k - NL] = (u32)acc & LMASK;
    acc >>= LB;
  }
  r.l[NL - 1] = (u32)acc;
  if constexpr (PIN) asm volatile("" :: "s"(tok));
  return r;
}
template <int PIN>
__global__ void __launch_bounds__(256) k_chain(u32* io, int iters) {
  Fe a, b;
  for (int i = 0; i < NL; i++) { a.l[i] = io[threadIdx.x * 18 + i]; b.l[i] = io[threadIdx.x * 18 + 9 + i]; }
  #pragma unroll 1
  for (int it = 0; it < iters; it++) { a = mul_s<false, PIN>(a, b); b = mul_s<true, PIN>(b, b); }
  for (int i = 0; i < NL; i++) io[threadIdx.x * 18 + i] = a.l[i] ^ b.l[i];
}
template __global__ void k_chain<0>(u32*, int);
template __global__ void k_chain<1>(u32*, int);
