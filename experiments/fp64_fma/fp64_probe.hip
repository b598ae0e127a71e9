// Experiment: is an FP64-FMA limb product cheaper than the 29-bit integer one on gfx950?
//
// v_fma_f64 issues at the same ~4.2 cycles per wave64 as v_mad_{u,i}64_{u,i}32 (profiles/ubench_r1.txt) and a pair of FMAs
// splits a 52 x 52-bit product exactly into its high and low 52 bits (round-toward-zero trick:  hi = fma(a, b, 2^104),
// lo = fma(a, b, (2^104 + 2^52) - hi) ), so 5 x 5 limb products cover 260 bits with 50 FMAs where the integer path needs
// 81 multiply-adds for 9 x 9.  What the FMA does NOT do is accumulate: each product needs a subtraction to form the second
// addend and each half needs a 64-bit integer add of its bit pattern into the column sum (Emmart et al.), and 64-bit
// adds are full-price VOP3 instructions here.  This probe builds both a*b column kernels, checks that they produce the
// same 512-bit product, and times them:   hipcc --offload-arch=gfx950 -O3 fp64_probe.hip -o fp64_probe && ./fp64_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64; typedef int64_t i64; typedef int32_t i32;

// ---- integer: 9 limbs x 29 bits, 17 column sums (the a*b half of jj_field.h's product)
__global__ void __launch_bounds__(256) k_int(const u32* in, u64* out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  u32 a[9], b[9];
  for (int i = 0; i < 9; i++) { a[i] = in[t * 18 + i]; b[i] = in[t * 18 + 9 + i]; }
  u64 acc[17];
  for (int k = 0; k < 17; k++) acc[k] = 0;
  #pragma unroll 1
  for (int it = 0; it < iters; it++) {
    _Pragma("unroll") for (int k = 0; k < 17; k++) {
      u64 c = acc[k];
      _Pragma("unroll") for (int i = 0; i < 9; i++) { const int j = k - i; if (j < 0 || j >= 9) continue; c = __builtin_annotation((i64)(c + (u64)a[i] * b[j]), "p"); }
      acc[k] = c;
    }
    if (iters > 1) { _Pragma("unroll") for (int i = 0; i < 9; i++) a[i] = (a[i] ^ (u32)acc[i]) & 0x1fffffffu; }   // serialise the iterations
  }
  for (int k = 0; k < 17; k++) out[t * 17 + k] = acc[k];
}
// ---- FP64: 5 limbs x 52 bits held as doubles, hi/lo split per product, integer accumulation of the bit patterns
__global__ void __launch_bounds__(256) k_fp(const u64* in, u64* out, int iters) {
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");            // double-precision rounding mode: toward zero
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double a[5], b[5];
  for (int i = 0; i < 5; i++) { a[i] = (double)in[t * 10 + i]; b[i] = (double)in[t * 10 + 5 + i]; }
  const double c1 = 0x1.0p104, c2 = 0x1.0p104 + 0x1.0p52;
  u64 lo[10], hi[10];
  for (int k = 0; k < 10; k++) lo[k] = hi[k] = 0;
  #pragma unroll 1
  for (int it = 0; it < iters; it++) {
    _Pragma("unroll") for (int k = 0; k < 9; k++) {
      _Pragma("unroll") for (int i = 0; i < 5; i++) {
        const int j = k - i; if (j < 0 || j >= 5) continue;
        const double ph = __builtin_fma(a[i], b[j], c1);
        const double pl = __builtin_fma(a[i], b[j], c2 - ph);
        lo[k] += (u64)__double_as_longlong(pl);
        hi[k + 1] += (u64)__double_as_longlong(ph);
      }
    }
    if (iters > 1) { _Pragma("unroll") for (int i = 0; i < 5; i++) a[i] = (double)((lo[i] ^ hi[i]) & 0xfffffffffffffull); }
  }
  for (int k = 0; k < 10; k++) { out[t * 20 + k] = lo[k]; out[t * 20 + 10 + k] = hi[k]; }
}

static void add_at(std::vector<u32>& acc, u64 v, int bit) {          // acc += v << bit   (little-endian 32-bit words)
  unsigned __int128 x = (unsigned __int128)v << (bit & 31);
  int w = bit >> 5;
  u64 carry = 0;
  for (int q = 0; q < 4 || carry; q++, w++) { u64 s = (u64)acc[w] + (u32)(x >> (32 * q)) * (q < 4) + carry; acc[w] = (u32)s; carry = s >> 32; }
}
int main() {
  const int blocks = 256 * 8, threads = 256, n = blocks * threads;
  std::vector<u32> hin((size_t)n * 18); std::vector<u64> hin52((size_t)n * 10);
  u64 s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (int t = 0; t < n; t++) {
    u32 wa[9], wb[9];                                                   // two 256-bit integers (< 2^256) in both limb forms
    unsigned char A[33] = {0}, B[33] = {0};
    for (int i = 0; i < 32; i++) { A[i] = (unsigned char)rnd(); B[i] = (unsigned char)rnd(); }
    auto bits = [](const unsigned char* p, int from, int cnt) { u64 v = 0; for (int q = 0; q < cnt; q++) { int bpos = from + q; if (bpos < 256) v |= (u64)((p[bpos >> 3] >> (bpos & 7)) & 1) << q; } return v; };
    for (int i = 0; i < 9; i++) { wa[i] = (u32)bits(A, 29 * i, 29); wb[i] = (u32)bits(B, 29 * i, 29); hin[(size_t)t * 18 + i] = wa[i]; hin[(size_t)t * 18 + 9 + i] = wb[i]; }
    for (int i = 0; i < 5; i++) { hin52[(size_t)t * 10 + i] = bits(A, 52 * i, 52); hin52[(size_t)t * 10 + 5 + i] = bits(B, 52 * i, 52); }
  }
  u32* din; u64 *din52, *dout, *dout2;
  hipMalloc(&din, hin.size() * 4); hipMalloc(&din52, hin52.size() * 8); hipMalloc(&dout, (size_t)n * 17 * 8); hipMalloc(&dout2, (size_t)n * 20 * 8);
  hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice); hipMemcpy(din52, hin52.data(), hin52.size() * 8, hipMemcpyHostToDevice);
  // ---- correctness: one product each, recombined on the host
  hipLaunchKernelGGL(k_int, dim3(blocks), dim3(threads), 0, 0, din, dout, 1);
  hipLaunchKernelGGL(k_fp, dim3(blocks), dim3(threads), 0, 0, din52, dout2, 1);
  std::vector<u64> o1((size_t)n * 17), o2((size_t)n * 20);
  hipMemcpy(o1.data(), dout, o1.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), dout2, o2.size() * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 4096; t++) {
    std::vector<u32> x(24, 0), y(24, 0);
    for (int k = 0; k < 17; k++) add_at(x, o1[(size_t)t * 17 + k], 29 * k);
    const u64 M52 = 0xfffffffffffffull;
    for (int k = 0; k < 10; k++) {
      // lo[k] holds cnt_k bit patterns 0x433.. | low52; hi[k] holds patterns 0x467.. | high52: strip the exponent fields
      const int cnt_lo = (k < 5) ? k + 1 : (k < 9 ? 9 - k : 0), cnt_hi = (k >= 1) ? ((k - 1 < 5) ? k : (k - 1 < 9 ? 10 - k : 0)) : 0;
      const u64 l = o2[(size_t)t * 20 + k] - (u64)cnt_lo * 0x4330000000000000ull, h = o2[(size_t)t * 20 + 10 + k] - (u64)cnt_hi * 0x4670000000000000ull;
      (void)M52;
      add_at(y, l, 52 * k); add_at(y, h, 52 * k);
    }
    if (memcmp(x.data(), y.data(), 16 * 4)) bad++;
  }
  printf("a*b of 4096 random 256-bit pairs, integer vs FP64 limb products: %s\n", bad ? "MISMATCH" : "identical");
  // ---- timing
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  float ms_int = 0, ms_fp = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0, 0); hipLaunchKernelGGL(k_int, dim3(blocks), dim3(threads), 0, 0, din, dout, iters); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms_int, e0, e1);
    hipEventRecord(e0, 0); hipLaunchKernelGGL(k_fp, dim3(blocks), dim3(threads), 0, 0, din52, dout2, iters); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms_fp, e0, e1);
  }
  const double prods = (double)n * iters;
  printf("integer 9x29 a*b columns : %8.3f ms  -> %.3f ns per 256x256-bit product per CU-equivalent lane batch, %.2f G products/s\n", ms_int, ms_int * 1e6 / prods, prods / ms_int / 1e6);
  printf("FP64    5x52 a*b columns : %8.3f ms  -> %.2f G products/s\n", ms_fp, prods / ms_fp / 1e6);
  printf("RESULT fp64_fma_product: %s (FP64 path is %.2fx the time of the integer path for the same 256x256-bit product columns)\n", ms_fp < ms_int ? "PASS" : "FAIL", ms_fp / ms_int);
  return bad != 0;
}
