// energy_probe — is the integer pipeline issue-bound or power-bound?  Every kernel runs N multiply-adds per lane on every SIMD (8 waves per
// CU, 8 independent chains per lane); the variants add K "cheap" instructions per multiply-add.  If the part were issue-bound, the time
// would grow by K x 100 %; at the socket power limit it grows by the ENERGY those instructions cost relative to a 32x32+64 multiply-add.
//   hipcc --offload-arch=gfx950 -O3 -o experiments/lds_probe/energy_probe experiments/lds_probe/energy_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 6: + one v_mov_b32 per mad; 7: + one v_bfi_b32 per mad
// MODE 0: mads only; 1: + one v_add_u32 per mad; 2: + one 64-bit add (v_lshl_add_u64) per mad; 3: + one v_and_b32 + one v_ashrrev_i64 per mad (the
// Montgomery column step); 4: + two v_add_u32 per mad; 5: mads replaced by v_mul_u32_u24-class 24-bit multiply-adds (v_mad_u32_u24)
template <int MODE>
__global__ void __launch_bounds__(512) k(uint32_t* out, int iters, uint32_t seed) {
  long long acc[8];
  uint32_t x[8], y[8];
  for (int q = 0; q < 8; q++) { acc[q] = q + threadIdx.x; x[q] = seed * (q + 3) + threadIdx.x; y[q] = seed * (q + 11) + blockIdx.x; }
  for (int it = 0; it < iters; it++) {
    #pragma unroll
    for (int r = 0; r < 8; r++) {
      #pragma unroll
      for (int q = 0; q < 8; q++) {
        if (MODE == 5) { asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x[q]) : "v"(y[q]), "v"(y[(q + 1) & 7])); }
        else { asm volatile("" : "+v"(x[q])); acc[q] += (long long)(int)x[q] * (long long)(int)y[q]; }   // (the empty asm keeps the product from being hoisted)
        if (MODE == 1 || MODE == 4) x[q] += y[(q + 1) & 7];
        if (MODE == 4) y[q] += x[(q + 3) & 7];
        if (MODE == 2) acc[(q + 1) & 7] += acc[q];
        if (MODE == 3) { x[q] = (uint32_t)acc[q] & 0x1fffffffu; acc[q] >>= 29; }
        if (MODE == 6) { uint32_t t; asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(y[q])); y[(q + 1) & 7] = t; }
        if (MODE == 7) { x[q] = (x[q] & y[(q + 1) & 7]) | (y[q] & ~y[(q + 1) & 7]); }
      }
    }
  }
  uint32_t r = 0; for (int q = 0; q < 8; q++) r ^= (uint32_t)acc[q] ^ (uint32_t)(acc[q] >> 32) ^ x[q] ^ y[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> static int run(const char* name, double* base) {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount, threads = 512, iters = 20000;
  uint32_t* out; CK(hipMalloc(&out, (size_t)blocks * threads * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u);
  float best = 1e9;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 12345u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  const double mads = (double)iters * 64 * threads * blocks;
  if (*base == 0) *base = best;
  printf("%-64s %8.3f ms  %6.2f T mad/s  time vs mads only: %.3f\n", name, best, mads / (best * 1e-3) / 1e12, best / *base);
  CK(hipFree(out)); return 0;
}
int main() {
  double base = 0;
  run<0>("multiply-adds only (v_mad_i64_i32)", &base);
  run<1>("+ 1 v_add_u32 per multiply-add", &base);
  run<4>("+ 2 v_add_u32 per multiply-add", &base);
  run<2>("+ 1 v_lshl_add_u64 per multiply-add", &base);
  run<3>("+ v_and_b32 + v_ashrrev_i64 per multiply-add", &base);
  run<6>("+ 1 v_mov_b32 per multiply-add", &base);
  run<7>("+ 1 v_bfi_b32 per multiply-add", &base);
  run<5>("24-bit multiply-adds instead (v_mad_u32_u24)", &base);
  run<0>("multiply-adds only again", &base);
  return 0;
}
