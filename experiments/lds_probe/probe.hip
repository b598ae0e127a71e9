// lds_probe — what one CU's LDS pipe sustains for the two instructions of the fixed-base comb's constant-time select
// (k_fixedbase_comb<true>: per mixed addition 54 ds_bpermute_b32 + 14 ds_read_b128), alone and beside a stream of multiply-adds.
//   hipcc --offload-arch=gfx950 -O3 -o experiments/lds_probe/probe experiments/lds_probe/probe.hip && ./experiments/lds_probe/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>   // 0: bpermute only; 1: ds_read_b128 only; 2: mads only; 3: 54 bpermute + 14 reads per 1400 mads (the comb's mix); 4: the mix without LDS
__global__ void __launch_bounds__(768) k(uint32_t* out, int iters) {
  __shared__ uint4 lds[2048];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = make_uint4(i, i + 1, i + 2, i + 3);
  __syncthreads();
  uint32_t v[8]; for (int q = 0; q < 8; q++) v[q] = threadIdx.x * 7 + q;
  int addr = ((lane * 5 + 3) & 63) << 2;
  long long acc[4] = {1, 2, 3, 4};
  int a = threadIdx.x | 1, b = (threadIdx.x * 3) | 1;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0 || MODE == 3) {
      const int nb = MODE == 0 ? 64 : 54;
      #pragma unroll
      for (int q = 0; q < nb; q++) v[q & 7] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v[(q + 1) & 7]);
    }
    if (MODE == 1 || MODE == 3) {
      const int nr = MODE == 1 ? 64 : 14;
      #pragma unroll
      for (int q = 0; q < nr; q++) { const uint4 t = lds[(threadIdx.x + q * 67 + (v[0] & 1)) & 2047]; v[q & 7] ^= t.x + t.y + t.z + t.w; }
    }
    if (MODE >= 2) {
      const int nm = MODE == 2 ? 1024 : 1400;
      #pragma unroll 8
      for (int q = 0; q < nm; q++) { acc[q & 3] += (long long)a * (long long)b; a += (int)(acc[q & 3] >> 40); }
    }
  }
  uint32_t r = 0; for (int q = 0; q < 8; q++) r ^= v[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r ^ (uint32_t)(acc[0] + acc[1] + acc[2] + acc[3]) ^ a;
}
template <int MODE> static int run(const char* name, int threads, int iters, double per_iter_units, const char* unit) {
  int dev = 0; hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
  const int blocks = p.multiProcessorCount;
  uint32_t* out; CK(hipMalloc(&out, (size_t)blocks * threads * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 10);
  CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters); CK(hipGetLastError()); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double clk = 2.2e9 * ms * 1e-3;   // nominal cycles
  const int waves = threads / 64;
  printf("%-46s threads/CU %4d: %8.3f ms  -> %7.2f cycles (at 2.2 GHz) per %s per CU (%d waves), %.2f per wave-instruction per SIMD-slot\n", name, threads, ms,
         clk / (iters * per_iter_units * waves), unit, waves, clk / (iters * per_iter_units * waves / 4.0));
  CK(hipFree(out)); return 0;
}
int main() {
  for (int t : {256, 512, 768}) {
    run<0>("ds_bpermute_b32 only", t, 2000, 64, "bpermute");
    run<1>("ds_read_b128 only", t, 2000, 64, "read");
    run<2>("v_mad_i64_i32 chain x4 only", t, 200, 1024, "mad");
    run<3>("comb mix: 54 bperm + 14 reads + 1400 mads", t, 200, 1, "mix iteration");
    run<4>("the same 1400 mads without LDS", t, 200, 1, "mix iteration");
  }
  return 0;
}
