// The integer-VALU roofline denominator, taken apart (round 6): the library's k_peak_mad (8 independent v_mad_u64_u32 chains per
// lane, nothing else) beside a stream with the constant-time ladder's instruction mix -- 153 multiply-adds : 34 other VALU instructions
// (k_varbase_ct3's inner loop, profiles/r5_varbase_pmc.txt) = 72 : 16 per iteration here -- and a stream of 32-bit adds only.
//   hipcc --offload-arch=gfx950 -O3 -o experiments/peak_clock/probe experiments/peak_clock/probe.hip
// Run under rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES (tools/peak_clock.sh): effective clock =
// GRBM_GUI_ACTIVE / kernel duration; cycles per wave-instruction per SIMD = GRBM_GUI_ACTIVE * 1024 / SQ_INSTS_VALU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32; typedef uint64_t u64;

__global__ void __launch_bounds__(256) k_pure_mad(u32* out, int iters, u32 seed) {
  u64 acc[8];
  const u32 a = seed * 2654435761u + threadIdx.x, b = (seed ^ (blockIdx.x * 40503u)) | 1u;
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = (((u64)a << 32) | b) + k * 77u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int k = 0; k < 8; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
    }
  }
  u64 s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) s ^= acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}
// 72 multiply-adds + 16 other VALU instructions per iteration (4.5 : 1, the ladder's ratio): after every 9 mads an add and a shift/and pair member
__global__ void __launch_bounds__(256) k_mix_mad(u32* out, int iters, u32 seed) {
  u64 acc[8];
  u32 x = seed + threadIdx.x, y = seed ^ 0x9e3779b9u;
  const u32 a = seed * 2654435761u + threadIdx.x, b = (seed ^ (blockIdx.x * 40503u)) | 1u;
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = (((u64)a << 32) | b) + k * 77u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int k = 0; k < 8; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[r]) : "v"(a), "v"(b) : "vcc");      // the ninth
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
      asm volatile("v_and_b32 %0, 0x1fffffff, %0\n" : "+v"(y));
    }
  }
  u64 s = x ^ y;
#pragma unroll
  for (int k = 0; k < 8; k++) s ^= acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}
__global__ void __launch_bounds__(256) k_pure_add(u32* out, int iters, u32 seed) {
  u32 acc[8];
  const u32 a = seed * 2654435761u + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = a + k * 77u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int k = 0; k < 8; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[k]) : "v"(a));
    }
  }
  u32 s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) s ^= acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class K> static void run(const char* name, K kern, int cus, u32* out, int iters, double valu_per_iter, double mads_per_iter) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = cus * 8;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double lanes = 256.0 * blocks;
    printf("%-11s rep %d: %.3f ms  %.2f T mads/s  %.2f T VALU lane-instructions/s  (= %.3f GHz x 1024 SIMDs x 16 lanes if every cycle issues)\n", name, rep, ms,
           mads_per_iter * iters * lanes / (ms * 1e-3) / 1e12, valu_per_iter * iters * lanes / (ms * 1e-3) / 1e12, valu_per_iter * iters * lanes / (ms * 1e-3) / (1024.0 * 16) / 1e9);
  }
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  u32* out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
  printf("# %s, %d CUs, clockRate %d kHz\n", p.gcnArchName, cus, p.clockRate);
  for (int round = 0; round < 2; round++) {
    run("pure_mad", k_pure_mad, cus, out, 4000, 64, 64);
    run("mix_mad", k_mix_mad, cus, out, 4000, 88, 72);
    run("pure_add", k_pure_add, cus, out, 16000, 64, 0);
  }
  hipFree(out);
  return 0;
}
