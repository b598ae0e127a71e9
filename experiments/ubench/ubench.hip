// Instruction-rate microbenchmark for gfx950 (MI355X): measures the wave64 issue cost of the integer
// instructions the Jubjub field arithmetic is built from, so the integer-VALU roofline denominator
// (peak v_mad_u64_u32 / s) is MEASURED, not assumed.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <string>
typedef uint32_t u32; typedef uint64_t u64;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

#define NCH 8      // independent chains per lane
#define UNROLL 8   // repetitions of the NCH-instruction group per loop iteration

// Each kernel: NCH independent accumulators, loop `iters` times over UNROLL*NCH instructions.
#define KERNEL32(name, ASM)                                                                       \
  __global__ void __launch_bounds__(256) name(u32* out, int iters, u32 seed) {                     \
    u32 acc[NCH]; u32 a = seed * 2654435761u + threadIdx.x, b = seed ^ (blockIdx.x * 40503u) | 1u;\
    _Pragma("unroll") for (int k = 0; k < NCH; k++) acc[k] = a + k * 77u;                           \
    for (int it = 0; it < iters; it++) {                                                           \
      _Pragma("unroll") for (int r = 0; r < UNROLL; r++) {                                         \
        _Pragma("unroll") for (int k = 0; k < NCH; k++) asm volatile(ASM : "+v"(acc[k]) : "v"(a), "v"(b)); \
      }                                                                                            \
    }                                                                                              \
    u32 s = 0; _Pragma("unroll") for (int k = 0; k < NCH; k++) s ^= acc[k];                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                \
  }
#define KERNEL64(name, ASM)                                                                       \
  __global__ void __launch_bounds__(256) name(u32* out, int iters, u32 seed) {                     \
    u64 acc[NCH]; u32 a = seed * 2654435761u + threadIdx.x, b = seed ^ (blockIdx.x * 40503u) | 1u;\
    u64 w = ((u64)a << 32) | b;                                                                    \
    _Pragma("unroll") for (int k = 0; k < NCH; k++) acc[k] = w + k * 77u;                           \
    for (int it = 0; it < iters; it++) {                                                           \
      _Pragma("unroll") for (int r = 0; r < UNROLL; r++) {                                         \
        _Pragma("unroll") for (int k = 0; k < NCH; k++) asm volatile(ASM : "+v"(acc[k]) : "v"(a), "v"(b), "v"(w)); \
      }                                                                                            \
    }                                                                                              \
    u64 s = 0; _Pragma("unroll") for (int k = 0; k < NCH; k++) s ^= acc[k];                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);                          \
  }

KERNEL32(k_add_u32,      "v_add_u32 %0, %0, %1")
KERNEL32(k_and_b32,      "v_and_b32 %0, %0, %1")
KERNEL32(k_sub_u32,      "v_sub_u32 %0, %0, %1")
KERNEL32(k_add3_u32,     "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_mul_lo_u32,   "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi_u32,   "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mad_u32_u24,  "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_mul_u32_u24,  "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL32(k_cndmask,      "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_alignbit,     "v_alignbit_b32 %0, %0, %1, 29")
KERNEL32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(k_fma_f32,      "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_bpermute,     "ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)")
KERNEL32(k_addco_chain,  "v_add_co_u32 %0, vcc, %0, %1\n\ts_nop 1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc")
KERNEL32(k_addco_nonop,  "v_add_co_u32 %0, vcc, %0, %1")
KERNEL64(k_mad_u64_u32,  "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(k_mad_i64_i32,  "v_mad_i64_i32 %0, vcc, %1, %2, %0")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %3")
KERNEL64(k_lshrrev_b64,  "v_lshrrev_b64 %0, 29, %0")
KERNEL64(k_lshlrev_b64,  "v_lshlrev_b64 %0, 3, %0")
KERNEL64(k_fma_f64,      "v_fma_f64 %0, %0, %3, %3")
KERNEL64(k_pk_fma_f32,   "v_pk_fma_f32 %0, %0, %3, %3")
KERNEL64(k_mul_f64,      "v_mul_f64 %0, %0, %3")
KERNEL64(k_add_f64,      "v_add_f64 %0, %0, %3")

// single dependent chain latency probes (NCH=1 equivalent): one accumulator, UNROLL*NCH deps
__global__ void __launch_bounds__(256) k_mad_u64_dep(u32* out, int iters, u32 seed) {
  u64 acc = seed; u32 a = seed * 2654435761u + threadIdx.x, b = seed | 1u;
  for (int it = 0; it < iters; it++) {
    _Pragma("unroll") for (int r = 0; r < UNROLL * NCH; r++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)acc ^ (u32)(acc >> 32);
}
__global__ void __launch_bounds__(256) k_add_u32_dep(u32* out, int iters, u32 seed) {
  u32 acc = seed; u32 a = seed * 2654435761u + threadIdx.x;
  for (int it = 0; it < iters; it++) {
    _Pragma("unroll") for (int r = 0; r < UNROLL * NCH; r++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(a));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// mixed: one mad followed by 1 / 2 / 4 simple ops on other registers: do simple ops hide behind mads?
#define KERNELMIX(name, NS)                                                                        \
  __global__ void __launch_bounds__(256) name(u32* out, int iters, u32 seed) {                     \
    u64 acc[NCH]; u32 x[NCH]; u32 a = seed * 2654435761u + threadIdx.x, b = seed | 1u;             \
    _Pragma("unroll") for (int k = 0; k < NCH; k++) { acc[k] = a + k; x[k] = b + k; }               \
    for (int it = 0; it < iters; it++) {                                                           \
      _Pragma("unroll") for (int r = 0; r < UNROLL; r++) {                                         \
        _Pragma("unroll") for (int k = 0; k < NCH; k++) {                                           \
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));        \
          _Pragma("unroll") for (int s = 0; s < NS; s++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k]) : "v"(a)); \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
    u64 s = 0; _Pragma("unroll") for (int k = 0; k < NCH; k++) s ^= acc[k] + x[k];                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);                          \
  }
KERNELMIX(k_mix_mad_1add, 1)
KERNELMIX(k_mix_mad_2add, 2)
KERNELMIX(k_mix_mad_4add, 4)

struct Bench { const char* name; void (*fn)(u32*, int, u32); int instr_per_group; };

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device: %s  CUs=%d  clockRate=%d kHz  arch=%s\n", prop.name, cus, prop.clockRate, prop.gcnArchName);
  u32* out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(u32) * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<Bench> benches = {
    {"v_add_u32", k_add_u32, 1}, {"v_and_b32", k_and_b32, 1}, {"v_sub_u32", k_sub_u32, 1}, {"v_add3_u32", k_add3_u32, 1},
    {"v_lshl_add_u32", k_lshl_add_u32, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_cndmask_b32", k_cndmask, 1},
    {"v_add_co_u32(vcc)", k_addco_nonop, 1}, {"add_co+nop1+addc", k_addco_chain, 2},
    {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
    {"v_mul_u32_u24", k_mul_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1}, {"v_mad_u32_u24", k_mad_u32_u24, 1},
    {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_mad_i64_i32", k_mad_i64_i32, 1},
    {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_lshrrev_b64", k_lshrrev_b64, 1}, {"v_lshlrev_b64", k_lshlrev_b64, 1},
    {"v_fma_f32", k_fma_f32, 1}, {"v_pk_fma_f32", k_pk_fma_f32, 1}, {"v_fma_f64", k_fma_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_add_f64", k_add_f64, 1},
    {"ds_bpermute_b32+wait", k_bpermute, 1},
    {"v_mad_u64_u32 (1 dep chain)", k_mad_u64_dep, 1}, {"v_add_u32 (1 dep chain)", k_add_u32_dep, 1},
    {"mad + 1 add", k_mix_mad_1add, 2}, {"mad + 2 add", k_mix_mad_2add, 3}, {"mad + 4 add", k_mix_mad_4add, 5},
  };
  const int iters = 2000;
  printf("%-30s %6s %12s %14s %16s\n", "instruction", "w/SIMD", "ms", "Ginstr/s(lane)", "clk/wave-instr@2.4GHz/SIMD");
  for (auto& b : benches) {
    for (int wps : {1, 2, 4, 8}) {
      int blocks = cus * wps;  // 256-thread blocks: 4 waves = 1 wave per SIMD per block
      b.fn<<<blocks, 256>>>(out, 10, 1); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      b.fn<<<blocks, 256>>>(out, iters, 12345);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double winstr = (double)iters * UNROLL * NCH * b.instr_per_group;        // wave-instructions per wave
      double lane_instr = winstr * 64.0 * blocks * 4;                           // lane-ops total
      double gips = lane_instr / (ms * 1e-3) / 1e9;
      double clk_per = (ms * 1e-3 * 2.4e9) / (winstr * wps);                   // cycles per wave-instr per SIMD at 2.4 GHz
      printf("%-30s %6d %12.4f %14.1f %16.2f\n", b.name, wps, ms, gips, clk_per);
    }
  }
  return 0;
}
