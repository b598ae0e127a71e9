// ubench2: issue behaviour of v_mad_u64_u32 chains on gfx950 — dependent vs independent chains, occupancy, s_nop cost,
// select instructions, LDS gather and ds_bpermute throughput.  Cycle counts come from s_memtime (shader clock).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32; typedef uint64_t u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int NCH>
__global__ void __launch_bounds__(256) k_mad_chains(u32* out, u64* cyc, int iters, u32 seed) {
  u64 acc[NCH]; u32 a = seed * 2654435761u + threadIdx.x, b = seed | 1u;
  #pragma unroll
  for (int k = 0; k < NCH; k++) acc[k] = a + k;
  u64 t0 = clock64();
  for (int it = 0; it < iters; it++) {
    #pragma unroll
    for (int r = 0; r < 48 / NCH; r++) {
      #pragma unroll
      for (int k = 0; k < NCH; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "vcc");
    }
  }
  u64 t1 = clock64();
  u64 s = 0;
  #pragma unroll
  for (int k = 0; k < NCH; k++) s ^= acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// one dependent mad chain with K independent VOP2 adds between consecutive mads
template <int K>
__global__ void __launch_bounds__(256) k_mad_dep_fill(u32* out, u64* cyc, int iters, u32 seed) {
  u64 acc = seed; u32 x[4]; u32 a = seed * 2654435761u + threadIdx.x, b = seed | 1u;
  #pragma unroll
  for (int k = 0; k < 4; k++) x[k] = a + k;
  u64 t0 = clock64();
  for (int it = 0; it < iters; it++) {
    #pragma unroll
    for (int r = 0; r < 48; r++) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
      #pragma unroll
      for (int k = 0; k < K; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k & 3]) : "v"(a));
    }
  }
  u64 t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)acc ^ (u32)(acc >> 32) ^ x[0] ^ x[1] ^ x[2] ^ x[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// dependent chain with s_nop 0 between
__global__ void __launch_bounds__(256) k_mad_dep_nop(u32* out, u64* cyc, int iters, u32 seed) {
  u64 acc = seed; u32 a = seed * 2654435761u + threadIdx.x, b = seed | 1u;
  u64 t0 = clock64();
  for (int it = 0; it < iters; it++) {
    #pragma unroll
    for (int r = 0; r < 48; r++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\ts_nop 0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
  }
  u64 t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)acc ^ (u32)(acc >> 32);
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// select instructions: 8 independent chains
#define KSEL(name, ASM)                                                                     \
  __global__ void __launch_bounds__(256) name(u32* out, u64* cyc, int iters, u32 seed) {    \
    u32 acc[8]; u32 a = seed * 2654435761u + threadIdx.x, b = seed | 1u; u64 msk = 0x5555aaaa5555aaaaull; \
    _Pragma("unroll") for (int k = 0; k < 8; k++) acc[k] = a + k;                           \
    u64 t0 = clock64();                                                                     \
    for (int it = 0; it < iters; it++) {                                                    \
      _Pragma("unroll") for (int r = 0; r < 6; r++) {                                       \
        _Pragma("unroll") for (int k = 0; k < 8; k++) asm volatile(ASM : "+v"(acc[k]) : "v"(a), "v"(b), "s"(msk)); \
      }                                                                                     \
    }                                                                                       \
    u64 t1 = clock64();                                                                     \
    u32 s = 0; _Pragma("unroll") for (int k = 0; k < 8; k++) s ^= acc[k];                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                         \
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                              \
  }
KSEL(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, %3")
KSEL(k_cndmask_vcc_e32, "v_cndmask_b32_e32 %0, %0, %1, vcc")
KSEL(k_cndmask_vcc_e64, "v_cndmask_b32_e64 %0, %0, %1, vcc")
KSEL(k_cndmask_vcc_e32_setvcc, "v_cmp_gt_u32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc")
KSEL(k_bfi, "v_bfi_b32 %0, %2, %1, %0")
KSEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KSEL(k_xor, "v_xor_b32 %0, %0, %1")
KSEL(k_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
KSEL(k_mov, "v_mov_b32 %0, %1")

// LDS: per-lane random 16-byte reads (ds_read_b128) from a 128 KiB table, 8 in flight
__global__ void __launch_bounds__(256) k_lds_gather(u32* out, u64* cyc, int iters, u32 seed) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = i * seed;
  __syncthreads();
  u32 idx = (threadIdx.x * 2654435761u) ^ seed; uint4 s = make_uint4(0, 0, 0, 0);
  u64 t0 = clock64();
  for (int it = 0; it < iters; it++) {
    #pragma unroll
    for (int r = 0; r < 8; r++) {
      idx = idx * 1664525u + 1013904223u;
      const uint4 v = *reinterpret_cast<const uint4*>(&lds[((idx >> 8) & 8191) * 4]);
      s.x ^= v.x; s.y ^= v.y; s.z ^= v.z; s.w ^= v.w;
    }
  }
  u64 t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x ^ s.y ^ s.z ^ s.w;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// ds_bpermute: 8 issued back-to-back, one wait
__global__ void __launch_bounds__(256) k_bperm8(u32* out, u64* cyc, int iters, u32 seed) {
  u32 v[8]; u32 addr = ((threadIdx.x * 7 + seed) & 63) * 4;
  #pragma unroll
  for (int k = 0; k < 8; k++) v[k] = threadIdx.x + k;
  u64 t0 = clock64();
  for (int it = 0; it < iters; it++) {
    #pragma unroll
    for (int k = 0; k < 8; k++) v[k] = __builtin_amdgcn_ds_bpermute(addr, v[k]);
  }
  u64 t1 = clock64();
  u32 s = 0;
  #pragma unroll
  for (int k = 0; k < 8; k++) s ^= v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

typedef void (*kern_t)(u32*, u64*, int, u32);
static void run(const char* name, kern_t fn, int instr_per_iter, int cus, u32* out, u64* cyc, size_t lds = 0, int maxw = 8) {
  const int ws[] = {1, 2, 3, 4, 6, 8};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-34s", name);
  for (int wps : ws) {
    if (wps > maxw) { printf(" %7s/%-6s", "-", "-"); continue; }
    const int iters = 1000, blocks = cus * wps;
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), lds, 0, out, cyc, 5, 1u); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), lds, 0, out, cyc, iters, 12345u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    u64 c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    double per_simd_instr = (double)iters * instr_per_iter * wps;
    printf(" %7.2f/%-6.2f", (double)c / per_simd_instr, ms * 1e-3 * 2.4e9 / per_simd_instr);
  }
  printf("\n");
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  u32* out; u64* cyc; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4)); CK(hipMalloc(&cyc, 64));
  CK(hipFuncSetAttribute((const void*)k_lds_gather, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  printf("cycles per wave-instruction per SIMD: <s_memtime-based>/<wall-clock @2.4GHz>; columns = waves/SIMD 1,2,3,4,6,8\n");
  run("mad 1 chain", k_mad_chains<1>, 48, cus, out, cyc);
  run("mad 2 chains", k_mad_chains<2>, 48, cus, out, cyc);
  run("mad 3 chains", k_mad_chains<3>, 48, cus, out, cyc);
  run("mad 4 chains", k_mad_chains<4>, 48, cus, out, cyc);
  run("mad 8 chains", k_mad_chains<8>, 48, cus, out, cyc);
  run("mad dep + 1 add (per 2 instr)", k_mad_dep_fill<1>, 96, cus, out, cyc);
  run("mad dep + 2 add (per 3 instr)", k_mad_dep_fill<2>, 144, cus, out, cyc);
  run("mad dep + s_nop 0 (per mad)", k_mad_dep_nop, 48, cus, out, cyc);
  run("v_cndmask_b32_e64 sgpr mask", k_cndmask_sgpr, 48, cus, out, cyc);
  run("v_cndmask_b32_e32 vcc", k_cndmask_vcc_e32, 48, cus, out, cyc);
  run("v_cndmask_b32_e64 vcc", k_cndmask_vcc_e64, 48, cus, out, cyc);
  run("v_cmp+v_cndmask_e32 (per 2)", k_cndmask_vcc_e32_setvcc, 96, cus, out, cyc);
  run("v_bfi_b32", k_bfi, 48, cus, out, cyc);
  run("v_and_or_b32", k_and_or, 48, cus, out, cyc);
  run("v_xor_b32", k_xor, 48, cus, out, cyc);
  run("v_lshrrev_b32", k_lshrrev_b32, 48, cus, out, cyc);
  run("v_mov_b32", k_mov, 48, cus, out, cyc);
  run("ds_read_b128 random (128KiB)", k_lds_gather, 8, cus, out, cyc, 131072, 1);
  run("ds_bpermute_b32 x8 batched", k_bperm8, 8, cus, out, cyc);
  return 0;
}
