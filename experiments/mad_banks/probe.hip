// Round 6: why does a stream of v_mad_u64_u32 / v_mad_i64_i32 issue one wave-instruction per 4.35 cycles per SIMD and not per 4 (profiles/r6_peak_clock.txt)?
// Streams of 64 multiply-adds per loop trip with EXPLICIT register numbers: does the rate depend on which VGPR banks the four source dwords
// (src0, src1, src2 lo/hi) and the two destination dwords live in, on a scalar operand, on dst != src2, on signedness?
//   hipcc --offload-arch=gfx950 -O2 -o probe probe.hip && ./probe            (8 waves per SIMD, then 1 and 2 waves per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(x) x x x x x x x x
// one trip = 8 x 8 mads over eight accumulator pairs; registers v[40..71] are ours (clobbered), a in %1, b in %2 are moved to fixed registers first
#define BODY(OP, A, B, P0, P1, P2, P3, P4, P5, P6, P7) \
  REP8(OP " v[" P0 "], vcc, " A ", " B ", v[" P0 "]\n" OP " v[" P1 "], vcc, " A ", " B ", v[" P1 "]\n" OP " v[" P2 "], vcc, " A ", " B ", v[" P2 "]\n" OP " v[" P3 "], vcc, " A ", " B ", v[" P3 "]\n" \
       OP " v[" P4 "], vcc, " A ", " B ", v[" P4 "]\n" OP " v[" P5 "], vcc, " A ", " B ", v[" P5 "]\n" OP " v[" P6 "], vcc, " A ", " B ", v[" P6 "]\n" OP " v[" P7 "], vcc, " A ", " B ", v[" P7 "]\n")
#define CLOB "vcc", "s40", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79"
#define KERNEL(NAME, SETUP, ...) \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, int iters, uint32_t seed) { \
    uint32_t a = seed * 2654435761u + threadIdx.x, b = (seed ^ (blockIdx.x * 40503u)) | 1u, r; \
    asm volatile(SETUP : : "v"(a), "v"(b) : CLOB); \
    for (int it = 0; it < iters; it++) asm volatile(BODY(__VA_ARGS__) : : : CLOB); \
    asm volatile("v_xor_b32 %0, v48, v57" : "=v"(r) : : CLOB); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r; }
// a -> v42 (bank 2), b -> v43 (bank 3); second copies a -> v44 (bank 0), b -> v40 (bank 0); scalar copy of b's first lane -> s40
#define SET "v_mov_b32 v42, %0\nv_mov_b32 v43, %1\nv_mov_b32 v44, %0\nv_mov_b32 v40, %1\nv_mov_b32 v41, %1\nv_readfirstlane_b32 s40, %1\n"
// accumulator pairs: even-aligned pairs sit in banks (0,1) or (2,3)
KERNEL(k_base,      SET, "v_mad_u64_u32", "v42", "v43", "48:49", "50:51", "52:53", "54:55", "56:57", "58:59", "60:61", "62:63")   // a bank 2, b bank 3, accumulators alternate (0,1)/(2,3)
KERNEL(k_signed,    SET, "v_mad_i64_i32", "v42", "v43", "48:49", "50:51", "52:53", "54:55", "56:57", "58:59", "60:61", "62:63")
KERNEL(k_ab_same,   SET, "v_mad_u64_u32", "v44", "v40", "48:49", "50:51", "52:53", "54:55", "56:57", "58:59", "60:61", "62:63")   // a and b both in bank 0
KERNEL(k_acc01,     SET, "v_mad_u64_u32", "v42", "v43", "48:49", "52:53", "56:57", "60:61", "64:65", "68:69", "72:73", "76:77")   // accumulators all in banks (0,1), a/b in (2,3): no source shares a bank
KERNEL(k_acc23,     SET, "v_mad_u64_u32", "v42", "v43", "50:51", "54:55", "58:59", "62:63", "66:67", "70:71", "74:75", "78:79")   // accumulators all in banks (2,3) = the banks of a and b
KERNEL(k_acc01_ab01, SET, "v_mad_u64_u32", "v40", "v41", "48:49", "52:53", "56:57", "60:61", "64:65", "68:69", "72:73", "76:77")  // everything in banks (0,1)
KERNEL(k_scalar_b,  SET, "v_mad_u64_u32", "v42", "s40", "48:49", "50:51", "52:53", "54:55", "56:57", "58:59", "60:61", "62:63")   // b from a scalar register
KERNEL(k_aa,        SET, "v_mad_u64_u32", "v42", "v42", "48:49", "50:51", "52:53", "54:55", "56:57", "58:59", "60:61", "62:63")   // a * a (one register read twice: a square's diagonal)
template <class K> static void run(const char* name, K kern, int blocks, uint32_t* out, int iters, const char* note) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mads = 64.0 * iters * 256.0 * blocks;
  printf("%-14s %5d workgroups  %8.3f ms  %6.2f T mads/s   %s\n", name, blocks, best, mads / (best * 1e-3) / 1e12, note);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  uint32_t* out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
  for (int wps : {8, 1, 2, 3, 4}) {
    const int blocks = cus * wps, iters = 4096 * 8 / wps;
    printf("# %d wave(s) per SIMD\n", wps);
    run("base", k_base, blocks, out, iters, "a bank 2, b bank 3, accumulator pairs alternate (0,1)/(2,3)");
    run("signed", k_signed, blocks, out, iters, "v_mad_i64_i32, same registers");
    run("ab_same_bank", k_ab_same, blocks, out, iters, "a and b in bank 0");
    run("acc01", k_acc01, blocks, out, iters, "accumulators all in banks (0,1), a / b in (2,3)");
    run("acc23", k_acc23, blocks, out, iters, "accumulators all in banks (2,3) with a and b");
    run("all01", k_acc01_ab01, blocks, out, iters, "accumulators, a and b all in banks (0,1)");
    run("scalar_b", k_scalar_b, blocks, out, iters, "b in a scalar register");
    run("a_times_a", k_aa, blocks, out, iters, "src0 = src1");
  }
  return 0;
}
