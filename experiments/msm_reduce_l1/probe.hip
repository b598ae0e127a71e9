// Costing the next step of DESIGN §8 item 2 (not part of the library): a FIRST LEVEL of the MSM's bucket reduce in lane form.
// Lane m of window slot s owns the strided buckets m, m + M, ..., m + (R - 1) M (R = B / M) and leaves
//   S_m = sum_i b_{m + i M},   T_m = sum_i i b_{m + i M}            (running sums: 2 (R - 2) + 1 whole-lane additions)
// so that the window's sum_j (j + 1) b_j = sum_m (m + 1) S_m + M sum_m T_m: the existing quad kernel then reduces the S_m (an R x smaller
// problem) and a plain sum takes the T_m.  Compile-only probe: instruction count and registers of the kernel give its time at one wave per
// SIMD (instructions x 4 cycles / clock); it was never run.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I jubjub_amd/csrc -Rpass-analysis=kernel-resource-usage -c -o /tmp/l1.o experiments/msm_reduce_l1/probe.hip
//   /opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading ... (count.sh next to this file)
#include <hip/hip_runtime.h>
#include "jj_kernels.h"
using namespace jj;

extern "C" __global__ void __launch_bounds__(256) k_reduce_l1(u32 B, u32 M, u32 Ws, ExtAoS buckets, ExtAoS S, ExtAoS T) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (size_t)Ws * M) return;
  const u32 s = (u32)(t / M), m = (u32)(t % M), R = B / M;
  const size_t base = (size_t)s * B + m;
  Ext running = aos_ext(buckets, base + (size_t)(R - 1) * M);
  Ext total = running;                                       // T = sum_{i >= 1} (b_i + b_{i+1} + ... + b_{R-1})
  Ext nxt = aos_ext(buckets, base + (size_t)(R - 2) * M);
  #pragma unroll 1
  for (int i = (int)R - 2; i >= 1; i--) {
    const Ext cur = nxt;
    nxt = aos_ext(buckets, base + (size_t)(i - 1) * M);      // the next bucket is in flight while this one is added
    running = Curve::add(running, Curve::to_niels(cur));
    total = Curve::add(total, Curve::to_niels(running));
  }
  running = Curve::add(running, Curve::to_niels(nxt));       // + b_0: S complete
  aos_put_ext(S, t, running);
  aos_put_ext(T, t, total);
}
