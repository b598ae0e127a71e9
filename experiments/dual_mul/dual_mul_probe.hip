#include "jj_field.h"
using namespace jj;
// two independent Montgomery products, FIPS order, carries pinned as first addends, statements interleaved A/B
template <class P>
__device__ __forceinline__ void mul2(Fe& ra, Fe& rb, const Fe& a1, const Fe& b1, const Fe& a2, const Fe& b2) {
  u32 m1[NL], m2[NL];
  u64 acc1 = 0, acc2 = 0;
  #pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) {
    #pragma unroll
    for (int i = 0; i < NL; i++) {
      const int j = k - i;
      if (j < 0 || j >= NL) continue;
      acc1 = mad_vv(a1.l[i], b1.l[j], acc1);
      acc2 = mad_vv(a2.l[i], b2.l[j], acc2);
    }
    #pragma unroll
    for (int i = 0; i < NL; i++) {
      const int j = k - i;
      if (i >= k || j < 1 || j >= NL) continue;
      acc1 = mad_vs(m1[i], P::P[j], acc1);
      acc2 = mad_vs(m2[i], P::P[j], acc2);
    }
    if (k < NL) {
      m1[k] = (0u - (u32)acc1) & LMASK; m2[k] = (0u - (u32)acc2) & LMASK;
      acc1 = mad_vs(m1[k], 1u, acc1); acc2 = mad_vs(m2[k], 1u, acc2);
    } else {
      ra.l[k - NL] = (u32)acc1 & LMASK; rb.l[k - NL] = (u32)acc2 & LMASK;
    }
    acc1 >>= LB; acc2 >>= LB;
  }
  ra.l[NL - 1] = (u32)acc1; rb.l[NL - 1] = (u32)acc2;
}
extern "C" __global__ void kM2(u32* out, const u32* a, const u32* b) {
  int tid = blockIdx.x*blockDim.x+threadIdx.x;
  Fe x, y, z, w;
  for (int i=0;i<9;i++){x.l[i]=a[tid*18+i]; y.l[i]=b[tid*18+i]; z.l[i]=a[tid*18+9+i]; w.l[i]=b[tid*18+9+i];}
  Fe r1, r2;
  mul2<FqP>(r1, r2, x, y, z, w);
  for (int i=0;i<9;i++) { out[tid*18+i]=r1.l[i]; out[tid*18+9+i]=r2.l[i]; }
}
