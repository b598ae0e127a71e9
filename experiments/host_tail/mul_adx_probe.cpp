// mul_adx_probe -- the host tail's 4 x 64-bit Montgomery product written with mulx + adcx / adox (BMI2 + ADX) against the compiler's
// code for the portable version (jj_host_tail.h).   g++ -O3 -std=c++17 -o mul_adx_probe mul_adx_probe.cpp && ./mul_adx_probe
// Build host (Xeon 2.1 GHz): dependent chain of products 34.8 against 34.3 ns (the same latency), point doubling 207 against 165 ns
// (-20 %); built into the library (runtime dispatch, not inlined across the target attribute) the whole 23-window tail measured
// 67.0 against 64.5 us, the 17-window one 53.5 against 55.4: inside the noise -> NOT shipped.
#include <stdio.h>
#include <time.h>
#include "../../jubjub_amd/csrc/jj_host_tail.h"
using namespace jjhost;

// a*b/2^256 mod q with mulx + the two carry chains of adcx / adox (BMI2 + ADX): one row of a*b_i and one reduction row per word
__attribute__((target("bmi2,adx"), noinline)) static Fe mul_adx(const Fe& a, const Fe& b) {
  constexpr uint64_t NINV = 0xfffffffeffffffffull;
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
  const uint64_t zero = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t lo, hi;
    __asm__(
        "xorl %%eax, %%eax\n\t"
        "mulx 0(%[a]), %[lo], %[hi]\n\t"  "adcx %[lo], %[t0]\n\t" "adox %[hi], %[t1]\n\t"
        "mulx 8(%[a]), %[lo], %[hi]\n\t"  "adcx %[lo], %[t1]\n\t" "adox %[hi], %[t2]\n\t"
        "mulx 16(%[a]), %[lo], %[hi]\n\t" "adcx %[lo], %[t2]\n\t" "adox %[hi], %[t3]\n\t"
        "mulx 24(%[a]), %[lo], %[hi]\n\t" "adcx %[lo], %[t3]\n\t" "adox %[hi], %[t4]\n\t"
        "adcx %[z], %[t4]\n\t"
        : [t0] "+&r"(t0), [t1] "+&r"(t1), [t2] "+&r"(t2), [t3] "+&r"(t3), [t4] "+&r"(t4), [lo] "=&r"(lo), [hi] "=&r"(hi)
        : [a] "r"(a.l), "d"(b.l[i]), [z] "r"(zero), "m"(a)
        : "rax", "cc");
    const uint64_t m = t0 * NINV;
    __asm__(
        "xorl %%eax, %%eax\n\t"
        "mulx 0(%[q]), %[lo], %[hi]\n\t"  "adcx %[lo], %[t0]\n\t" "adox %[hi], %[t1]\n\t"
        "mulx 8(%[q]), %[lo], %[hi]\n\t"  "adcx %[lo], %[t1]\n\t" "adox %[hi], %[t2]\n\t"
        "mulx 16(%[q]), %[lo], %[hi]\n\t" "adcx %[lo], %[t2]\n\t" "adox %[hi], %[t3]\n\t"
        "mulx 24(%[q]), %[lo], %[hi]\n\t" "adcx %[lo], %[t3]\n\t" "adox %[hi], %[t4]\n\t"
        "adcx %[z], %[t4]\n\t"
        : [t0] "+&r"(t0), [t1] "+&r"(t1), [t2] "+&r"(t2), [t3] "+&r"(t3), [t4] "+&r"(t4), [lo] "=&r"(lo), [hi] "=&r"(hi)
        : [q] "r"(QL), "d"(m), [z] "r"(zero), "m"(QL)
        : "rax", "cc");
    t0 = t1; t1 = t2; t2 = t3; t3 = t4; t4 = 0;      // t0 became 0: divide by 2^64
  }
  // t < 2q: one conditional subtraction, branch-free
  uint64_t d0, d1, d2, d3, br;
  d0 = t0 - QL[0]; br = t0 < QL[0];
  uint64_t x = t1 - QL[1]; uint64_t b2 = (t1 < QL[1]) | (x < br); d1 = x - br; br = b2;
  x = t2 - QL[2]; b2 = (t2 < QL[2]) | (x < br); d2 = x - br; br = b2;
  x = t3 - QL[3]; b2 = (t3 < QL[3]) | (x < br); d3 = x - br; br = b2;
  const uint64_t keep = (uint64_t)0 - br;          // borrow: t < q
  Fe r = {{(t0 & keep) | (d0 & ~keep), (t1 & keep) | (d1 & ~keep), (t2 & keep) | (d2 & ~keep), (t3 & keep) | (d3 & ~keep)}};
  return r;
}
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static inline Ext dbl_adx(const Ext& p) {
  const Fe uu = mul_adx(p.u, p.u), vv = mul_adx(p.v, p.v), zz = mul_adx(p.z, p.z), s2 = add(p.u, p.v); const Fe uv2 = mul_adx(s2, s2);
  const Fe zz2 = dbl(zz), vpu = add(vv, uu), vmu = sub(vv, uu);
  const Fe cu = sub(uv2, vpu), ct = sub(zz2, vmu);
  return Ext{mul_adx(cu, ct), mul_adx(vpu, vmu), mul_adx(vmu, ct), cu, vpu};
}
int main() {
  {
    Ext p = identity(); p.u = consts().d2; p.v = consts().r2;
    Ext q = p;
    const int M = 2000000;
    double t0 = now(); for (int i = 0; i < M; i++) p = point_dbl(p); double t1 = now();
    for (int i = 0; i < M; i++) q = dbl_adx(q); double t2 = now();
    printf("point_dbl: plain %.1f ns, adx %.1f ns (%d)\n", (t1 - t0) / M * 1e9, (t2 - t1) / M * 1e9, memcmp(&p, &q, sizeof p));
  }
  Fe a = consts().d2, b = consts().r2;
  // correctness: a chain of products with both versions
  Fe x = a, y = a;
  for (int i = 0; i < 100000; i++) { x = mul(x, b); y = mul_adx(y, b); b = add(b, x); if (memcmp(&x, &y, 32)) { printf("MISMATCH at %d\n", i); return 1; } }
  const int N = 20000000;
  double t0 = now(); x = a; for (int i = 0; i < N; i++) x = mul(x, b); double t1 = now();
  y = a; for (int i = 0; i < N; i++) y = mul_adx(y, b); double t2 = now();
  printf("plain C (u128): %.2f ns per product, mulx/adcx/adox: %.2f ns per product  (%llx %llx)\n", (t1 - t0) / N * 1e9, (t2 - t1) / N * 1e9, (unsigned long long)x.l[0], (unsigned long long)y.l[0]);
  return memcmp(&x, &y, 32) != 0;
}
