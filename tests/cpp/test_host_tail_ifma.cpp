// Unit test of the AVX-512 IFMA host tail (jubjub_amd/csrc/jj_host_tail_ifma.h) against the scalar host tail of the same file set
// (jj_host_tail.h, which the C oracle pins through jj_msm_combine in tests/test_dist_cpu.py): the four-lane Montgomery product on random
// and zero operands (limbs normalised, value equal to the scalar product), point doubling / addition incl. the identity, P + P and chains of
// 40 doublings with additions (the ranges must close), the Horner chain for several window layouts with missing windows and identity sums,
// and the timing of both chains.  CPU only; needs avx512ifma + avx512vl (tests/test_abi.py skips it otherwise).
//   g++ -O2 -std=c++17 -mavx512f -mavx512vl -mavx512ifma -o test_host_tail_ifma test_host_tail_ifma.cpp && ./test_host_tail_ifma
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <random>
#include "../../jubjub_amd/csrc/jj_host_tail.h"
using namespace jjhost;
static std::mt19937_64 rng(12345);
static Fe rnd_fe() { Fe a; for (int i = 0; i < 4; i++) a.l[i] = rng(); a.l[3] &= 0x3fffffffffffffffull; while (geq_q(a.l)) sub_q(a.l); return a; }
// random curve point via scalar arithmetic: take generator-ish: solve? simpler: random multiples of a fixed point built from identity additions
static Ext rnd_point(const Ext& base) {
  Ext acc = identity();
  uint64_t k = rng();
  Ext b = base;
  for (int i = 0; i < 64; i++) { if ((k >> i) & 1) acc = point_add(acc, b); b = point_dbl(b); }
  return acc;
}
static Ext SP(const ifma::V4& v) { ifma::P4 o; ifma::st(o, v); return ifma::unpack_point(o); }
static bool same_point(const Ext& a, const Ext& b) {   // U1 Z2 == U2 Z1 and V1 Z2 == V2 Z1
  const Fe x1 = mul(a.u, b.z), x2 = mul(b.u, a.z), y1 = mul(a.v, b.z), y2 = mul(b.v, a.z);
  return !memcmp(&x1, &x2, 32) && !memcmp(&y1, &y2, 32);
}
int main() {
  printf("ifma available: %d\n", (int)ifma::available());
  // field mul check: mont260(a,b) * 2^260 == a*b  <=>  compare through scalar: mont256(a,b) = ab/2^256; mont260 = ab/2^260 -> mont260 * 16 == mont256
  for (int it = 0; it < 100000; it++) {
    Fe a[4], b[4];
    ifma::V4 va, vb;
    uint64_t la[4][5], lb[4][5];
    for (int i = 0; i < 4; i++) { a[i] = rnd_fe(); b[i] = rnd_fe(); ifma::split52(a[i].l, la[i]); ifma::split52(b[i].l, lb[i]); }
    for (int j = 0; j < 5; j++) { va.l[j] = _mm256_set_epi64x(la[3][j], la[2][j], la[1][j], la[0][j]); vb.l[j] = _mm256_set_epi64x(lb[3][j], lb[2][j], lb[1][j], lb[0][j]); }
    ifma::V4 r = ifma::mul(va, vb);
    alignas(32) uint64_t lane[5][4];
    for (int j = 0; j < 5; j++) _mm256_store_si256((__m256i*)lane[j], r.l[j]);
    for (int i = 0; i < 4; i++) {
      uint64_t l[5] = {lane[0][i], lane[1][i], lane[2][i], lane[3][i], lane[4][i]};
      for (int j = 0; j < 5; j++) if (l[j] > ifma::M52) { printf("limb not normalised it=%d lane=%d j=%d %llx\n", it, i, j, (unsigned long long)l[j]); return 1; }
      Fe c; ifma::join52(l, c.l); while (geq_q(c.l)) sub_q(c.l);
      for (int k = 0; k < 4; k++) c = dbl(c);          // * 16
      const Fe want = mul(a[i], b[i]);
      if (memcmp(&c, &want, 32)) { printf("mul mismatch it=%d lane=%d\n", it, i); return 1; }
    }
  }
  printf("field mul ok\n");
  // a base point: from a valid record-like point: use scalar code: find point by (u,v) of the generator in Montgomery form
  // generator (reference src/lib.rs full generator): u = 0x62edcbb8bf3787c88b0f03ddd60a8187caf55d1b29bf81afe4b3d35df1a7adfe, v = 0x0b
  const uint8_t gu[32] = {0xfe,0xad,0xa7,0xf1,0x5d,0xd3,0xb3,0xe4,0xaf,0x81,0xbf,0x29,0x1b,0x5d,0xf5,0xca,0x87,0x81,0x0a,0xd6,0xdd,0x03,0x0f,0x8b,0xc8,0x87,0x37,0xbf,0xb8,0xcb,0xed,0x62};
  uint8_t gv[32] = {0x0b};
  Fe U = from_canon(gu), V = from_canon(gv);
  Ext G{U, V, consts().one, U, V};
  // sanity: G on curve? -u^2 + v^2 = 1 + d u^2 v^2
  {
    Fe uu = sqr(U), vv = sqr(V); Fe lhs = sub(vv, uu); Fe d = consts().d2; // 2d
    Fe rhs2 = add(dbl(consts().one), mul(d, mul(uu, vv)));   // 2 + 2d u^2 v^2
    Fe lhs2 = dbl(lhs);
    printf("generator on curve: %d\n", !memcmp(&lhs2, &rhs2, 32));
  }
  for (int it = 0; it < 2000; it++) {
    Ext p = rnd_point(G), q = rnd_point(G);
    if (it % 7 == 0) q = p;               // doubling through add
    if (it % 11 == 0) q = identity();
    if (it % 13 == 0) p = identity();
    ifma::P4 pp, pq; ifma::pack_point(pp, p.u, p.v, p.z, mul(p.t1, p.t2)); ifma::pack_point(pq, q.u, q.v, q.z, mul(q.t1, q.t2));
    ifma::V4 vp = ifma::ld(pp), vq = ifma::ld(pq);
    Ext d1 = SP(ifma::point_dbl<false, false>(vp)), d0 = point_dbl(p);
    if (!same_point(d1, d0)) { printf("dbl mismatch %d\n", it); return 1; }
    Ext a1 = SP(ifma::point_add(vp, vq)), a0 = point_add(p, q);
    if (!same_point(a1, a0)) { printf("add mismatch %d\n", it); return 1; }
    // T consistency: t1*t2 * z == u * v
    Fe T = mul(a1.t1, a1.t2); Fe lhs = mul(T, a1.z), rhs = mul(a1.u, a1.v);
    if (memcmp(&lhs, &rhs, 32)) { printf("T inconsistent after add %d\n", it); return 1; }
    // chains: repeated dbl/add keep ranges
    ifma::V4 acc = vp;
    Ext sacc = p;
    for (int k = 0; k < 40; k++) { acc = ifma::point_dbl<false, false>(acc); sacc = point_dbl(sacc); if (k % 5 == 4) { acc = ifma::point_add(acc, vq); sacc = point_add(sacc, q); } }
    if (!same_point(SP(acc), sacc)) { printf("chain mismatch %d\n", it); return 1; }
  }
  printf("point ops ok\n");
  // Horner
  for (int W : {16, 17, 23, 64, 1, 2}) {
    bool have[64]; Ext sum[64];
    for (int rep = 0; rep < 50; rep++) {
      for (int w = 0; w < W; w++) { have[w] = (rng() % 5) != 0; sum[w] = rnd_point(G); if (rng() % 9 == 0) sum[w] = identity(); }
      WindowSums ws; ws.W = W; for (int w = 0; w < W; w++) { ws.have[w] = have[w]; ws.sum[w] = sum[w]; }
      ifma::P4 ps[64]; for (int w = 0; w < W; w++) ifma::pack_point(ps[w], sum[w].u, sum[w].v, sum[w].z, mul(sum[w].t1, sum[w].t2));
      Ext s0 = ws.finish_scalar(), s1 = ifma::horner(W, have, ps);
      if (!same_point(s0, s1)) { printf("horner mismatch W=%d rep=%d\n", W, rep); return 1; }
      uint8_t o0[64], o1[64]; to_affine64(o0, s0); to_affine64(o1, s1);
      if (memcmp(o0, o1, 64)) { printf("affine mismatch W=%d\n", W); return 1; }
    }
  }
  printf("horner ok\n");
  for (int trial = 0; trial < 3; trial++) {
    int W = 16; bool have[64]; Ext sum[64];
    for (int w = 0; w < W; w++) { have[w] = true; sum[w] = rnd_point(G); }
    WindowSums ws; ws.W = W; for (int w = 0; w < W; w++) { ws.have[w] = have[w]; ws.sum[w] = sum[w]; }
    auto t0 = std::chrono::steady_clock::now();
    Ext s; for (int i = 0; i < 500; i++) s = ws.finish_scalar();
    auto t1 = std::chrono::steady_clock::now();
    ifma::P4 ps[64]; for (int w = 0; w < W; w++) ifma::pack_point(ps[w], sum[w].u, sum[w].v, sum[w].z, mul(sum[w].t1, sum[w].t2));
    Ext v; for (int i = 0; i < 500; i++) v = ifma::horner(W, have, ps);
    auto t2 = std::chrono::steady_clock::now();
    uint8_t o[64]; for (int i = 0; i < 500; i++) to_affine64(o, v);
    auto t3 = std::chrono::steady_clock::now();
    printf("W=16 Horner: scalar %.2f us, ifma %.2f us; to_affine %.2f us  (%d)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 500,
           std::chrono::duration<double, std::micro>(t2 - t1).count() / 500, std::chrono::duration<double, std::micro>(t3 - t2).count() / 500, (int)same_point(s, v));
  }
  printf("IFMA HOST TAIL OK\n");
  return 0;
}
