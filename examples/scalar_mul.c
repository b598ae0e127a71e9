/* Minimal C caller of the drop-in boundary (include/jubjub_hip.h):
 *   cc -Iinclude examples/scalar_mul.c -Ljubjub_amd/lib -ljubjub_hip -Wl,-rpath,$PWD/jubjub_amd/lib -o scalar_mul
 * Computes [k](8G) for k = 1..16 with the var-base ladder, the constant-time ladder and a fixed-base table, compresses the
 * results and prints them: they are the 16 encodings of the reference's test_serialization_consistency (src/lib.rs:1811-1876).
 * Then sum_k k * (8G) three ways: jj_msm, jj_msm_begin / jj_msm_finish, and jj_msm_partial (two window parts) + jj_msm_combine;
 * all must equal [136](8G) from the fixed-base table. */
#include <stdio.h>
#include <string.h>

#include "jubjub_hip.h"

int main(void) {
  jj_ctx* ctx = NULL;
  int rc = jj_ctx_create(0, &ctx);
  if (rc != JJ_OK) { fprintf(stderr, "jj_ctx_create: %d (no gfx950 GPU? there is no CPU fallback)\n", rc); return 1; }

  /* generator (src/lib.rs:1380-1396), affine u || v, little-endian */
  static const uint8_t GEN_U[32] = {0xfe, 0xad, 0xa7, 0xf1, 0x5d, 0xd3, 0xb3, 0xe4, 0xaf, 0x81, 0xbf, 0x29, 0x1b, 0x5d, 0xf5, 0xca,
                                    0x87, 0x81, 0x0a, 0xd6, 0xdd, 0x03, 0x0f, 0x8b, 0xc8, 0x87, 0x37, 0xbf, 0xb8, 0xcb, 0xed, 0x62};
  uint8_t g[64] = {0}, g8[64];
  memcpy(g, GEN_U, 32);
  g[32] = 11;
  if ((rc = jj_point_mul_by_cofactor(ctx, 1, g, g8))) goto fail;            /* 8G: generator of the prime-order subgroup */

  enum { N = 16 };
  uint8_t scalars[N][32] = {{0}}, points[N][64], out[N][64], enc[N][32], enc2[N][32];
  for (int i = 0; i < N; i++) { scalars[i][0] = (uint8_t)(i + 1); memcpy(points[i], g8, 64); }
  if ((rc = jj_varbase_mul(ctx, N, scalars, points, out))) goto fail;      /* ExtendedPoint * Fr */
  if ((rc = jj_compress(ctx, N, out, enc))) goto fail;                     /* AffinePoint::to_bytes */

  jj_table* table = NULL;
  if ((rc = jj_fixedbase_table_create(ctx, g8, 0, &table))) goto fail;     /* AffineNielsPoint * Fr with a device-resident table */
  if ((rc = jj_fixedbase_mul_compressed(ctx, table, N, scalars, enc2))) goto fail;
  uint8_t out_ct[N][64], enc3[N][32];
  if ((rc = jj_varbase_mul_ct(ctx, N, scalars, points, out_ct))) goto fail; /* the same products, no scalar-dependent address or branch */
  if ((rc = jj_compress(ctx, N, out_ct, enc3))) goto fail;
  /* sum_k k * (8G) = [1 + 2 + ... + 16](8G) = [136](8G) */
  uint8_t k136[32] = {136}, want[64], sum1[64], sum2[64], sum3[64];
  static uint8_t recs[2][JJ_MSM_PARTIAL_BYTES];
  jj_msm_job* job = NULL;
  if ((rc = jj_fixedbase_mul(ctx, table, 1, k136, want))) goto fail;
  if ((rc = jj_msm(ctx, N, scalars, points, sum1))) goto fail;             /* iter.map(|(p, k)| p * k).sum() */
  if ((rc = jj_msm_begin(ctx, N, scalars, points, &job))) goto fail;       /* the same in two halves */
  if ((rc = jj_msm_finish(job, sum2))) goto fail;
  for (int g = 0; g < 2; g++) if ((rc = jj_msm_partial(ctx, N, scalars, points, g, 2, recs[g]))) goto fail;   /* ... and in two window parts */
  if ((rc = jj_msm_combine(2, recs, sum3))) goto fail;
  jj_fixedbase_table_destroy(ctx, table);

  for (int i = 0; i < N; i++) {
    printf("%2d*(8G) = ", i + 1);
    for (int b = 0; b < 32; b++) printf("%02x", enc[i][b]);
    printf("%s\n", (memcmp(enc[i], enc2[i], 32) || memcmp(enc[i], enc3[i], 32)) ? "  MISMATCH between the ladders and the fixed-base table" : "");
  }
  printf("sum_k k*(8G): msm %s, begin/finish %s, partial+combine %s\n", memcmp(sum1, want, 64) ? "MISMATCH" : "ok", memcmp(sum2, want, 64) ? "MISMATCH" : "ok",
         memcmp(sum3, want, 64) ? "MISMATCH" : "ok");
  jj_ctx_destroy(ctx);
  return 0;
fail:
  fprintf(stderr, "libjubjub_hip error %d: %s\n", rc, jj_last_error(ctx));
  jj_ctx_destroy(ctx);
  return 1;
}
