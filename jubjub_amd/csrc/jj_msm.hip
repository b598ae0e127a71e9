// libjubjub_hip.so: multi-scalar multiplication (kernels: jj_msm_kernels.h; host tail: jj_host_tail.h).
#define JJ_KERNELS_MSM
#include "jj_engine.h"

// MSM on the device up to the record of partial window sums (jj_msm_kernels.h); everything here is launches only (the
// workspaces are grown first), so a pass can be queued behind another one without any host synchronisation in between.
//   part_w0 / part_stride: the pass owns windows part_w0, part_w0 + part_stride, ... (0 / 1: all of them)
//   rec_dev: MSM_REC bytes of device memory for the record
static_assert(MSM_SMALL_BLK_MAX == MSM_TREE_QUADS, "JJ_MSM_SMALL_BLK bound");
struct MsmGeometry { bool small; MsmParams mp; u32 nblk; };
static void msm_layout(MsmParams& mp, int W, int w0, int wstride) {
  mp.W = W; mp.c = 253 / W; mp.r = 253 % W;
  mp.w0 = w0; mp.wstride = wstride < 1 ? 1 : wstride;
  mp.Ws = w0 < W ? (W - w0 + mp.wstride - 1) / mp.wstride : 0;
  mp.B = 1u << (mp.c + (mp.r ? 1 : 0) - 1);
  memset(mp.recode, 0, sizeof mp.recode);
  int bit = 0;
  for (int w = 0; w < W - 1; w++) { const int width = mp.c + (w < mp.r ? 1 : 0); const int b = bit + width - 1; mp.recode[b >> 5] |= 1u << (b & 31); bit += width; }
}
// number of windows for n terms (measured on MI355X; JJ_MSM_WINDOWS overrides): the windows tile the 253 scalar bits exactly, so
// any W is as good as its entry count n W and its bucket count ~ W 2^(253/W - 1) make it
// from this many terms the large-input configuration (17 / 16 windows, length-sorted segments) is faster than 23 windows + chunks + fix-up
// (experiments/misc/msm_crossover.py: 131 072 terms 0.438 against 0.448 ms, 150 000 terms 0.485 against 0.459 ms, 235 000 terms 0.727 against 0.533 ms)
constexpr size_t MSM_LARGE_MIN = (size_t)9 << 14;
static int msm_windows_for(jj_ctx* c, size_t n) {
  if (c->msm_windows >= MSM_WINDOWS_MIN && c->msm_windows <= MSM_WINDOWS_MAX) return c->msm_windows;
  // measured (experiments/misc/msm_sweep*.sh, profiles/r3_msm_window_sweep.txt, r3_msm_reduce_grid.txt): 16 windows (13 of 16 bits, 3 of
  // 15) from 2^18 terms (round 6; from 2^20 before: with the two-level reduce -- 8 level-1 rows -- the 2^15 buckets per window no longer cost
  // a 210 us chain, and 16 windows beat 17 by 0.5 % at 2^18 and 2 % at 2^19 terms, profiles/r5_msm_mid_sweep.txt); 17 windows (15 of 15 bits,
  // 2 of 14: half the buckets for 6 % more additions) from MSM_LARGE_MIN terms; below, 23 windows of 11 bits: wider windows cut the additions
  // but their buckets (4096+ per window) make the fix-up and reduce chains longer than the additions they save
  return n >= ((size_t)1 << 18) ? 16 : n >= MSM_LARGE_MIN ? 17 : 23;
}
// counters (MSM_COUNTER_WORDS words, cleared by the first kernel of a pass) | big-bucket work list | workgroup partial sums
constexpr size_t MSM_BIG_OFF = 512, MSM_PART_OFF = MSM_BIG_OFF + sizeof(BigBucket) * FIXUP_BIG_MAX;
static int msm_ensure_ctl(jj_ctx* c, MsmLane& L) { return ensure(c, L.ctl, MSM_PART_OFF + (size_t)64 * MSM_TREE_QUADS * MSM_PART_WORDS * 4); }
static int msm_enqueue_small(jj_ctx* c, MsmLane& L, size_t n, const void* ds, const void* dp, int part_w0, int part_stride, void* rec_dev) {
  MsmParams mp;
  msm_layout(mp, SM_W, part_w0, part_stride);
  int rc;
  if ((rc = ensure(c, L.buf[0], n * 32))) return rc;
  if ((rc = ensure(c, L.buf[1], n * (size_t)(SM_SLOTS * ENIELS_WORDS) * 4))) return rc;
  if ((rc = msm_ensure_ctl(c, L))) return rc;
  u32* counters = (u32*)L.ctl.p; u32* part = (u32*)((uint8_t*)L.ctl.p + MSM_PART_OFF);
  // workgroups of 64 quads per window: about 4 terms per quad, at most msm_small_blk (4 x 64 windows = one workgroup per CU)
  const u32 nblk = (u32)std::min<size_t>(c->msm_small_blk, std::max<size_t>(1, (n + 255) / 256));
  hipLaunchKernelGGL(k_msm_small_tables, dim3(blocks_for(4 * n)), dim3(256), 0, L.stream, n, ds, dp, mp, (u32*)L.buf[1].p, (u32*)L.buf[0].p, counters);
  hipLaunchKernelGGL(k_msm_small_sum, dim3(nblk, mp.Ws), dim3(4 * MSM_TREE_QUADS), 0, L.stream, n, mp, nblk, (const u32*)L.buf[1].p, (const u32*)L.buf[0].p, part, counters, (u32*)rec_dev);
  return JJ_OK;
}

static int msm_enqueue_pippenger(jj_ctx* c, MsmLane& ln, size_t n, const void* ds, const void* dp, int part_w0, int part_stride, void* rec_dev) {
  MsmParams mp;
  msm_layout(mp, msm_windows_for(c, n), part_w0, part_stride);
  const u32 B = mp.B, Ws = (u32)mp.Ws;
  const size_t nb = (size_t)Ws * B;
  // buckets per reduce chunk: a quad walks L buckets (2 L additions), then ~2 c operations multiply by the chunk's first index, and
  // every workgroup of 64 quads is one wave per SIMD of a CU.  The chains are bound by the instructions a wave issues, and a
  // second workgroup on a CU slows both by ~1.6x, so: the smallest L (at least 4) for which the workgroups that have chunks fit
  // one per CU.  JJ_MSM_REDUCE_CHUNK overrides; never more than one window.
  auto reduce_blocks = [&](u32 l) { const u32 nk = std::min<u32>(MSM_TREE_QUADS, (B / l + MSM_TREE_QUADS - 1) / MSM_TREE_QUADS); u32 t = 0; for (u32 s = 0; s < Ws; s++) t += msm_reduce_blocks(mp, (int)s, l, nk); return t; };
  u32 L_auto = 4;
  while (L_auto < B && reduce_blocks(L_auto) > (u32)c->cus) L_auto <<= 1;
  const u32 L = std::min<u32>(c->msm_reduce_chunk ? (u32)c->msm_reduce_chunk : L_auto, B);
  if (B % L || (L & (L - 1))) { c->err = "inconsistent MSM tuning override (option msm_reduce_chunk must be a power of two dividing the bucket count)"; return JJ_ERR_INVALID; }
  const u32 K = B / L, nblk = std::min<u32>(MSM_TREE_QUADS, (K + MSM_TREE_QUADS - 1) / MSM_TREE_QUADS);      // workgroups of 64 quads per window, at most
  const u32 reduce_grid = reduce_blocks(L);
  int jbits = 0; while ((1u << jbits) < B) jbits++;
  // Two-level reduce (k_msm_reduce_l1 + k_msm_reduce_l2) for wide windows: level 1 sums R rows of the bucket matrix per lane (whole-lane
  // additions at full throughput), level 2 is the quad chain over M = B / R columns per window.  Narrow windows (2^17-term passes: 1024
  // buckets per window) keep the one-level kernel: level 1 would be an extra launch and ~20 us of chain for a level 2 that is as deep.
  u32 l1_rows = 0;
  if (c->msm_l1_rows > 0) l1_rows = (u32)c->msm_l1_rows;
  else if (c->msm_l1_rows < 0 && B >= 16384) l1_rows = B >= 32768 ? 8 : 4;
  const u32 Bmin = mp.r ? B / 2 : B;                                     // buckets of the narrowest window of the layout
  while (l1_rows > 1 && (l1_rows > Bmin || B / l1_rows < 64)) l1_rows >>= 1;     // every window has at least one row; whole waves per window
  if (l1_rows < 2) l1_rows = 0;
  int mbits = 0; u32 L2 = 0, nblk2 = 0;
  if (l1_rows) {
    const u32 M = B / l1_rows;
    while ((1u << mbits) < M) mbits++;
    L2 = 4;
    while (L2 < M && ((u64)Ws * ((M / L2 + MSM_TREE_QUADS - 1) / MSM_TREE_QUADS) > (u64)c->cus || (M / L2 + MSM_TREE_QUADS - 1) / MSM_TREE_QUADS > (u32)MSM_TREE_QUADS)) L2 <<= 1;
    if (c->msm_l2_chunk && (u32)c->msm_l2_chunk <= M && (M / (u32)c->msm_l2_chunk + MSM_TREE_QUADS - 1) / MSM_TREE_QUADS <= (u32)MSM_TREE_QUADS) L2 = (u32)c->msm_l2_chunk;
    nblk2 = (M / L2 + MSM_TREE_QUADS - 1) / MSM_TREE_QUADS;
  }
  const size_t l1_bytes = l1_rows ? (size_t)Ws * (B / l1_rows) * ENIELS_WORDS * 4 : 0;      // one array of extended-Niels records (S, then T)
  int rc;
  DevBuf &kprime = ln.buf[0], &niels = ln.buf[1], &offb = ln.buf[2], &idx = ln.buf[3], &buckets = ln.buf[4], &ra = ln.buf[5], &tcnt = ln.buf[7];
  // Entries per lane of the chunked accumulation (below MSM_LARGE_MIN terms; above, the segments decide).  The launch is Ws x ceil(nchunk / 256)
  // workgroups of one wave per SIMD, and the dispatcher deals them round the CUs: what counts is how many ROUNDS of workgroups a CU gets, so the
  // chunk is the shortest one that fits the launch into msm_chunk_waves rounds (round 6: 2^17 terms ran 16 entries in 2.9 rounds -- three -- where
  // 24 entries in 1.9 rounds cost the same accumulation time and leave a third fewer heads to the fix-up: 353 -> 339 us per call; 2^16 terms ran
  // 1.4 rounds, i.e. two, of 16 entries for 23 entries' worth of work).  Rounds 2-5: 16 entries up to 2^19 terms, 8 below 2^15.
  u32 chunk = 8;
  // Jobs in flight (lanes of their own) share the CUs with the neighbouring job's kernels, the rounds argument does not hold for them and
  // shorter chunks interleave better: one round more (measured, four jobs in flight at 2^17 terms: 0.228-0.241 ms per MSM against 0.249-0.258).
  { const size_t cap = (size_t)(std::max(1, c->msm_chunk_waves) + (&ln != &c->lanes[0] ? 1 : 0)) * (size_t)c->cus;
    while (chunk < 1024 && (size_t)Ws * ((((n + chunk - 1) / chunk) + 255) / 256) > cap) chunk++; }
  if (c->msm_chunk) chunk = (u32)c->msm_chunk;
  const u32 nchunk = (u32)((n + chunk - 1) / chunk);
  const bool two_pass = B > 4096 || (B == 4096 && c->msm_two_pass != 0);      // the one-pass plan kernel covers 4096 buckets per window
  const u32 HB = B >> MSM_LO_BITS;
  if (two_pass && HB > MSM_HB_MAX) { c->err = "MSM window layout has more coarse bins per window than the two-pass sort holds (option msm_windows must be 16..36)"; return JJ_ERR_INVALID; }
  const u32 ptiles = (u32)((n + MSM_P1_TILE - 1) / MSM_P1_TILE);        // the first pass of the two-pass sort orders a whole tile in LDS
  const size_t pm = (size_t)HB * ptiles;                               // runs per slot
  // one-pass sort tiles: enough (tile, window) blocks to fill the GPU, each at least 4096 terms
  const u32 ntiles = (u32)std::max<size_t>(1, std::min<size_t>((size_t)(c->cus + Ws - 1) / Ws, (n + 4095) / 4096));
  const size_t tile = (n + ntiles - 1) / ntiles;
  // one-pass sort in two launches (round 6): counting blocks (part, slot), as many parts as give every CU one block, each part at least 4096 terms
  const u32 f2_parts = (u32)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)c->msm_sort_blocks_per_cu * c->cus / Ws, MSM_F2_PARTS_MAX), (n + 4095) / 4096));
  const size_t f2_part_terms = (n + f2_parts - 1) / f2_parts;
  const bool front1 = !two_pass && c->msm_front1 && B <= 4096;
  const bool use_segments = c->msm_segments == 1 || (c->msm_segments < 0 && n >= MSM_LARGE_MIN);
  // segments of at most P entries, sorted by length; P bounds the serial depth of one lane: twice the mean bucket of the widest windows
  u32 P = (u32)std::min<size_t>(SEG_PMAX, std::max<size_t>(32, 2 * n / B));
  if (c->msm_seg_len >= 8 && c->msm_seg_len <= SEG_PMAX) P = (u32)c->msm_seg_len;
  const u32 stiles = (u32)std::max<size_t>(1, std::min<size_t>(4096, (nb + 255) / 256));          // tiles of the two segment passes: 256 buckets each, more above 2^20 buckets
  const u32 per_tile = (u32)((nb + stiles - 1) / stiles);
  // with the two-pass sort a tile of the segment passes is exactly the 256 buckets of one k_msm_part_sort workgroup, which then writes the tile's
  // histogram of segment lengths itself (no k_seg_hist launch)
  const bool seg_fused = use_segments && two_pass && per_tile == (1u << MSM_LO_BITS) && (size_t)stiles * per_tile == nb && stiles == Ws * HB;
  const size_t max_segs = nb + (n * (size_t)Ws) / P + 1;
  const size_t bh_words = (size_t)stiles * (P + 1), hdr_words = bh_words + 2 * (P + 2) + 16;
  if ((rc = ensure(c, kprime, n * 32))) return rc;
  if ((rc = ensure(c, niels, n * (size_t)GNIELS_WORDS * 4))) return rc;
  if ((rc = ensure(c, offb, (size_t)Ws * (B + 1) * 4))) return rc;
  if ((rc = ensure(c, idx, n * (size_t)Ws * 4))) return rc;
  if ((rc = ensure(c, tcnt, two_pass ? ((size_t)Ws * (2 * pm + 1)) * 4 : front1 ? (size_t)Ws * f2_parts * B * 4 : (size_t)Ws * ntiles * B * 4))) return rc;
  if ((rc = ensure(c, buckets, (size_t)EXT_AOS_WORDS * 4 * nb))) return rc;
  // ra: first the two-pass sort's records (4 + 1 bytes per entry), then the chunk heads / segment heads
  // (and, once the heads are folded in, the two arrays level 1 of the reduce hands to level 2)
  if ((rc = ensure(c, ra, std::max<size_t>(std::max<size_t>((size_t)EXT_AOS_WORDS * 4 * std::max<size_t>((size_t)Ws * nchunk, (n * (size_t)Ws) / 8 + 1), n * (size_t)Ws * 5 + 64), 2 * l1_bytes)))) return rc;
  if ((rc = msm_ensure_ctl(c, ln))) return rc;                                                           // counters, big-bucket work list, workgroup partial sums
  if ((rc = ensure(c, ln.bigpart, (size_t)5 * NL * 4 * FIXUP_BIG_MAX * FIXUP_BIG_QUADS))) return rc;      // the big buckets' partial sums
  if (use_segments && (rc = ensure(c, ln.seg, hdr_words * 4 + 16 + nb * sizeof(MergeItem) + max_segs * sizeof(Seg)))) return rc;   // bh [stiles][P+1] | count [P+1] | offset [P+2] | merge list | segments
  hipStream_t st = ln.stream;
  u32* off = (u32*)offb.p;
  u32* counters = (u32*)ln.ctl.p;                          // cleared by the sort's plan kernel
  BigBucket* big = (BigBucket*)((uint8_t*)ln.ctl.p + MSM_BIG_OFF);
  u32* part = (u32*)((uint8_t*)ln.ctl.p + MSM_PART_OFF);
  // One conversion launch for scalars and points.  (Rounds 2-3 ran the point half on a second stream beside the sort from 2^18
  // terms; with the entries staged through LDS the conversion is short enough that the fork, its two events and the contention
  // with the sort's first kernel cost more than the overlap returns: 2^18 terms 0.565 -> 0.556 ms, 2^20 1.262 -> 1.254 ms.)
  // Two-pass sort, round 5: the coarse histogram is taken by the conversion kernel itself (k_msm_convert_hist: totals per (slot, bin)), the tiles of
  // the first pass reserve their runs with one global atomic per bin: no k_msm_part_hist, no k_msm_part_plan.  JJ_MSM_SORT_HIST=separate keeps the
  // round-4 kernels (per-tile counts + a scan: the order of the entries inside a bucket is then the order of the terms).
  // (measured, experiments/misc/msm_sort_hist_ab.py, three boxes: 2^20 terms -3 ... -7 % (1.32 -> 1.24 ms), 2^19 -1 ... -4 %, 2^18 and 2^21 0 ... -2 %, 2^22 +0.7 ... -1.5 %:
  // at 2^22 terms the 8192 tiles' atomics on the same 2048 cursors cost about what the histogram pass they replace costs, so the fused form stops at 3 x 2^20 terms)
  const bool fused_hist = two_pass && c->msm_fused_hist && Ws <= 64 && Ws * HB <= 4096 && n <= ((size_t)3 << 20);
  if (fused_hist) {
    const bool fresh = ln.bins.cap == 0;
    if ((rc = ensure(c, ln.bins, (size_t)2 * 2 * MSM_BINS_WORDS * 4))) return rc;
    if (fresh) { HIPCHK(c, hipMemsetAsync(ln.bins.p, 0, ln.bins.cap, st)); ln.bins_parity = 0; }     // afterwards every pass clears the other parity's half
    // 160 KB: the staging of sixteen waves + up to 4096 counters.  Once per CONTEXT, with its device current: the attribute belongs to the
    // device's copy of the kernel (jj_multi_* drives several devices from one process)
    if (!c->msm_hist_lds_set) { HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_convert_hist), hipFuncAttributeMaxDynamicSharedMemorySize, MSM_CH_STAGE_WORDS * 4 + 16384)); c->msm_hist_lds_set = true; }
    u32* totals = (u32*)ln.bins.p + (size_t)ln.bins_parity * 2 * MSM_BINS_WORDS;
    u32* cursor = totals + MSM_BINS_WORDS;
    u32* other = (u32*)ln.bins.p + (size_t)(ln.bins_parity ^ 1) * 2 * MSM_BINS_WORDS;
    ln.bins_parity ^= 1;
    u32* rec = (u32*)ra.p; uint8_t* lo8 = (uint8_t*)ra.p + n * (size_t)Ws * 4;     // the head buffer is free until the accumulation
    // terms per thread: four (4096 per workgroup: the fewest counter flushes) when that still gives every CU a workgroup, fewer for smaller inputs
    const int per = (int)std::max<size_t>(1, std::min<size_t>(MSM_CH_PER, n / ((size_t)c->cus * MSM_CH_THREADS)));
    hipLaunchKernelGGL(k_msm_convert_hist, dim3((unsigned)((n + (size_t)per * MSM_CH_THREADS - 1) / ((size_t)per * MSM_CH_THREADS))), dim3(MSM_CH_THREADS), MSM_CH_STAGE_WORDS * 4 + Ws * HB * 4, st, n, ds, dp, mp, (u32*)kprime.p, (u32*)niels.p, totals, other, counters, per);
    hipLaunchKernelGGL(k_msm_part_scatter, dim3(ptiles, Ws), dim3(MSM_SORT_THREADS), 0, st, n, (size_t)MSM_P1_TILE, mp, (const u32*)kprime.p, (const u32*)nullptr, rec, lo8, (const u32*)totals, cursor);
    if (seg_fused) hipLaunchKernelGGL(k_msm_part_sort<true>, dim3(HB, Ws), dim3(MSM_P2_THREADS), 0, st, mp, (u32)n, (const u32*)nullptr, (const u32*)rec, (const uint8_t*)lo8, (u32*)idx.p, off, P, ExtAoS{(u32*)buckets.p}, (u32*)ln.seg.p, (const u32*)totals);
    else hipLaunchKernelGGL(k_msm_part_sort<false>, dim3(HB, Ws), dim3(MSM_P2_THREADS), 0, st, mp, (u32)n, (const u32*)nullptr, (const u32*)rec, (const uint8_t*)lo8, (u32*)idx.p, off, P, ExtAoS{nullptr}, (u32*)nullptr, (const u32*)totals);
  } else if (two_pass) {
    hipLaunchKernelGGL(k_msm_convert, dim3(blocks_for(n)), dim3(256), 0, st, n, ds, dp, mp, (u32*)kprime.p, (u32*)niels.p, 3);
    u32* tc = (u32*)tcnt.p; u32* tcs = tc + (size_t)Ws * pm;
    u32* rec = (u32*)ra.p; uint8_t* lo8 = (uint8_t*)ra.p + n * (size_t)Ws * 4;     // the head buffer is free until the accumulation
    hipLaunchKernelGGL(k_msm_part_hist, dim3(ptiles, Ws), dim3(MSM_SORT_THREADS), 0, st, n, (size_t)MSM_P1_TILE, mp, (const u32*)kprime.p, tc);
    hipLaunchKernelGGL(k_msm_part_plan, dim3(Ws), dim3(1024), 0, st, n, (u32)pm, (const u32*)tc, tcs, counters);
    hipLaunchKernelGGL(k_msm_part_scatter, dim3(ptiles, Ws), dim3(MSM_SORT_THREADS), 0, st, n, (size_t)MSM_P1_TILE, mp, (const u32*)kprime.p, (const u32*)tcs, rec, lo8, (const u32*)nullptr, (u32*)nullptr);
    if (seg_fused) hipLaunchKernelGGL(k_msm_part_sort<true>, dim3(HB, Ws), dim3(MSM_P2_THREADS), 0, st, mp, ptiles, (const u32*)tcs, (const u32*)rec, (const uint8_t*)lo8, (u32*)idx.p, off, P, ExtAoS{(u32*)buckets.p}, (u32*)ln.seg.p, (const u32*)nullptr);
    else hipLaunchKernelGGL(k_msm_part_sort<false>, dim3(HB, Ws), dim3(MSM_P2_THREADS), 0, st, mp, ptiles, (const u32*)tcs, (const u32*)rec, (const uint8_t*)lo8, (u32*)idx.p, off, P, ExtAoS{nullptr}, (u32*)nullptr, (const u32*)nullptr);
  } else if (front1) {
    // round 6: counting (from the raw scalars) + point conversion in one launch, plan + scatter in the next (k_msm_front2 / k_msm_scatter2); no k'
    const size_t f2_lds = std::max<size_t>((size_t)B * 4, (size_t)MSM_F2_STAGE_WORDS * 4);
    if (!c->msm_front1_lds_set) { HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_front2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>((size_t)4096 * 4, (size_t)MSM_F2_STAGE_WORDS * 4))); c->msm_front1_lds_set = true; }
    hipLaunchKernelGGL(k_msm_front2, dim3(f2_parts * Ws + (unsigned)((n + MSM_F2_THREADS - 1) / MSM_F2_THREADS)), dim3(MSM_F2_THREADS), f2_lds, st, n, ds, dp, mp, (u32*)niels.p, (u32*)tcnt.p, counters, f2_parts, f2_part_terms);
    hipLaunchKernelGGL(k_msm_scatter2, dim3(f2_parts * 8 * ((Ws + 7) / 8)), dim3(MSM_SORT_THREADS), B * 4, st, n, f2_parts, f2_part_terms, mp, ds, (const u32*)tcnt.p, off, (u32*)idx.p);
  } else {
    hipLaunchKernelGGL(k_msm_convert, dim3(blocks_for(n)), dim3(256), 0, st, n, ds, dp, mp, (u32*)kprime.p, (u32*)niels.p, 3);
    hipLaunchKernelGGL(k_msm_hist, dim3(ntiles, Ws), dim3(MSM_SORT_THREADS), B * 4, st, n, tile, mp, (const u32*)kprime.p, (u32*)tcnt.p);
    hipLaunchKernelGGL(k_msm_plan, dim3(Ws), dim3(1024), 0, st, n, B, ntiles, (u32*)tcnt.p, off, counters);
    hipLaunchKernelGGL(k_msm_scatter, dim3(ntiles * 8 * ((Ws + 7) / 8)), dim3(MSM_SORT_THREADS), B * 4, st, n, tile, ntiles, mp, (const u32*)kprime.p, (const u32*)tcnt.p, (u32*)idx.p);
  }
  const ExtAoS head{(u32*)ra.p}, bk{(u32*)buckets.p};
  SoA partial = soa_of(ln.bigpart, (size_t)FIXUP_BIG_MAX * FIXUP_BIG_QUADS);
  const MergeItem* merge_list = nullptr;
  if (use_segments) {
    u32* bh = (u32*)ln.seg.p; u32* soff = bh + bh_words + (P + 1);
    MergeItem* merge = (MergeItem*)(((uintptr_t)(bh + hdr_words) + 15) & ~(uintptr_t)15);
    Seg* seg = (Seg*)(merge + nb);
    if (!seg_fused) hipLaunchKernelGGL(k_seg_hist, dim3(stiles), dim3(256), 0, st, nb, B, per_tile, P, (const u32*)off, bk, bh);
    hipLaunchKernelGGL(k_seg_plan, dim3(P + 1), dim3(256), 0, st, stiles, bh, soff);
    hipLaunchKernelGGL(k_seg_scatter, dim3(stiles), dim3(256), 0, st, nb, B, per_tile, P, (const u32*)off, (const u32*)bh, (const u32*)soff, soff + (P + 1), seg, counters, merge, big);
    hipLaunchKernelGGL(k_msm_accumulate_seg, dim3(blocks_for(max_segs)), dim3(256), 0, st, (const u32*)(soff + (P + 1)), (const Seg*)seg, (const u32*)idx.p, (const u32*)niels.p, bk, head);
    merge_list = merge;
  } else {
    if (B <= MSM_ACC_LDS_BUCKETS && c->msm_acc_lds) hipLaunchKernelGGL(k_msm_accumulate<true>, dim3(blocks_for(nchunk), Ws), dim3(256), (B + 1) * 4, st, n, B, chunk, nchunk, (const u32*)off, (const u32*)idx.p, (const u32*)niels.p, bk, head);
    else hipLaunchKernelGGL(k_msm_accumulate<false>, dim3(blocks_for(nchunk), Ws), dim3(256), 0, st, n, B, chunk, nchunk, (const u32*)off, (const u32*)idx.p, (const u32*)niels.p, bk, head);
    hipLaunchKernelGGL(k_msm_fixup, dim3(blocks_for(2 * nb)), dim3(256), 0, st, n, B, Ws, chunk, nchunk, (const u32*)off, bk, head);      // (big buckets included: no second launch)
  }
  // (segment path: 512 workgroups, the merge list of repeated scalars is walked by the same launch as the big buckets)
  if (use_segments) hipLaunchKernelGGL(k_msm_fixup_big, dim3(2u * (unsigned)c->cus), dim3(256), 0, st, counters, (const BigBucket*)big, bk, head, partial, merge_list);
  if (l1_rows) {
    u32* SN = (u32*)ra.p; u32* TN = (u32*)((uint8_t*)ra.p + l1_bytes);        // the heads are dead: k_msm_fixup_big was their last reader
    hipLaunchKernelGGL(k_msm_reduce_l1, dim3(blocks_for((size_t)Ws << mbits)), dim3(256), 0, st, mp, mbits, bk, SN, TN);
    hipLaunchKernelGGL(k_msm_reduce_l2, dim3(Ws * nblk2), dim3(4 * MSM_TREE_QUADS), 0, st, n, mp, mbits, L2, nblk2, (const u32*)SN, (const u32*)TN, part, counters, (u32*)rec_dev);
  } else if (K > MSM_TREE_QUADS * nblk) hipLaunchKernelGGL(k_msm_reduce_fold<true>, dim3(reduce_grid), dim3(4 * MSM_TREE_QUADS), 0, st, n, mp, L, nblk, jbits, bk, part, counters, (u32*)rec_dev);
  else hipLaunchKernelGGL(k_msm_reduce_fold<false>, dim3(reduce_grid), dim3(4 * MSM_TREE_QUADS), 0, st, n, mp, L, nblk, jbits, bk, part, counters, (u32*)rec_dev);
  return JJ_OK;
}
// one pass (at most 2^24 terms: 32-bit sort indices), record left at rec_dev
static int msm_enqueue(jj_ctx* c, MsmLane& L, size_t n, const void* ds, const void* dp, int part_w0, int part_stride, void* rec_dev, size_t* rec_bytes) {
  const bool small = n <= (size_t)c->msm_small_max;
  *rec_bytes = jjhost::rec_bytes(small ? SM_W : msm_windows_for(c, n));
  return small ? msm_enqueue_small(c, L, n, ds, dp, part_w0, part_stride, rec_dev) : msm_enqueue_pippenger(c, L, n, ds, dp, part_w0, part_stride, rec_dev);
}
// lane k of the context, ready for use: lane 0 IS the context's launch stream (synchronous calls, host-staged inputs); the others own
// their stream, created on first use, and start a job after everything already queued on the launch stream (the inputs may have been
// produced there).  Jobs in flight (device-pointer inputs) alternate over lanes 1 .. msm_lanes and never run on lane 0: a lane that is the
// launch stream makes every other lane's next job wait for ITS job, and the jobs then run in pairs that start together (round 5,
// profiles/r5_msm_allgather_timeline.txt: -6 % with two or three jobs in flight once the lanes are independent)
static int msm_lane(jj_ctx* c, int k, MsmLane** out) {
  MsmLane& L = c->lanes[k];
  if (k == 0) { L.stream = c->stream; *out = &L; return JJ_OK; }
  if (!L.owned) {
    // launch stream + two copy streams + this lane: from the second lane on the context has more streams than HIP's default of four hardware
    // queues.  Streams that share a queue serialise (1.7-2x slower pipelines); the library does not touch the environment, it says so once.
    if (k >= 2) {
      static std::atomic<bool> warned{false};
      const char* q = getenv("GPU_MAX_HW_QUEUES");
      if ((!q || atoi(q) < 8) && !warned.exchange(true)) fprintf(stderr, "libjubjub_hip: MSM jobs in flight use 5+ streams; export GPU_MAX_HW_QUEUES=8 before the first HIP call (or set option msm_lanes=1), streams sharing a hardware queue serialise\n");
    }
    HIPCHK(c, hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&L.ready_ev, hipEventDisableTiming));
    L.owned = true;
  }
  HIPCHK(c, hipEventRecord(L.ready_ev, c->stream));
  HIPCHK(c, hipStreamWaitEvent(L.stream, L.ready_ev, 0));
  *out = &L;
  return JJ_OK;
}

// ---- asynchronous jobs: jj_msm_begin queues every pass of one MSM and the copy of its records into the job's own page-locked
// buffer, then returns; jj_msm_finish waits for that copy only and runs the host tail (window sums, Horner, one inversion),
// while the kernels of jobs begun meanwhile keep the device busy.  The context's workspaces are shared by all jobs: the stream
// orders them.
static int msm_job_get(jj_ctx* c, size_t nrec, jj_msm_job** out) {
  jj_msm_job* j = nullptr;
  if (!c->job_pool.empty()) { j = c->job_pool.back(); c->job_pool.pop_back(); }
  else {
    j = new jj_msm_job();
    j->c = c;
    if (hipEventCreateWithFlags(&j->ev, hipEventDisableTiming) != hipSuccess) { delete j; c->err = "hipEventCreate failed"; return JJ_ERR_HIP; }
  }
  const size_t want = std::max<size_t>(nrec, 1) * jjhost::REC_MAX_BYTES;
  if (j->cap < want) {
    if (j->host) (void)hipHostFree(j->host);
    j->host = nullptr; j->cap = 0;
    if (hipHostMalloc((void**)&j->host, want, hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); if (j->gdev) (void)hipFree(j->gdev); (void)hipEventDestroy(j->ev); delete j; c->err = "hipHostMalloc failed"; return JJ_ERR_NOMEM; }
    j->cap = want;
  }
  j->nrec = 0; j->gathered = 0; j->folded = false;
  for (size_t r = 0; r < std::max<size_t>(nrec, 1); r++) memset(j->host + r * jjhost::REC_MAX_BYTES, 0, jjhost::REC_HDR_BYTES);   // a stale header of a pooled buffer must never validate
  *out = j;
  return JJ_OK;
}
void msm_job_put(jj_ctx* c, jj_msm_job* j) {
  if (c->job_pool.size() < 8) { c->job_pool.push_back(j); return; }
  if (j->host) (void)hipHostFree(j->host);
  if (j->gdev) (void)hipFree(j->gdev);
  (void)hipEventDestroy(j->ev);
  delete j;
}
// spread: device-pointer jobs alternate over the context's lanes 1 .. msm_lanes (jj_msm_begin); otherwise lane 0 (jj_msm; host arrays are
// staged through buffers the launch stream owns; JJ_MSM_LANES=1: every job)
int msm_begin_locked(jj_ctx* c, size_t n, const void* scalars, const void* points, int part_w0, int part_stride, bool spread, jj_msm_job** out) {
  size_t PASS = (size_t)1 << c->msm_pass_log2;
  // Host arrays of 2^19 terms and more are reduced in SEVERAL passes (one record each, one host tail): the copy of a pass's slice runs on the
  // copy stream beside the kernels of the pass before it -- 96 bytes per term over the link cost more than the whole reduction (2^20 terms:
  // 1.7 ms of copy, 1.3 ms of kernels).  Two to eight passes of at least 2^19 terms (smaller passes reduce less efficiently than the copy
  // they hide): page-locked arrays 2^20 terms 3.15 -> 2.67 ms, 2^22 terms 12.3 -> 9.7 ms with two passes.  JJ_MSM_HOST_SPLIT=0: one pass
  // after the whole copy (round 3).
  const bool host_in = n && !is_device_ptr(scalars) && !is_device_ptr(points);
  const bool split = host_in && c->msm_host_split && n >= ((size_t)1 << 19);
  PASS = msm_host_pass_terms(n, c->msm_pass_log2, split);
  const size_t npass = n ? (n + PASS - 1) / PASS : 0;
  jj_msm_job* j;
  int rc = msm_job_get(c, npass, &j); if (rc) return rc;
  int k = 0;
  if (spread && n && c->msm_lanes > 1 && is_device_ptr(scalars) && is_device_ptr(points)) k = 1 + (int)(c->next_lane++ % (unsigned)c->msm_lanes);
  MsmLane* L = nullptr;
  if ((rc = msm_lane(c, k, &L))) { msm_job_put(c, j); return rc; }
  auto fail = [&](int code) { if (split && c->pipe.h2d) (void)hipStreamSynchronize(c->pipe.h2d); (void)hipStreamSynchronize(L->stream); (void)hipGetLastError(); msm_job_put(c, j); return code; };   // kernels may still be writing into the job's buffer
  if (n) {
    const void *ds, *dp;
    if (split) {
      // the passes' slices are copied on the copy stream (ordered after what the launch stream has queued: the staging buffers may still
      // be read by an earlier call's kernels), each pass's kernels wait for their slice only
      if ((rc = ensure(c, c->in[0], 32 * n)) || (rc = ensure(c, c->in[1], 64 * n)) || (rc = pipe_prepare(c, 0, 0))) return fail(rc);
      ds = c->in[0].p; dp = c->in[1].p;
      hipError_t e0 = hipEventRecord(c->order_ev, c->stream);
      if (e0 == hipSuccess) e0 = hipStreamWaitEvent(c->pipe.h2d, c->order_ev, 0);
      if (e0 != hipSuccess) { c->err = std::string("MSM staging failed: ") + hipGetErrorString(e0); return fail(JJ_ERR_HIP); }
    } else if ((rc = stage_in(c, 0, scalars, 32 * n, &ds)) || (rc = stage_in(c, 1, points, 64 * n, &dp))) return fail(rc);
    size_t stage_seq = 0;             // staging slots of the bounce path, counted over all arrays of all passes
    for (size_t lo = 0; lo < n; lo += PASS) {
      const size_t cnt = std::min(PASS, n - lo);
      size_t used = 0;
      if (split) {
        const struct { const void* host; void* dev; size_t elem; } arr[2] = {{scalars, c->in[0].p, 32}, {points, c->in[1].p, 64}};
        for (const auto& a : arr) {
          const uint8_t* src = (const uint8_t*)a.host + lo * a.elem; uint8_t* dst = (uint8_t*)a.dev + lo * a.elem;
          if (cnt * a.elem >= BOUNCE_THRESHOLD && !is_pinned_host(src, cnt * a.elem)) { if ((rc = host_to_dev_bounced(c, dst, src, cnt * a.elem, c->pipe.h2d, &stage_seq))) return fail(rc); }
          else if (hipMemcpyAsync(dst, src, cnt * a.elem, hipMemcpyHostToDevice, c->pipe.h2d) != hipSuccess) { c->err = "MSM staging copy failed"; return fail(JJ_ERR_HIP); }
        }
        const int ei = (int)((lo / PASS) & 1);
        if (hipEventRecord(c->pipe.ev_in[ei], c->pipe.h2d) != hipSuccess || hipStreamWaitEvent(L->stream, c->pipe.ev_in[ei], 0) != hipSuccess) { c->err = "MSM staging event failed"; return fail(JJ_ERR_HIP); }
      }
      // the kernels that finish a window write its point straight into the job's page-locked buffer (device-visible host memory):
      // no copy operation between the last kernel and the host tail
      if ((rc = msm_enqueue(c, *L, cnt, (const uint8_t*)ds + lo * 32, (const uint8_t*)dp + lo * 64, part_w0, part_stride, j->host + j->nrec * jjhost::REC_MAX_BYTES, &used))) return fail(rc);
      j->nrec++;
    }
    if (stage_seq && (rc = stage_in_drain(c, stage_seq))) return fail(rc);     // the staging slots are free for the next call
  }
  hipError_t e = hipEventRecord(j->ev, L->stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) { c->err = std::string("MSM launch failed: ") + hipGetErrorString(e); return fail(JJ_ERR_HIP); }
  *out = j;
  return JJ_OK;
}
JJ_API int jj_msm_begin(jj_ctx* c, size_t n, const void* scalars, const void* points, jj_msm_job** job) {
  if (!c || !job) return JJ_ERR_INVALID;
  *job = nullptr;
  JJ_ENTER(c);
  return msm_begin_locked(c, n, scalars, points, 0, 1, true, job);
}
// waits for the job's records, host tail, result to out64 (host pointer: written before the call returns; device pointer: a
// 64-byte copy queued on the context's stream).  The job is released in every case.
JJ_API int jj_msm_finish(jj_msm_job* j, void* out64) {
  if (!j || !j->c) return JJ_ERR_INVALID;
  jj_ctx* c = j->c;
  if (!out64) { (void)hipEventSynchronize(j->ev); std::lock_guard<std::recursive_mutex> lk(c->mu); msm_job_put(c, j); return JJ_ERR_INVALID; }   // the job's kernels may still be writing into its buffer
  hipError_t e = hipEventSynchronize(j->ev);                       // no context lock while waiting: other threads may queue work
  jjhost::Ext total = jjhost::identity();
  if (e == hipSuccess && j->gathered && j->folded) {
    // a job of jj_msm_allgather_begin: the fold kernel left ONE record -- or a zero header: records of different window layouts (ranks
    // with different term counts), which the host adds after one copy of all of them out of the job's own device buffer
    uint32_t magic; memcpy(&magic, j->host, 4);
    if (magic != MSM_REC_MAGIC) {
      // (the context's device current, the copy on the context's own stream and a wait for THAT stream only: a blocking hipMemcpy runs on the
      // null stream, which synchronises with every blocking stream of the process -- e.g. a caller's stream that carries the next job's gather)
      std::lock_guard<std::recursive_mutex> lk(c->mu);
      e = hipSetDevice(c->device);
      if (e == hipSuccess) e = hipMemcpyAsync(j->host, (const uint8_t*)j->gdev + JJ_MSM_PARTIAL_BYTES, (size_t)j->gathered * JJ_MSM_PARTIAL_BYTES, hipMemcpyDeviceToHost, c->own_stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->own_stream);
      j->nrec = (size_t)j->gathered;
    }
  }
  const bool ok = e == hipSuccess && jjhost::combine_records(j->host, j->nrec, jjhost::REC_MAX_BYTES, &total);
  JJ_ENTER(c);
  int rc = JJ_OK;
  if (e != hipSuccess) { c->err = std::string("hipEventSynchronize failed: ") + hipGetErrorString(e); rc = JJ_ERR_HIP; }
  else if (!ok) { c->err = "MSM record is damaged (bad header)"; rc = JJ_ERR_HIP; }
  else if (is_device_ptr(out64)) {
    jjhost::to_affine64(c->host_out[c->host_out_next], total);
    e = hipMemcpyAsync(out64, c->host_out[c->host_out_next], 64, hipMemcpyHostToDevice, c->stream);
    c->host_out_next = (c->host_out_next + 1) % 8;
    if (e != hipSuccess) { c->err = std::string("hipMemcpyAsync failed: ") + hipGetErrorString(e); rc = JJ_ERR_HIP; }
  } else jjhost::to_affine64((uint8_t*)out64, total);
  msm_job_put(c, j);
  return rc;
}
JJ_API int jj_msm(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out64) {
  if (!c || !out64) return JJ_ERR_INVALID;
  jj_msm_job* j = nullptr;
  {
    JJ_ENTER(c);
    prof_mark(c, 0);
    const int rc = msm_begin_locked(c, n, scalars, points, 0, 1, false, &j);
    if (rc) return rc;
  }
  const int rc = jj_msm_finish(j, out64);
  { JJ_ENTER(c); prof_mark(c, 1); prof_mark(c, 2); }
  return rc;
}
// Opt-in device-side finish: the record stays on the device, one quad runs the Horner chain and the inversion there, the affine sum
// is written to DEVICE memory; nothing is copied to the host and no host thread waits (fully asynchronous on the context's stream).
JJ_API int jj_msm_dev(jj_ctx* c, size_t n, const void* scalars, const void* points, void* out64_dev) {
  if (!c || !out64_dev) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  if (!is_device_ptr(out64_dev) || ((uintptr_t)out64_dev & 15u)) { c->err = "jj_msm_dev writes its result to (16-byte aligned) device memory; use jj_msm for a host result"; return JJ_ERR_INVALID; }
  if (n > ((size_t)1 << c->msm_pass_log2)) { c->err = "jj_msm_dev takes at most one pass of terms (2^24); use jj_msm, or add the sums of the parts with jj_point_add"; return JJ_ERR_INVALID; }
  if (n == 0) { HIPCHK(c, hipMemcpyAsync(out64_dev, AFFINE_IDENTITY_BYTES, 64, hipMemcpyHostToDevice, c->stream)); return JJ_OK; }
  int rc;
  const void *ds, *dp;
  if ((rc = stage_in(c, 0, scalars, 32 * n, &ds))) return rc;
  if ((rc = stage_in(c, 1, points, 64 * n, &dp))) return rc;
  MsmLane* L = nullptr;
  if ((rc = msm_lane(c, 0, &L))) return rc;
  if ((rc = ensure(c, L->rec, JJ_MSM_PARTIAL_BYTES))) return rc;
  size_t used = 0;
  if ((rc = msm_enqueue(c, *L, n, ds, dp, 0, 1, L->rec.p, &used))) return rc;
  hipLaunchKernelGGL(k_msm_finish_dev, dim3(1), dim3(64), 0, c->stream, (const u32*)L->rec.p, out64_dev);
  return finish(c, false);
}
// First half of an MSM that is cut across devices or ranks: the record of partial window sums, left where the
// caller wants it (device memory: ready for an all_gather over RCCL; host memory: the call waits for the copy).
//   part_index / part_count = 0 / 1   all windows of the n terms given (term partition: every rank passes its own terms)
//   part_index = g, part_count = G    windows g, g + G, ... of the n terms given (window partition: every rank passes ALL terms)
JJ_API int jj_msm_partial(jj_ctx* c, size_t n, const void* scalars, const void* points, int part_index, int part_count, void* record) {
  if (!c || !record || part_count < 1 || part_index < 0 || part_index >= part_count) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  if (n > ((size_t)1 << c->msm_pass_log2)) { c->err = "jj_msm_partial takes at most one pass of terms (2^24); cut larger inputs"; return JJ_ERR_INVALID; }
  int rc; OutRef o;
  if ((rc = stage_out(c, c->out[0], record, JJ_MSM_PARTIAL_BYTES, &o))) return rc;
  static_assert(JJ_MSM_PARTIAL_BYTES == jjhost::REC_MAX_BYTES, "record size");
  HIPCHK(c, hipMemsetAsync(o.dev, 0, JJ_MSM_PARTIAL_BYTES, c->stream));
  // the window count this call's layout has (as msm_enqueue picks it); a window partition over more parts than windows leaves
  // the parts beyond the last window nothing to do (zero-sized grids would fail the launches and no header would be written)
  const int layout_W = n <= (size_t)c->msm_small_max ? SM_W : msm_windows_for(c, n);
  if (n == 0 || part_index >= layout_W) {
    // an empty shard: a valid record without windows
    uint32_t hdr[MSM_REC_HDR_WORDS] = {MSM_REC_MAGIC, 2u, (uint32_t)(n == 0 ? SM_W : layout_W), 1u};
    hdr[6] = (uint32_t)n; hdr[7] = (uint32_t)((uint64_t)n >> 32);
    memcpy(c->host_out[c->host_out_next], hdr, 64);
    HIPCHK(c, hipMemcpyAsync(o.dev, c->host_out[c->host_out_next], 64, hipMemcpyHostToDevice, c->stream));
    c->host_out_next = (c->host_out_next + 1) % 8;
  } else {
    const void *ds, *dp;
    if ((rc = stage_in(c, 0, scalars, 32 * n, &ds))) return rc;
    if ((rc = stage_in(c, 1, points, 64 * n, &dp))) return rc;
    size_t used = 0;
    MsmLane* L = nullptr;
    if ((rc = msm_lane(c, 0, &L))) return rc;
    if ((rc = msm_enqueue(c, *L, n, ds, dp, part_index, part_count, o.dev, &used))) return rc;
  }
  bool sync = false;
  if ((rc = finish_out(c, o, &sync))) return rc;
  return finish(c, sync);
}
// Second half: `count` records (HOST memory, JJ_MSM_PARTIAL_BYTES apart: what the ranks' all_gather delivered, copied back once)
// -> one affine point.  Host only, no context: window sums of all records, one Horner chain per window layout, one inversion.
JJ_API int jj_msm_combine(size_t count, const void* records_host, void* out64_host) {
  if (!out64_host || (count && !records_host) || is_device_ptr(out64_host) || (count && is_device_ptr(records_host))) return JJ_ERR_INVALID;
  jjhost::Ext total = jjhost::identity();
  if (!jjhost::combine_records((const uint8_t*)records_host, count, JJ_MSM_PARTIAL_BYTES, &total)) return JJ_ERR_INVALID;
  jjhost::to_affine64((uint8_t*)out64_host, total);
  return JJ_OK;
}

// ---- multi-rank MSM behind the C ABI ("context: streams, tables, RCCL comm"; 8(e)).  The communicator is the caller's
// (its rendezvous -- who carries the ncclUniqueId to whom -- belongs to the application: examples/msm_rccl.cpp does it with a file,
// bench.py over torch.distributed); the context borrows it.  libjubjub_hip.so does not link RCCL: ncclAllGather is taken from the
// caller, or looked up in the process, or in librccl.so.1 -- it must be the ncclAllGather of the library that made the communicator.
JJ_API int jj_ctx_set_comm(jj_ctx* c, void* nccl_comm, int rank, int nranks, void* all_gather_fn) {
  if (!c) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  if (!nccl_comm) { c->comm = nullptr; c->comm_rank = 0; c->comm_nranks = 1; c->all_gather = nullptr; return JJ_OK; }   // detach
  if (nranks < 1 || nranks > 4096 || rank < 0 || rank >= nranks) { c->err = "jj_ctx_set_comm: bad rank / nranks"; return JJ_ERR_INVALID; }
  void* fn = all_gather_fn;
  if (!fn) fn = dlsym(RTLD_DEFAULT, "ncclAllGather");
  if (!fn) { void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL); if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL); if (h) fn = dlsym(h, "ncclAllGather"); }
  if (!fn) { c->err = "jj_ctx_set_comm: ncclAllGather not found (pass its address, or load librccl first)"; return JJ_ERR_INVALID; }
  c->comm = nccl_comm; c->comm_rank = rank; c->comm_nranks = nranks; c->all_gather = (jj_ctx::AllGatherFn)fn;
  // With more than one rank the jobs in flight stay on ONE lane: consecutive all-gathers of one communicator are then queued on one stream, in
  // the same order on every rank.  Gathers alternating over streams rely on RCCL ordering collectives of a communicator across streams in
  // submission order -- seen with one rank and with the loopback stand-in only, never between GPUs (no multi-GPU box in five rounds).  A caller
  // that has checked it on its node sets option msm_lanes back to 2..4 AFTER this call.
  if (nranks > 1) c->msm_lanes = 1;
  return JJ_OK;
}
// `count` records in DEVICE memory (JJ_MSM_PARTIAL_BYTES apart: what an all_gather delivered) -> the sum of the MSMs they stand for.
// From msm_fold_min records (option, default 8) they are folded window by window ON THE DEVICE into one (k_msm_fold_records), that one is
// copied to the host (8 KB, whatever count is) and takes the single-record host tail.  Fewer records, records of different window layouts, or
// option msm_fold_dev = 0: all records are copied and the host adds them (round 4's path).  The context's lock is held by the caller.
static int msm_combine_dev_locked(jj_ctx* c, size_t count, const uint8_t* recs_dev, jjhost::Ext* total) {
  *total = jjhost::identity();
  if (count == 0) return JJ_OK;
  const size_t host_need = std::max<size_t>(count, 1) * JJ_MSM_PARTIAL_BYTES;
  if (c->gather_host_cap < host_need) {
    if (c->gather_host) (void)hipHostFree(c->gather_host);
    c->gather_host = nullptr; c->gather_host_cap = 0;
    // (coherent: the fold kernel writes its record straight into this buffer, as the MSM kernels write theirs into a job's)
    if (hipHostMalloc((void**)&c->gather_host, host_need, hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); c->err = "hipHostMalloc failed"; return JJ_ERR_NOMEM; }
    c->gather_host_cap = host_need;
  }
  bool folded = false;
  if (c->msm_fold_dev && count >= (size_t)c->msm_fold_min) {
    memset(c->gather_host, 0, jjhost::REC_HDR_BYTES);               // a stale header must never validate
    hipLaunchKernelGGL(k_msm_fold_records, dim3(jjhost::REC_MAX_W), dim3(4 * MSM_TREE_QUADS), 0, c->stream, (const u32*)recs_dev, (u32)count, (u32)(JJ_MSM_PARTIAL_BYTES / 4), (u32*)c->gather_host);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint32_t magic; memcpy(&magic, c->gather_host, 4);
    folded = magic == MSM_REC_MAGIC;                    // 0: the records have different window layouts (or one is damaged: the host path reports it)
  }
  if (folded) { if (!jjhost::combine_records(c->gather_host, 1, JJ_MSM_PARTIAL_BYTES, total)) { c->err = "the folded MSM record is damaged (bad header)"; return JJ_ERR_HIP; } return JJ_OK; }
  HIPCHK(c, hipMemcpyAsync(c->gather_host, recs_dev, count * JJ_MSM_PARTIAL_BYTES, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (!jjhost::combine_records(c->gather_host, count, JJ_MSM_PARTIAL_BYTES, total)) { c->err = "a gathered MSM record is damaged (bad header)"; return JJ_ERR_HIP; }
  return JJ_OK;
}
static int msm_write_total(jj_ctx* c, const jjhost::Ext& total, void* out64) {
  if (is_device_ptr(out64)) {
    jjhost::to_affine64(c->host_out[c->host_out_next], total);
    HIPCHK(c, hipMemcpyAsync(out64, c->host_out[c->host_out_next], 64, hipMemcpyHostToDevice, c->stream));
    c->host_out_next = (c->host_out_next + 1) % 8;
  } else jjhost::to_affine64((uint8_t*)out64, total);
  return JJ_OK;
}
JJ_API int jj_msm_combine_dev(jj_ctx* c, size_t count, const void* records_dev, void* out64) {
  if (!c || !out64 || (count && !records_dev)) return JJ_ERR_INVALID;
  JJ_ENTER(c);
  if (count && (!is_device_ptr(records_dev) || ((uintptr_t)records_dev & 15u))) { c->err = "jj_msm_combine_dev takes records in (16-byte aligned) device memory; jj_msm_combine takes host records"; return JJ_ERR_INVALID; }
  jjhost::Ext total;
  const int rc = msm_combine_dev_locked(c, count, (const uint8_t*)records_dev, &total);
  if (rc) return rc;
  return msm_write_total(c, total, out64);
}
// A rank that fails BEFORE its all-gather (its term count is over the pass limit, a workspace cannot grow, a launch fails) must not leave the
// other ranks waiting in theirs: it still takes part, with an all-zero record -- no valid header, so every rank's fold / host tail rejects the
// set and every rank's call returns an error ("a gathered MSM record is damaged") instead of hanging or summing without this rank's terms.
// Best effort: if even this gather cannot be queued, the communicator is lost, as after any failed RCCL collective.
static void msm_post_poison(jj_ctx* c, hipStream_t st) {
  (void)hipGetLastError();
  if (!c->comm || !c->all_gather) return;
  const size_t G = (size_t)c->comm_nranks;
  if (ensure(c, c->poison_dev, (G + 1) * JJ_MSM_PARTIAL_BYTES) != JJ_OK) return;
  uint8_t* mine = (uint8_t*)c->poison_dev.p;
  if (hipMemsetAsync(mine, 0, JJ_MSM_PARTIAL_BYTES, st) != hipSuccess) { (void)hipGetLastError(); return; }
  (void)c->all_gather(mine, mine + JJ_MSM_PARTIAL_BYTES, JJ_MSM_PARTIAL_BYTES, /* ncclUint8 */ 1, c->comm, st);
  (void)hipStreamSynchronize(st);
  (void)hipGetLastError();
}
// jj_msm_allgather in two halves (as jj_msm_begin / jj_msm_finish for the one-GPU sum): everything up to the folded record is queued on
// one of the context's lanes -- the rank's window sums, the ncclAllGather (stream-ordered like any kernel), the fold of the G records
// into one written straight into the job's page-locked buffer -- and the call returns; jj_msm_finish waits for that job and runs the
// single-record host tail.  With several jobs in flight the gather, the fold, the wait and the host tail of one MSM run beside the
// kernels of the next: a rank's sustained rate is that of its kernels, not of the call's latency.
JJ_API int jj_msm_allgather_begin(jj_ctx* c, size_t n, const void* scalars, const void* points, int partition, jj_msm_job** job) {
  if (!c || !job || (partition != 0 && partition != 1)) return JJ_ERR_INVALID;
  *job = nullptr;
  JJ_ENTER(c);
  if (!c->comm || !c->all_gather) { c->err = "jj_msm_allgather_begin: no communicator (jj_ctx_set_comm)"; return JJ_ERR_INVALID; }
  // (every failure from here to the gather posts a poison record: the other ranks are already on their way into the collective)
  if (n > ((size_t)1 << c->msm_pass_log2)) { msm_post_poison(c, c->stream); c->err = "jj_msm_allgather_begin takes at most one pass of terms (2^24) per rank; cut larger inputs"; return JJ_ERR_INVALID; }
  const int G = c->comm_nranks;
  const int part_index = partition ? c->comm_rank : 0, part_count = partition ? G : 1;
  jj_msm_job* j;
  int rc = msm_job_get(c, (size_t)G, &j); if (rc) { const std::string keep = c->err; msm_post_poison(c, c->stream); c->err = keep; return rc; }
  const size_t need = (size_t)(G + 1) * JJ_MSM_PARTIAL_BYTES;
  if (j->gdev_cap < need) {
    if (j->gdev) (void)hipFree(j->gdev);
    j->gdev = nullptr; j->gdev_cap = 0;
    if (hipMalloc(&j->gdev, need) != hipSuccess) { (void)hipGetLastError(); msm_job_put(c, j); msm_post_poison(c, c->stream); c->err = "hipMalloc failed"; return JJ_ERR_NOMEM; }
    j->gdev_cap = need;
  }
  uint8_t* mine = (uint8_t*)j->gdev;                                // this rank's record, then the G gathered ones
  uint8_t* all = mine + JJ_MSM_PARTIAL_BYTES;
  int k = 0;
  if (n && c->msm_lanes > 1 && is_device_ptr(scalars) && is_device_ptr(points)) k = 1 + (int)(c->next_lane++ % (unsigned)c->msm_lanes);
  MsmLane* L = nullptr;
  if ((rc = msm_lane(c, k, &L))) { const std::string keep = c->err; msm_job_put(c, j); msm_post_poison(c, c->stream); c->err = keep; return rc; }
  bool gathered = false;          // a failure before the gather posts the poison record in its place
  auto fail = [&](int code) { const std::string keep = c->err; (void)hipStreamSynchronize(L->stream); (void)hipGetLastError(); if (!gathered) msm_post_poison(c, L->stream); msm_job_put(c, j); c->err = keep; return code; };
  hipError_t e = hipMemsetAsync(mine, 0, JJ_MSM_PARTIAL_BYTES, L->stream);
  if (e != hipSuccess) { c->err = std::string("hipMemsetAsync failed: ") + hipGetErrorString(e); return fail(JJ_ERR_HIP); }
  const int layout_W = n <= (size_t)c->msm_small_max ? SM_W : msm_windows_for(c, n);
  if (n == 0 || part_index >= layout_W) {
    // an empty shard: a valid record without windows (as jj_msm_partial writes it)
    uint32_t hdr[MSM_REC_HDR_WORDS] = {MSM_REC_MAGIC, 2u, (uint32_t)(n == 0 ? SM_W : layout_W), 1u};
    hdr[6] = (uint32_t)n; hdr[7] = (uint32_t)((uint64_t)n >> 32);
    memcpy(c->host_out[c->host_out_next], hdr, 64);
    e = hipMemcpyAsync(mine, c->host_out[c->host_out_next], 64, hipMemcpyHostToDevice, L->stream);
    c->host_out_next = (c->host_out_next + 1) % 8;
    if (e != hipSuccess) { c->err = std::string("hipMemcpyAsync failed: ") + hipGetErrorString(e); return fail(JJ_ERR_HIP); }
  } else {
    const void *ds, *dp;
    size_t used = 0;
    if ((rc = stage_in(c, 0, scalars, 32 * n, &ds)) || (rc = stage_in(c, 1, points, 64 * n, &dp))) return fail(rc);
    if ((rc = msm_enqueue(c, *L, n, ds, dp, part_index, part_count, mine, &used))) return fail(rc);
  }
  const int nrc = c->all_gather(mine, all, JJ_MSM_PARTIAL_BYTES, /* ncclUint8 */ 1, c->comm, L->stream);
  gathered = true;                // (a failure of the collective itself is fatal for the communicator, as with any RCCL collective)
  if (nrc != 0) { c->err = "ncclAllGather failed with ncclResult_t " + std::to_string(nrc); return fail(JJ_ERR_HIP); }
  j->gathered = G;
  if (c->msm_fold_dev && G >= c->msm_fold_min) {
    hipLaunchKernelGGL(k_msm_fold_records, dim3(jjhost::REC_MAX_W), dim3(4 * MSM_TREE_QUADS), 0, L->stream, (const u32*)all, (u32)G, (u32)(JJ_MSM_PARTIAL_BYTES / 4), (u32*)j->host);
    j->folded = true; j->nrec = 1;
  } else {
    e = hipMemcpyAsync(j->host, all, (size_t)G * JJ_MSM_PARTIAL_BYTES, hipMemcpyDeviceToHost, L->stream);
    if (e != hipSuccess) { c->err = std::string("hipMemcpyAsync failed: ") + hipGetErrorString(e); return fail(JJ_ERR_HIP); }
    j->nrec = (size_t)G;
  }
  e = hipEventRecord(j->ev, L->stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) { c->err = std::string("MSM launch failed: ") + hipGetErrorString(e); return fail(JJ_ERR_HIP); }
  *job = j;
  return JJ_OK;
}
// One MSM over the terms (partition 0: each rank passes ITS terms) or the windows (partition 1: each rank passes ALL terms) of every
// rank of the communicator: record of window sums on this device -> ncclAllGather of JJ_MSM_PARTIAL_BYTES per rank over xGMI -> fold on
// the device (from msm_fold_min records) or ONE copy of the gathered records -> ONE host tail on every rank.  Every rank gets the same point.
JJ_API int jj_msm_allgather(jj_ctx* c, size_t n, const void* scalars, const void* points, int partition, void* out64) {
  if (!c || !out64 || (partition != 0 && partition != 1)) return JJ_ERR_INVALID;
  JJ_ENTER(c);                                                       // held through the gather and the host tail (jj_msm_partial re-enters it: the mutex is recursive)
  if (!c->comm || !c->all_gather) { c->err = "jj_msm_allgather: no communicator (jj_ctx_set_comm)"; return JJ_ERR_INVALID; }
  const int G = c->comm_nranks;
  int rc;
  if ((rc = ensure(c, c->gather_dev, (size_t)(G + 1) * JJ_MSM_PARTIAL_BYTES))) { const std::string keep = c->err; msm_post_poison(c, c->stream); c->err = keep; return rc; }
  uint8_t* mine = (uint8_t*)c->gather_dev.p;                       // this rank's record, then the G gathered ones
  uint8_t* all = mine + JJ_MSM_PARTIAL_BYTES;
  if ((rc = jj_msm_partial(c, n, scalars, points, partition ? c->comm_rank : 0, partition ? G : 1, mine))) { const std::string keep = c->err; msm_post_poison(c, c->stream); c->err = keep; return rc; }
  const int nrc = c->all_gather(mine, all, JJ_MSM_PARTIAL_BYTES, /* ncclUint8 */ 1, c->comm, c->stream);
  if (nrc != 0) { c->err = "ncclAllGather failed with ncclResult_t " + std::to_string(nrc); return JJ_ERR_HIP; }
  jjhost::Ext total;
  if ((rc = msm_combine_dev_locked(c, (size_t)G, all, &total))) return rc;
  return msm_write_total(c, total, out64);
}

