// Multi-scalar multiplication kernels (gfx950).  Included by jj_kernels.h after the point records and the quad-lane point
// operations; launched by msm_* in jj_msm.hip.
//
// sum_i k_i P_i  (reference semantics: the iterator `Sum` of `p * k`, src/lib.rs:183-193 + 873-879; the reference has no MSM
// algorithm of its own).  Only +-P is used, so the result is exact on the whole curve (cofactor-8 points, scalars >= r).
//
// Windows.  k (252 bits) is recoded as k' = k + sum_w 2^(start_w + width_w - 1) over all windows but the top one; W windows
// TILE the 253 bits of k' EXACTLY (253 = W c + r: the r low windows are c + 1 bits wide, the others c bits), so there is never a
// short top window whose few buckets would collect n entries each (round 2 needed c | 253, i.e. c = 11, for that).  Window w < W-1
// holds the signed digit raw - 2^(width-1) in [-2^(width-1), 2^(width-1)); the top window is unsigned and, because k < 2^252, at
// most 2^(width-1): every window needs 2^(width-1) buckets (bucket |d| - 1), B = 2^(c_max - 1) slots per window are allocated.
//
// A pass may own a SUBSET of the windows (w0 + s * wstride, s < Ws: the by-window partition of a multi-GPU MSM);
// everything below is indexed by the slot s, only the digit extraction uses the window w itself.
//
// Two algorithms produce the same output record (one point per window: U, V, Z and T = T1 T2 as 4 x 32 bytes in the host tail's
// own Montgomery form, radix 2^256; the host runs Horner over the windows and inverts once: jj_host_tail.h):
//   small batches (<= ~2^14 terms)   k_msm_small_tables + k_msm_small_sum: per-term table {0..8}P, 64 windows of 3-4 bits, every
//                                    window is a tree sum over the terms' table entries on quads of lanes -- two launches
//   Pippenger                        k_msm_convert; counting sort by (window, |digit|) without global atomics, one pass with the
//                                    window's histogram in LDS (k_msm_hist / _plan / _scatter) or two passes for >= 4096 buckets
//                                    per window (k_msm_part_hist / _plan / _scatter / _sort); bucket accumulation over fixed
//                                    chunks (k_msm_accumulate + k_msm_fixup) or length-sorted segments (k_seg_*,
//                                    k_msm_accumulate_seg); k_msm_fixup_big (+ the segment path's merge list); k_msm_reduce_fold
#pragma once
// (inside namespace jj: this file is included from the middle of jj_kernels.h)

// The kernels that end in a window sum are chains of dependent point operations on quads of lanes: they run fastest at ONE wave
// per SIMD (measured: a second wave per SIMD slows each by ~1.45x), so their workgroups are 256 threads = 64 quads, as many
// workgroups per window as the work needs, and the LAST workgroup of a window to finish (a device-side counter) folds the
// window's partial sums and writes the window's point into the record.
constexpr int MSM_TREE_QUADS = 64;
constexpr int MSM_REC_HDR_WORDS = 16;             // 64-byte header: magic, version, W, 1, window mask (2 words), n (2 words)
constexpr int MSM_REC_PT_WORDS = 32;              // one window: U, V, Z, T, 8 words each
constexpr int MSM_PART_WORDS = 56;                // a workgroup's partial sum on its way to the window's last workgroup: U V Z T1 T2 T + pad
constexpr u32 MSM_REC_MAGIC = 0x504D4A4Au;        // "JJMP"

struct MsmParams {
  int W;                  // windows tiling bits [0, 253) of k'
  int c, r;               // 253 = W c + r: windows w < r are c + 1 bits wide, the others c bits
  int Ws, w0, wstride;    // windows of this pass: w0 + s * wstride for s < Ws
  u32 B;                  // bucket slots per window = 2^(cmax - 1), cmax = c + (r > 0)
  u32 recode[8];          // sum over w < W - 1 of 2^(start_w + width_w - 1)
};
static JJ_DEV int msm_win_start(const MsmParams& mp, int w) { return w < mp.r ? w * (mp.c + 1) : mp.r * (mp.c + 1) + (w - mp.r) * mp.c; }
static JJ_DEV int msm_win_width(const MsmParams& mp, int w) { return mp.c + (w < mp.r ? 1 : 0); }
static JJ_DEV int msm_slot_window(const MsmParams& mp, int s) { return mp.w0 + s * mp.wstride; }
// signed digit of window w from its raw bits: returns |d| (0 = no contribution) and the sign
static JJ_DEV u32 msm_digit_raw(u32 raw, const MsmParams& mp, int w, int width, u32& neg) {
  if (w == mp.W - 1) { neg = 0; return raw; }
  const int d = (int)raw - (int)(1u << (width - 1));
  neg = d < 0 ? 1u : 0u;
  return (u32)(d < 0 ? -d : d);
}
// the recoded scalars k' are kept word-major (kp[j * n + i] = word j of term i) so that a block working on one
// window reads just the one or two words that hold it, coalesced
static JJ_DEV u32 msm_digit_wm(const u32* kp, size_t n, size_t i, const MsmParams& mp, int w, u32& neg) {
  const int bit = msm_win_start(mp, w), width = msm_win_width(mp, w), wi = bit >> 5, sh = bit & 31;
  u64 both = kp[(size_t)wi * n + i];
  if (sh + width > 32 && wi < 7) both |= (u64)kp[(size_t)(wi + 1) * n + i] << 32;
  return msm_digit_raw((u32)(both >> sh) & ((1u << width) - 1u), mp, w, width, neg);
}
// the same from the eight words of one k' held in registers
static JJ_DEV u32 msm_digit_reg(const u32 (&k)[8], const MsmParams& mp, int w, u32& neg) {
  const int bit = msm_win_start(mp, w), width = msm_win_width(mp, w), wi = bit >> 5, sh = bit & 31;
  u32 lo = k[0], hi = k[1];
  _Pragma("unroll") for (int q = 1; q < 8; q++) { lo = (wi == q) ? k[q] : lo; hi = (wi == q) ? (q < 7 ? k[q + 1] : 0u) : hi; }
  const u64 both = ((u64)hi << 32) | lo;
  return msm_digit_raw((u32)(both >> sh) & ((1u << width) - 1u), mp, w, width, neg);
}
static JJ_DEV void msm_recode(u32 (&k)[8], const MsmParams& mp) {
  k[7] &= 0x0fffffffu;
  u64 cy = 0;
  _Pragma("unroll") for (int j = 0; j < 8; j++) { const u64 t = (u64)k[j] + mp.recode[j] + cy; k[j] = (u32)t; cy = t >> 32; }
}
// header of the output record, written by one thread of the kernel that produces the window sums
static JJ_DEV void msm_write_header(u32* hdr, const MsmParams& mp, size_t n) {
  u64 mask = 0;
  for (int s = 0; s < mp.Ws; s++) mask |= 1ull << msm_slot_window(mp, s);
  hdr[0] = MSM_REC_MAGIC; hdr[1] = 2u; hdr[2] = (u32)mp.W; hdr[3] = 1u;
  hdr[4] = (u32)mask; hdr[5] = (u32)(mask >> 32); hdr[6] = (u32)n; hdr[7] = (u32)((u64)n >> 32);
  for (int j = 8; j < MSM_REC_HDR_WORDS; j++) hdr[j] = 0;
}
// a window's point leaves as (U, V, Z, T = T1 T2), each the canonical integer of value * 2^256 mod q: the host tail's Montgomery
// form (jj_host_tail.h), so the host starts its Horner chain without a single conversion product
static JJ_DEV void msm_store_window(u32* points, int w, const Ext& acc, const Fe& T) {
  const Fe hr = Fq::konst(FqP::HOST_R);
  u32* dst = points + (size_t)w * MSM_REC_PT_WORDS;
  u32 wd[8];
  Fq::pack(wd, Fq::canon_plain_product(Fq::mul(acc.u, hr))); store8(dst, 0, wd);
  Fq::pack(wd, Fq::canon_plain_product(Fq::mul(acc.v, hr))); store8(dst, 1, wd);
  Fq::pack(wd, Fq::canon_plain_product(Fq::mul(acc.z, hr))); store8(dst, 2, wd);
  Fq::pack(wd, Fq::canon_plain_product(Fq::mul(T, hr))); store8(dst, 3, wd);
}

// ---- quad-lane tree over the MSM_TREE_QUADS quads of a 256-thread workgroup through LDS: every quad hands in a point and
// T = t1 * t2; quad 0 ends up with the sum of the first `live` quads' points.  6 levels of three-round additions.
// A point waits in LDS in the form the first multiplication round of the addition consumes, (V - U, V + U, T, Z): lane r of
// the receiving quad reads coordinate r only (one indexed read, 9 words), which keeps the tree at ~100 VGPRs.
constexpr int LDS_PT_WORDS = 4 * NL;     // 36 words = 144 B per waiting point
static JJ_DEV void lds_put_pt(u32* st, u32 slot, const Ext& e, const Fe& T) {
  const Fe vmu = Fq::sub(e.v, e.u), vpu = Fq::carry(Fq::add(e.v, e.u));
  u32* p = st + (size_t)slot * LDS_PT_WORDS;
  _Pragma("unroll") for (int l = 0; l < NL; l++) { p[l] = vmu.l[l]; p[NL + l] = vpu.l[l]; p[2 * NL + l] = T.l[l]; p[3 * NL + l] = e.z.l[l]; }
}
// p + q for q waiting in LDS; same rounds as quad_add_ext_t without the side product
static JJ_DEV Ext quad_add_lds_t(const Ext& p, const Fe& Tp, const u32* st, u32 slot, u32 role, Fe& Tout) {
  const u32* q = st + (size_t)slot * LDS_PT_WORDS + role * NL;
  Fe qb;
  _Pragma("unroll") for (int l = 0; l < NL; l++) qb.l[l] = q[l];
  const Fe r1 = Fq::mul(role_select4(Fq::sub(p.v, p.u), Fq::add(p.v, p.u), Tp, p.z, role), qb);
  const Fe a = quad_bcast<0>(r1), b = quad_bcast<1>(r1), tt = quad_bcast<2>(r1), zz = quad_bcast<3>(r1);
  const Fe c = Fq::mul(tt, Fq::konst(FqP::D2));
  return quad_add_finish(a, b, c, Fq::add(zz, zz), role, Tout);
}
static JJ_DEV void quad_tree_sum(u32* st, u32 quad, u32 role, u32 live, Ext& acc, Fe& T) {
  #pragma unroll 1
  for (u32 s = MSM_TREE_QUADS / 2; s > 0; s >>= 1) {
    if (s >= live) continue;                                        // uniform over the workgroup: nothing to hand over at this level
    if (quad >= s && quad < 2 * s && quad < live && role == 0) lds_put_pt(st, quad, acc, T);     // upper half hands over
    __syncthreads();
    if (quad < s && quad + s < live) acc = quad_add_lds_t(acc, T, st, quad + s, role, T);
    __syncthreads();
  }
}

// Workgroup (blk, s) of nblk holds its sum in quad 0.  With several workgroups per window each parks its sum in `part`; the last
// one to arrive (counter MSM_COUNTERS + s, cleared by the first kernel of the pass) folds all of them and writes the window's point.
constexpr int MSM_COUNTERS = 8;                       // [0] heads, [1] merge items, [2] big buckets, [3] big-bucket blocks done; then one arrival counter per slot
constexpr int MSM_COUNTER_WORDS = MSM_COUNTERS + 64;
// (nblk workgroups of this window; `stride` part slots per window, the same for every window of the pass)
static JJ_DEV void msm_finish_window(u32* st, u32 quad, u32 role, u32 nblk, u32 stride, u32 blk, u32 s, int w, Ext& acc, Fe& T, u32* part, u32* counters, u32* rec) {
  __shared__ u32 last_s;
  if (nblk > 1) {
    if (quad == 0 && role == 0) {
      const Fe t1 = Fq::carry(acc.t1), t2 = Fq::carry(acc.t2);
      u32* p = part + ((size_t)s * stride + blk) * MSM_PART_WORDS;
      _Pragma("unroll") for (int l = 0; l < NL; l++) { p[l] = acc.u.l[l]; p[NL + l] = acc.v.l[l]; p[2 * NL + l] = acc.z.l[l]; p[3 * NL + l] = t1.l[l]; p[4 * NL + l] = t2.l[l]; p[5 * NL + l] = T.l[l]; }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last_s = atomicAdd(&counters[MSM_COUNTERS + s], 1u) == nblk - 1 ? 1u : 0u;
    __syncthreads();
    if (!last_s) return;
    __threadfence();
    if (quad < nblk) {
      const u32* p = part + ((size_t)s * stride + quad) * MSM_PART_WORDS;
      _Pragma("unroll") for (int l = 0; l < NL; l++) { acc.u.l[l] = p[l]; acc.v.l[l] = p[NL + l]; acc.z.l[l] = p[2 * NL + l]; acc.t1.l[l] = p[3 * NL + l]; acc.t2.l[l] = p[4 * NL + l]; T.l[l] = p[5 * NL + l]; }
    }
    quad_tree_sum(st, quad, role, nblk, acc, T);
  }
  if (quad == 0 && role == 0) msm_store_window(rec + MSM_REC_HDR_WORDS, w, acc, T);
}

// ================================================================================================ small batches
// One quad of lanes per term builds the term's table {0 .. 8} P (extended-Niels, 144 B per entry, entry 0 = the identity so
// that a zero digit is a plain table read) and stores the recoded scalar; then one workgroup per (window, block of terms) adds
// up the entries the terms' digits select: every quad takes a strided share of the terms (two multiplication rounds per
// addition), the quads of a workgroup are folded through LDS, and the last workgroup of the window folds the workgroups' sums.  With W = 64 the windows are 3 or 4 bits wide: digits in [-8, 8].
constexpr int SM_W = 64;                 // windows of the small-batch layout (253 = 64 * 3 + 61: 61 windows of 4 bits, 3 of 3 bits)
constexpr int SM_SLOTS = 9;              // table entries per term: multiples 0 .. 8
__global__ void __launch_bounds__(256) k_msm_small_tables(size_t n, const void* scalars, const void* points, MsmParams mp, u32* tables, u32* kprime, u32* counters) {
  if (blockIdx.x == 0 && threadIdx.x < MSM_COUNTER_WORDS) counters[threadIdx.x] = 0;
  const size_t q = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const u32 role = threadIdx.x & 3u;
  if (q >= n) return;                                       // whole quads leave together
  u32 k[8];
  load8(k, scalars, q);
  msm_recode(k, mp);
  if (role == 0) store8(kprime, q, k);
  const Affine P = load_affine(points, q);
  const ANiels pn = Curve::to_niels(P);
  Ext cur = Curve::from_affine(P);
  Fe T = Fq::mul(P.u, P.v);
  u32* slot = tables + q * (size_t)(SM_SLOTS * ENIELS_WORDS);
  ENiels en;
  en.vpu = pn.vpu; en.vmu = pn.vmu; en.z2 = Fq::add(Fq::one(), Fq::one()); en.t2d = pn.t2d;
  if (role == 0) { store_eniels(slot, Curve::eniels_identity()); store_eniels(slot + ENIELS_WORDS, en); }
  #pragma unroll 1
  for (int j = 2; j < SM_SLOTS; j++) {
    cur = quad_add_aniels(cur, T, pn, role, T);
    en.vpu = Fq::carry(Fq::add(cur.v, cur.u)); en.vmu = Fq::sub(cur.v, cur.u); en.z2 = Fq::add(cur.z, cur.z);
    en.t2d = Fq::mul(T, Fq::konst(FqP::D2));
    if (role == 0) store_eniels(slot + j * ENIELS_WORDS, en);
  }
}
__global__ void __launch_bounds__(4 * MSM_TREE_QUADS) k_msm_small_sum(size_t n, MsmParams mp, u32 nblk, const u32* tables, const u32* kprime, u32* part, u32* counters, u32* rec) {
  __shared__ __attribute__((aligned(16))) u32 st[MSM_TREE_QUADS * LDS_PT_WORDS];
  const u32 role = threadIdx.x & 3u, quad = threadIdx.x >> 2, blk = blockIdx.x;
  const int w = msm_slot_window(mp, (int)blockIdx.y);
  if (blk == 0 && blockIdx.y == 0 && threadIdx.x == 0) msm_write_header(rec, mp, n);
  const size_t first = (size_t)blk * MSM_TREE_QUADS + quad, stride = (size_t)MSM_TREE_QUADS * nblk;
  Ext acc = Curve::identity();
  Fe T = Fq::zero();
  // the entry of the quad's next term is fetched (index arithmetic on k', then one 144-byte line) while the current one is added
  u32 kq[8];
  ENiels e = Curve::eniels_identity();
  u32 neg = 0;
  if (first < n) {
    load8(kq, kprime, first);
    const u32 a = msm_digit_reg(kq, mp, w, neg);
    e = load_eniels(tables + (first * SM_SLOTS + a) * (size_t)ENIELS_WORDS);
  }
  #pragma unroll 1
  for (size_t i = first; i < n; i += stride) {
    const ENiels cur = e;
    const u32 cneg = neg;
    const size_t nx = i + stride;
    if (nx < n) {
      load8(kq, kprime, nx);
      const u32 a = msm_digit_reg(kq, mp, w, neg);
      e = load_eniels(tables + (nx * SM_SLOTS + a) * (size_t)ENIELS_WORDS);
    }
    acc = quad_add_eniels(acc, T, cur, cneg, role, T);
  }
  const size_t base = (size_t)blk * MSM_TREE_QUADS;
  const u32 live = base >= n ? 0u : (n - base < MSM_TREE_QUADS ? (u32)(n - base) : (u32)MSM_TREE_QUADS);      // quads of this block that hold a term
  quad_tree_sum(st, quad, role, live, acc, T);                                            // identity when the block has no term
  msm_finish_window(st, quad, role, nblk, nblk, blk, blockIdx.y, w, acc, T, part, counters, rec);
}

// ================================================================================================ Pippenger: conversion
// recode scalars (k' word-major) and convert points to affine-Niels AoS (27 words in a 128-byte record)
// what: 1 = scalars, 2 = points, 3 = both (the two halves are independent: the sort needs only the scalars, so the host may run
// the point half on a second stream beside it)
// The 128-byte entries leave through LDS: a lane's own entry is eight 16-byte pieces, and stored lane by lane every store
// instruction would touch 64 different lines with 16 bytes each; staged per wave, consecutive lanes store consecutive pieces (1 KB
// per instruction, whole lines).
__global__ void __launch_bounds__(256) k_msm_convert(size_t n, const void* scalars, const void* points, MsmParams mp, u32* kprime, u32* niels, int what) {
  constexpr int PIECES = GNIELS_WORDS / 4, ROW = PIECES + 1;       // 16-byte pieces per entry; row stride padded against bank conflicts
  __shared__ uint4 stage[4][64 * ROW];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if ((what & 1) && i < n) {
    u32 k[8];
    load8(k, scalars, i);
    msm_recode(k, mp);
    _Pragma("unroll") for (int j = 0; j < 8; j++) kprime[(size_t)j * n + i] = k[j];
  }
  if (!(what & 2)) return;
  const u32 lane = threadIdx.x & 63u;
  uint4* my = stage[threadIdx.x >> 6];
  if (i < n) {
    const ANiels t = Curve::to_niels(load_affine(points, i));
    u32 wv[GNIELS_WORDS];
    _Pragma("unroll") for (int l = 0; l < NL; l++) { wv[l] = t.vpu.l[l]; wv[NL + l] = t.vmu.l[l]; wv[2 * NL + l] = t.t2d.l[l]; }
    _Pragma("unroll") for (int l = ANIELS_WORDS - 1; l < GNIELS_WORDS; l++) wv[l] = 0;
    _Pragma("unroll") for (int v = 0; v < PIECES; v++) my[lane * ROW + v] = make_uint4(wv[4 * v], wv[4 * v + 1], wv[4 * v + 2], wv[4 * v + 3]);
  }
  __syncthreads();
  const size_t r0 = (size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u);          // first entry of this wave
  uint4* out = reinterpret_cast<uint4*>(niels + r0 * GNIELS_WORDS);
  _Pragma("unroll") for (int cpc = 0; cpc < PIECES; cpc++) {
    const u32 q = (u32)cpc * 64u + lane, rec = q / PIECES, piece = q % PIECES;
    if (r0 + rec < n) out[q] = my[rec * ROW + piece];
  }
}

// The same conversion with the COARSE HISTOGRAM of the two-pass sort done on the way (round 5): the recoded scalar is in registers here, so the
// window digits cost no second pass over k' (k_msm_part_hist re-read it: 25 us at 2^20 terms, bound by its 16.7 M LDS atomics, which now overlap this
// kernel's memory traffic), and the per-(tile, bin) counts of k_msm_part_hist / the scan of k_msm_part_plan are not needed at all: the kernel leaves
// only the totals per (window slot, coarse bin) -- one global atomic per workgroup and bin -- and k_msm_part_scatter reserves the run of a tile in a
// bin with one global atomic per (tile, bin).  Workgroups of 1024 threads x 4 terms (4096 terms: 2048 counters flushed per workgroup; 256-thread
// workgroups would flush 7 M atomics onto the same 2048 addresses); per-wave staging of the 128-byte entries as in k_msm_convert.
constexpr int MSM_CH_THREADS = 1024, MSM_CH_PER = 4, MSM_CH_TERMS = MSM_CH_THREADS * MSM_CH_PER;
constexpr int MSM_BINS_WORDS = 64 * 128;                 // totals [slot][bin] (as many again for the cursors), sized for 64 slots x MSM_HB_MAX bins
constexpr int MSM_CH_STAGE_WORDS = (MSM_CH_THREADS / 64) * 64 * (GNIELS_WORDS / 4 + 1) * 4;
__global__ void __launch_bounds__(MSM_CH_THREADS) k_msm_convert_hist(size_t n, const void* scalars, const void* points, MsmParams mp, u32* kprime, u32* niels,
                                                                     u32* totals, u32* clear_next /* the other parity's totals + cursors */, u32* counters, int per /* terms per thread: 1 .. MSM_CH_PER */) {
  constexpr int PIECES = GNIELS_WORDS / 4, ROW = PIECES + 1;
  extern __shared__ __attribute__((aligned(16))) u32 ch_lds[];
  uint4* my = reinterpret_cast<uint4*>(ch_lds) + (size_t)(threadIdx.x >> 6) * 64 * ROW;      // this wave's staging area
  u32* hist = ch_lds + MSM_CH_STAGE_WORDS;
  const u32 HB = mp.B >> 8, NH = (u32)mp.Ws * HB, lane = threadIdx.x & 63u;
  if (blockIdx.x == 0) {
    if (threadIdx.x < MSM_COUNTER_WORDS) counters[threadIdx.x] = 0;                         // (the plan kernel's job on the other sort paths)
    for (u32 j = threadIdx.x; j < 2 * MSM_BINS_WORDS; j += MSM_CH_THREADS) clear_next[j] = 0;
  }
  for (u32 j = threadIdx.x; j < NH; j += MSM_CH_THREADS) hist[j] = 0;
  __syncthreads();
  #pragma unroll 1
  for (int it = 0; it < per; it++) {
    const size_t i = ((size_t)blockIdx.x * per + it) * MSM_CH_THREADS + threadIdx.x;
    if (i < n) {
      u32 k[8];
      load8(k, scalars, i);
      msm_recode(k, mp);
      _Pragma("unroll") for (int j = 0; j < 8; j++) kprime[(size_t)j * n + i] = k[j];
      // the windows tile k' from bit 0 upwards: a copy of k' is shifted right by one window's width per step (eight funnel shifts by a
      // wave-uniform amount), the digit is its low bits -- no indexed access into the eight words
      u32 sh[8];
      _Pragma("unroll") for (int j = 0; j < 8; j++) sh[j] = k[j];
      int next_slot_window = mp.w0, sl = 0;
      #pragma unroll 1
      for (int w = 0; w < mp.W && sl < mp.Ws; w++) {
        const int width = msm_win_width(mp, w);
        if (w == next_slot_window) {
          u32 neg;
          const u32 a = msm_digit_raw(sh[0] & ((1u << width) - 1u), mp, w, width, neg);
          if (a) atomicAdd(&hist[(u32)sl * HB + ((a - 1) >> 8)], 1u);
          sl++; next_slot_window += mp.wstride;
        }
        _Pragma("unroll") for (int j = 0; j < 7; j++) sh[j] = __builtin_amdgcn_alignbit(sh[j + 1], sh[j], (u32)width);      // (a 64-bit shift by a run-time amount is quarter-rate)
        sh[7] >>= width;
      }
      const ANiels t = Curve::to_niels(load_affine(points, i));
      u32 wv[GNIELS_WORDS];
      _Pragma("unroll") for (int l = 0; l < NL; l++) { wv[l] = t.vpu.l[l]; wv[NL + l] = t.vmu.l[l]; wv[2 * NL + l] = t.t2d.l[l]; }
      _Pragma("unroll") for (int l = ANIELS_WORDS - 1; l < GNIELS_WORDS; l++) wv[l] = 0;
      _Pragma("unroll") for (int v = 0; v < PIECES; v++) my[lane * ROW + v] = make_uint4(wv[4 * v], wv[4 * v + 1], wv[4 * v + 2], wv[4 * v + 3]);
    }
    __syncthreads();
    const size_t r0 = ((size_t)blockIdx.x * per + it) * MSM_CH_THREADS + (threadIdx.x & ~63u);      // first entry of this wave
    uint4* out = reinterpret_cast<uint4*>(niels + r0 * GNIELS_WORDS);
    _Pragma("unroll") for (int cpc = 0; cpc < PIECES; cpc++) {
      const u32 q = (u32)cpc * 64u + lane, rec = q / PIECES, piece = q % PIECES;
      if (r0 + rec < n) out[q] = my[rec * ROW + piece];
    }
    __syncthreads();
  }
  for (u32 j = threadIdx.x; j < NH; j += MSM_CH_THREADS) { const u32 v = hist[j]; if (v) atomicAdd(&totals[j], v); }
}

// ================================================================================================ Pippenger: counting sort
// Entries of slot s live in [s n, (s + 1) n) of idx (every term contributes at most one entry per window), and
// off[s (B + 1) + j] is the first entry of bucket j of slot s (j = B: the end of the slot's entries): the windows are independent,
// no scan crosses a window.
// One pass (B <= 4096), tile by tile with the histogram of one window in LDS: block (t, s) handles terms [t tile, (t+1) tile):
//   k_msm_hist    : LDS histogram of the tile -> tcount[s][t][b]                                            (LDS atomics only)
//   k_msm_plan    : one block per slot: bucket totals over the tiles, exclusive scan -> off, and tcount[s][t][b] <- first
//                   slot of the tile's run of bucket b; block 0 also clears the fix-up counters of the pass
//   k_msm_scatter : LDS cursors start at the tile bases; every term takes the next slot of its bucket
constexpr int MSM_SORT_THREADS = 1024;
#ifndef JJ_MSM_SORT_UNROLL
#define JJ_MSM_SORT_UNROLL 4
#endif
constexpr int MSM_SORT_UNROLL = JJ_MSM_SORT_UNROLL;   // terms per thread and trip: that many loads / LDS atomics / stores in flight
__global__ void __launch_bounds__(MSM_SORT_THREADS) k_msm_hist(size_t n, size_t tile, MsmParams mp, const u32* kp, u32* tcount) {
  extern __shared__ u32 msm_lds[];
  const int w = msm_slot_window(mp, (int)blockIdx.y);
  for (u32 b = threadIdx.x; b < mp.B; b += MSM_SORT_THREADS) msm_lds[b] = 0;
  __syncthreads();
  const size_t lo = (size_t)blockIdx.x * tile, hi = lo + tile < n ? lo + tile : n;
  for (size_t i0 = lo + threadIdx.x; i0 < hi; i0 += MSM_SORT_UNROLL * MSM_SORT_THREADS) {
    u32 a[MSM_SORT_UNROLL];
    _Pragma("unroll") for (int q = 0; q < MSM_SORT_UNROLL; q++) {
      const size_t i = i0 + (size_t)q * MSM_SORT_THREADS;
      u32 neg; a[q] = i < hi ? msm_digit_wm(kp, n, i, mp, w, neg) : 0u;
    }
    _Pragma("unroll") for (int q = 0; q < MSM_SORT_UNROLL; q++) if (a[q]) atomicAdd(&msm_lds[a[q] - 1], 1u);
  }
  __syncthreads();
  u32* out = tcount + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * mp.B;
  for (u32 b = threadIdx.x; b < mp.B; b += MSM_SORT_THREADS) out[b] = msm_lds[b];
}
// exclusive scan of one value per thread over a 1024-thread workgroup (wave shuffles, then the 16 wave totals); returns the
// thread's prefix, *total = the workgroup sum.  `part` is 17 words of LDS.
static JJ_DEV u32 block_exclusive_scan_1024(u32 v, u32* part, u32* total) {
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  u32 inc = v;
  _Pragma("unroll") for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
  if (lane == 63) part[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) { u32 run = 0; for (int k = 0; k < 16; k++) { const u32 c = part[k]; part[k] = run; run += c; } part[16] = run; }
  __syncthreads();
  const u32 r = part[wave] + inc - v;
  *total = part[16];
  __syncthreads();
  return r;
}
constexpr int MSM_PLAN_PER = 4;      // at most this many buckets per thread of the plan kernel: B <= 4096
__global__ void __launch_bounds__(1024) k_msm_plan(size_t n, u32 B, u32 ntiles, u32* tcount, u32* off, u32* counters) {
  __shared__ u32 part[17];
  const u32 s = blockIdx.x;
  if (s == 0 && threadIdx.x < MSM_COUNTER_WORDS) counters[threadIdx.x] = 0;
  u32* tc = tcount + (size_t)s * ntiles * B;
  const u32 per = (B + 1023u) / 1024u;                       // consecutive buckets per thread: all 1024 threads work when B >= 1024
  const u32 b0 = threadIdx.x * per;
  u32 tot[MSM_PLAN_PER], sum = 0;
  _Pragma("unroll") for (int j = 0; j < MSM_PLAN_PER; j++) {
    tot[j] = 0;
    if ((u32)j < per && b0 + j < B) { _Pragma("unroll 8") for (u32 t = 0; t < ntiles; t++) tot[j] += tc[(size_t)t * B + b0 + j]; }     // independent loads in flight together
    sum += tot[j];
  }
  u32 total;
  u32 run = (u32)(s * n) + block_exclusive_scan_1024(sum, part, &total);
  u32* o = off + (size_t)s * (B + 1);
  _Pragma("unroll") for (int j = 0; j < MSM_PLAN_PER; j++) {
    if ((u32)j >= per || b0 + j >= B) break;
    o[b0 + j] = run;
    u32 r2 = run;
    for (u32 t0 = 0; t0 < ntiles; t0 += 8) {                  // eight counts read together, then replaced by their first slots
      u32 cc[8];
      _Pragma("unroll") for (u32 q = 0; q < 8; q++) cc[q] = t0 + q < ntiles ? tc[(size_t)(t0 + q) * B + b0 + j] : 0u;
      _Pragma("unroll") for (u32 q = 0; q < 8; q++) if (t0 + q < ntiles) { tc[(size_t)(t0 + q) * B + b0 + j] = r2; r2 += cc[q]; }
    }
    run += tot[j];
  }
  if (threadIdx.x == 0) o[B] = (u32)(s * n) + total;
}
// idx[slot] = term | sign<<31
// XCD-aware block mapping: workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  All tiles of one window
// write into that window's 4-byte index segment, so a window is given to ONE XCD (slot = 8 * (j / ntiles) + xcd): the
// partial-line writes of its tiles then meet in the same L2 and leave it as full lines.
__global__ void __launch_bounds__(MSM_SORT_THREADS) k_msm_scatter(size_t n, size_t tile, u32 ntiles, MsmParams mp, const u32* kp, const u32* tbase, u32* idx) {
  extern __shared__ u32 msm_lds[];
  const u32 xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
  const int s = (int)((j / ntiles) * 8 + xcd);
  const u32 tile_id = j % ntiles;
  if (s >= mp.Ws) return;
  const int w = msm_slot_window(mp, s);
  const u32* base = tbase + ((size_t)s * ntiles + tile_id) * mp.B;
  for (u32 b = threadIdx.x; b < mp.B; b += MSM_SORT_THREADS) msm_lds[b] = base[b];
  __syncthreads();
  const size_t lo = (size_t)tile_id * tile, hi = lo + tile < n ? lo + tile : n;
  // MSM_SORT_UNROLL terms per trip: the digit loads, then the LDS cursor updates, then the stores (more memory operations in flight)
  for (size_t i0 = lo + threadIdx.x; i0 < hi; i0 += MSM_SORT_UNROLL * MSM_SORT_THREADS) {
    u32 a[MSM_SORT_UNROLL], neg[MSM_SORT_UNROLL];
    _Pragma("unroll") for (int q = 0; q < MSM_SORT_UNROLL; q++) {
      const size_t i = i0 + (size_t)q * MSM_SORT_THREADS;
      a[q] = 0; neg[q] = 0;
      if (i < hi) a[q] = msm_digit_wm(kp, n, i, mp, w, neg[q]);
    }
    u32 slot[MSM_SORT_UNROLL];
    _Pragma("unroll") for (int q = 0; q < MSM_SORT_UNROLL; q++) slot[q] = a[q] ? atomicAdd(&msm_lds[a[q] - 1], 1u) : 0u;
    _Pragma("unroll") for (int q = 0; q < MSM_SORT_UNROLL; q++) if (a[q]) idx[slot[q]] = (u32)(i0 + (size_t)q * MSM_SORT_THREADS) | (neg[q] << 31);
  }
}
// ---- One-pass sort in TWO launches (round 6; 2^14 .. 147 456 terms, windows of 11 bits).  Conversion, histogram, plan and scatter were four
// launches of 6-30 us each, shorter than the host's launch interval: up to 35-40 us of the 2^17-term call were an idle GPU between them.
// LDS atomics run at about ONE LANE PER CLOCK PER CU (k_msm_hist: 3.0 M in 4.5 us on 256 CUs; a first fused kernel that counted on 64 CUs took
// 18 us), so the counting keeps the decomposition of k_msm_hist -- block (part, slot), every CU counts -- and what goes is the pass that
// produced its input: the blocks read the RAW scalars and recode them on the fly (two 16-byte loads and an eight-word carry chain per term, 23
// times over, out of L2: the 4 MB of scalars are read once per XCD from memory), k' is never written.
//   k_msm_front2   : workgroups [0, nparts Ws): block (part, slot) counts the part's entries per bucket of the slot's window in LDS ->
//                    tc[slot][part][bucket]; the workgroups after them convert 1024 points each to affine-Niels records (per-wave staging as
//                    k_msm_convert, half a wave at a time).  Two workgroups fit a CU: counting (LDS-atomic-bound) and conversion (VALU) overlap.
//   k_msm_scatter2 : block (part, slot) forms what k_msm_plan left in memory itself: every thread sums its bucket's counts over the parts
//                    (nparts coalesced rows) -> bucket totals and the part's prefix, one workgroup scan -> off (written by part 0), LDS
//                    cursors, then the scatter of k_msm_scatter over the part's terms, digits from the raw scalars again.  The runs of a
//                    bucket are in part order.
constexpr int MSM_F2_THREADS = 512, MSM_F2_PARTS_MAX = 64;         // 512: counting and conversion workgroups both spread over all CUs (2^17 terms: 253 + 256 of them)
constexpr int MSM_F2_STAGE_WORDS = (MSM_F2_THREADS / 64) * 32 * (GNIELS_WORDS / 4 + 1) * 4;      // 32 entries per wave at a time: 36 864 bytes
// digit of window w (wave-uniform) of k' = k + recode, k read from the caller's scalar array
static JJ_DEV u32 msm_digit_scalar(const void* scalars, size_t i, const MsmParams& mp, int w, u32& neg) {
  u32 k[8];
  load8(k, scalars, i);
  msm_recode(k, mp);
  const int bit = msm_win_start(mp, w), width = msm_win_width(mp, w), wi = bit >> 5, sh = bit & 31;
  u32 lo, hi;
  switch (wi) {                                     // uniform over the workgroup
    case 0: lo = k[0]; hi = k[1]; break;
    case 1: lo = k[1]; hi = k[2]; break;
    case 2: lo = k[2]; hi = k[3]; break;
    case 3: lo = k[3]; hi = k[4]; break;
    case 4: lo = k[4]; hi = k[5]; break;
    case 5: lo = k[5]; hi = k[6]; break;
    case 6: lo = k[6]; hi = k[7]; break;
    default: lo = k[7]; hi = 0; break;
  }
  return msm_digit_raw(__builtin_amdgcn_alignbit(hi, lo, (u32)sh) & ((1u << width) - 1u), mp, w, width, neg);
}
__global__ void __launch_bounds__(MSM_F2_THREADS) k_msm_front2(size_t n, const void* scalars, const void* points, MsmParams mp, u32* niels, u32* tc /* [Ws][nparts][B] */,
                                                               u32* counters, u32 nparts, size_t part_terms) {
  constexpr int PIECES = GNIELS_WORDS / 4, ROW = PIECES + 1;
  extern __shared__ __attribute__((aligned(16))) u32 f2_lds[];
  if (blockIdx.x == 0 && threadIdx.x < MSM_COUNTER_WORDS) counters[threadIdx.x] = 0;                 // (k_msm_plan's job on the four-launch path)
  const u32 nhist = nparts * (u32)mp.Ws;
  if (blockIdx.x >= nhist) {
    // ---- points: 64 entries per wave, staged and stored 32 at a time (the staging area is the wave's own: no workgroup barrier)
    const u32 lane = threadIdx.x & 63u;
    uint4* my = reinterpret_cast<uint4*>(f2_lds) + (size_t)(threadIdx.x >> 6) * 32 * ROW;
    const size_t i = (size_t)(blockIdx.x - nhist) * MSM_F2_THREADS + threadIdx.x;
    u32 wv[GNIELS_WORDS];
    if (i < n) {
      const ANiels t = Curve::to_niels(load_affine(points, i));
      _Pragma("unroll") for (int l = 0; l < NL; l++) { wv[l] = t.vpu.l[l]; wv[NL + l] = t.vmu.l[l]; wv[2 * NL + l] = t.t2d.l[l]; }
      _Pragma("unroll") for (int l = ANIELS_WORDS - 1; l < GNIELS_WORDS; l++) wv[l] = 0;
    }
    const size_t r0 = (size_t)(blockIdx.x - nhist) * MSM_F2_THREADS + (threadIdx.x & ~63u);      // first entry of this wave
    _Pragma("unroll") for (u32 h = 0; h < 2; h++) {
      if ((lane >> 5) == h && i < n) { _Pragma("unroll") for (int v = 0; v < PIECES; v++) my[(lane & 31u) * ROW + v] = make_uint4(wv[4 * v], wv[4 * v + 1], wv[4 * v + 2], wv[4 * v + 3]); }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      uint4* out = reinterpret_cast<uint4*>(niels + (r0 + 32 * h) * GNIELS_WORDS);
      _Pragma("unroll") for (int cpc = 0; cpc < PIECES / 2; cpc++) {
        const u32 q = (u32)cpc * 64u + lane, rec = q / PIECES, piece = q % PIECES;
        if (r0 + 32 * h + rec < n) out[q] = my[rec * ROW + piece];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    return;
  }
  // ---- counting: block (part, slot)
  const u32 s = blockIdx.x / nparts, part = blockIdx.x - s * nparts;
  const int w = msm_slot_window(mp, (int)s);
  for (u32 b = threadIdx.x; b < mp.B; b += MSM_F2_THREADS) f2_lds[b] = 0;
  __syncthreads();
  const size_t lo = (size_t)part * part_terms, hi = lo + part_terms < n ? lo + part_terms : n;
  for (size_t i0 = lo + threadIdx.x; i0 < hi; i0 += MSM_SORT_UNROLL * MSM_F2_THREADS) {
    u32 a[MSM_SORT_UNROLL];
    _Pragma("unroll") for (int q = 0; q < MSM_SORT_UNROLL; q++) {
      const size_t i = i0 + (size_t)q * MSM_F2_THREADS;
      u32 neg; a[q] = i < hi ? msm_digit_scalar(scalars, i, mp, w, neg) : 0u;
    }
    _Pragma("unroll") for (int q = 0; q < MSM_SORT_UNROLL; q++) if (a[q]) atomicAdd(&f2_lds[a[q] - 1], 1u);
  }
  __syncthreads();
  u32* out = tc + ((size_t)s * nparts + part) * mp.B;
  for (u32 b = threadIdx.x; b < mp.B; b += MSM_F2_THREADS) out[b] = f2_lds[b];
}
// (Staging the part's entries through LDS and copying them out run by run, as k_msm_part_scatter does, was measured and dropped: 29-36 us
// against 31 for the direct stores below -- with ~11 entries per run the copy-out saves few transactions and the block pays a second scan
// and two more barriers.)
__global__ void __launch_bounds__(MSM_SORT_THREADS) k_msm_scatter2(size_t n, u32 nparts, size_t part_terms, MsmParams mp, const void* scalars, const u32* tc, u32* off, u32* idx) {
  extern __shared__ u32 msm_lds[];
  __shared__ u32 part_s[17];
  const u32 xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
  const int s = (int)((j / nparts) * 8 + xcd);                 // a window's parts on ONE XCD (as k_msm_scatter)
  const u32 part = j % nparts;
  if (s >= mp.Ws) return;
  const int w = msm_slot_window(mp, s);
  const u32 B = mp.B;
  const u32 bper = (B + 1023u) / 1024u, b0 = threadIdx.x * bper;             // consecutive buckets per thread (B <= 4096)
  const u32* rows = tc + (size_t)s * nparts * B;
  u32 tot[MSM_PLAN_PER], pre[MSM_PLAN_PER], sum = 0;
  _Pragma("unroll") for (int q = 0; q < MSM_PLAN_PER; q++) {
    tot[q] = 0; pre[q] = 0;
    if ((u32)q < bper && b0 + q < B) {
      _Pragma("unroll 8") for (u32 t = 0; t < nparts; t++) { const u32 c = rows[(size_t)t * B + b0 + q]; tot[q] += c; pre[q] += t < part ? c : 0u; }
    }
    sum += tot[q];
  }
  u32 total;
  u32 run = (u32)((size_t)s * n) + block_exclusive_scan_1024(sum, part_s, &total);
  u32* o = off + (size_t)s * (B + 1);
  _Pragma("unroll") for (int q = 0; q < MSM_PLAN_PER; q++) {
    if ((u32)q >= bper || b0 + q >= B) break;
    if (part == 0) o[b0 + q] = run;
    msm_lds[b0 + q] = run + pre[q];
    run += tot[q];
  }
  if (part == 0 && threadIdx.x == 0) o[B] = (u32)((size_t)s * n) + total;
  __syncthreads();
  const size_t lo = (size_t)part * part_terms, hi = lo + part_terms < n ? lo + part_terms : n;
  for (size_t i0 = lo + threadIdx.x; i0 < hi; i0 += MSM_SORT_UNROLL * MSM_SORT_THREADS) {
    u32 a[MSM_SORT_UNROLL], neg[MSM_SORT_UNROLL];
    _Pragma("unroll") for (int q = 0; q < MSM_SORT_UNROLL; q++) {
      const size_t i = i0 + (size_t)q * MSM_SORT_THREADS;
      a[q] = 0; neg[q] = 0;
      if (i < hi) a[q] = msm_digit_scalar(scalars, i, mp, w, neg[q]);
    }
    u32 slot[MSM_SORT_UNROLL];
    _Pragma("unroll") for (int q = 0; q < MSM_SORT_UNROLL; q++) slot[q] = a[q] ? atomicAdd(&msm_lds[a[q] - 1], 1u) : 0u;
    _Pragma("unroll") for (int q = 0; q < MSM_SORT_UNROLL; q++) if (a[q]) idx[slot[q]] = (u32)(i0 + (size_t)q * MSM_SORT_THREADS) | (neg[q] << 31);
  }
}
// Two-pass sort for wide windows (>= 4096 buckets per window).  The single-pass scatter above ends in one 4-byte store per
// entry into an index segment shared by all tiles of the window: every store opens its own cache line and most lines leave L2
// partially written.  Here the bucket index is split into a coarse bin (bits 8 and up, <= 128 bins per window) and its low 8 bits:
//   k_msm_part_hist    : block (tile, slot): entries per coarse bin -> tc[slot][bin][tile]
//   k_msm_part_plan    : one block per slot: exclusive scan over (bin, tile) -> first entry of every run; clears the counters
//   k_msm_part_scatter : block (tile, slot) writes (term | sign << 31) and the low 8 bits into its run of every bin: 128 open
//                        lines per block, each filled front to back by one CU
//   k_msm_part_sort    : block (bin, slot): counts the 256 low values, writes the bucket offsets of its bin (the offsets of
//                        the whole sort: no scan over the buckets), orders the bin in LDS and copies it out coalesced
// A bin larger than the LDS stage (skewed scalars) is scattered directly; its stores still stay within the one block.
constexpr int MSM_LO_BITS = 8;
constexpr u32 MSM_HB_MAX = 128;        // coarse bins per window the LDS arrays below are sized for (windows of at most 16 bits)
#ifndef JJ_MSM_P2_THREADS
#define JJ_MSM_P2_THREADS 512
#endif
constexpr int MSM_P2_THREADS = JJ_MSM_P2_THREADS;
constexpr u32 MSM_P2_CAP = 12288;     // entries staged in LDS (48 KB): 1.5 x the mean bin of a 2^20-term, 16-bit-window pass
__global__ void __launch_bounds__(MSM_SORT_THREADS) k_msm_part_hist(size_t n, size_t tile, MsmParams mp, const u32* kp, u32* tc) {
  __shared__ u32 h[(MSM_SORT_THREADS / 64) * MSM_HB_MAX];          // one histogram per wave: fewer same-address collisions
  const int w = msm_slot_window(mp, (int)blockIdx.y);
  const u32 HB = mp.B >> MSM_LO_BITS, wave = threadIdx.x >> 6;
  for (u32 b = threadIdx.x; b < (MSM_SORT_THREADS / 64) * HB; b += MSM_SORT_THREADS) h[b] = 0;
  __syncthreads();
  const size_t lo = (size_t)blockIdx.x * tile, hi = lo + tile < n ? lo + tile : n;
  for (size_t i0 = lo + threadIdx.x; i0 < hi; i0 += MSM_SORT_UNROLL * MSM_SORT_THREADS) {
    u32 a[MSM_SORT_UNROLL];
    _Pragma("unroll") for (int q = 0; q < MSM_SORT_UNROLL; q++) {
      const size_t i = i0 + (size_t)q * MSM_SORT_THREADS;
      u32 neg; a[q] = i < hi ? msm_digit_wm(kp, n, i, mp, w, neg) : 0u;
    }
    _Pragma("unroll") for (int q = 0; q < MSM_SORT_UNROLL; q++) if (a[q]) atomicAdd(&h[wave * HB + ((a[q] - 1) >> MSM_LO_BITS)], 1u);
  }
  __syncthreads();
  for (u32 b = threadIdx.x; b < HB; b += MSM_SORT_THREADS) {
    u32 sm = 0;
    for (u32 v = 0; v < MSM_SORT_THREADS / 64; v++) sm += h[v * HB + b];
    tc[((size_t)blockIdx.y * HB + b) * gridDim.x + blockIdx.x] = sm;
  }
}
// tcs[s (m + 1) + k], m = HB ptiles: first entry of run k = bin * ptiles + tile of slot s; k = m: the end of the slot's entries
__global__ void __launch_bounds__(1024) k_msm_part_plan(size_t n, u32 m, const u32* tc, u32* tcs, u32* counters) {
  __shared__ u32 part[17];
  const u32 s = blockIdx.x;
  if (s == 0 && threadIdx.x < MSM_COUNTER_WORDS) counters[threadIdx.x] = 0;
  const u32* in = tc + (size_t)s * m;
  u32* out = tcs + (size_t)s * (m + 1);
  // a thread owns `per` consecutive counts, a multiple of 4 read as 16-byte words (m = HB ptiles is a multiple of 16 and tc is 16-byte
  // aligned): the loads of a thread are independent of each other, where a scalar loop waited for each one (18.7 -> 6 us at 2^20 terms)
  const u32 per = ((m + 1023) / 1024 + 3u) & ~3u, lo = threadIdx.x * per, hi = lo + per < m ? lo + per : m;
  u32 sum = 0;
  if (lo < hi) {
    _Pragma("unroll 4") for (u32 k = lo; k < hi; k += 4) { const uint4 q = *reinterpret_cast<const uint4*>(in + k); sum += q.x + q.y + q.z + q.w; }
  }
  u32 total;
  u32 run = (u32)(s * n) + block_exclusive_scan_1024(sum, part, &total);
  if (lo < hi) {
    _Pragma("unroll 4") for (u32 k = lo; k < hi; k += 4) {
      const uint4 q = *reinterpret_cast<const uint4*>(in + k);
      out[k] = run; out[k + 1] = run + q.x; out[k + 2] = run + q.x + q.y; out[k + 3] = run + q.x + q.y + q.z;
      run += q.x + q.y + q.z + q.w;
    }
  }
  if (threadIdx.x == 0) out[m] = (u32)(s * n) + total;
}
// tile of at most MSM_P1_TILE terms: ranks from one LDS atomic per entry, the tile ordered by bin in LDS, then copied out run by run
// (consecutive stage slots of one bin are consecutive in rec / lo8: a wave's store touches a few lines instead of 64)
constexpr u32 MSM_P1_TILE = 8192;
constexpr int MSM_P1_PER = MSM_P1_TILE / MSM_SORT_THREADS;
// tcs == nullptr (round 5, after k_msm_convert_hist): no per-tile offsets exist; the block forms the bins' first entries from the TOTALS per (slot, bin)
// itself (a scan over at most 128 values) and reserves its run in every bin with one global atomic on the bin's cursor.  The order of the tiles' runs
// inside a bin then depends on the order the blocks arrive in -- the entries of a bucket are added in another order, the sum is the same point.
__global__ void __launch_bounds__(MSM_SORT_THREADS) k_msm_part_scatter(size_t n, size_t tile, MsmParams mp, const u32* kp, const u32* tcs, u32* rec, uint8_t* lo8, const u32* totals, u32* cursor) {
  __shared__ u32 cnt[MSM_HB_MAX], live_s;
  __shared__ u32 st_rec[MSM_P1_TILE];
  __shared__ uint8_t st_lo[MSM_P1_TILE], st_bin[MSM_P1_TILE];
  const int w = msm_slot_window(mp, (int)blockIdx.y);
  const u32 HB = mp.B >> MSM_LO_BITS, tid = threadIdx.x;
  const u32* run0 = tcs + (size_t)blockIdx.y * ((size_t)HB * gridDim.x + 1);
  __shared__ u32 gbase[MSM_HB_MAX], loff[MSM_HB_MAX];      // first global slot of this tile's run of a bin | first stage slot of the bin
  if (tid < 128) cnt[tid] = 0;
  if (!tcs && tid >= 64 && tid < 128) {                     // wave 1: the bins' first entries from the totals, while the digits are being loaded
    const u32 j = tid - 64;
    const u32* tot = totals + (size_t)blockIdx.y * HB;
    const u32 t0 = 2 * j < HB ? tot[2 * j] : 0u, t1 = 2 * j + 1 < HB ? tot[2 * j + 1] : 0u;
    u32 tinc = t0 + t1;
    _Pragma("unroll") for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(tinc, d, 64); if ((int)j >= d) tinc += o; }
    const u32 b0 = (u32)(blockIdx.y * n) + tinc - (t0 + t1);
    gbase[2 * j] = b0; gbase[2 * j + 1] = b0 + t0;
  }
  __syncthreads();
  const size_t lo = (size_t)blockIdx.x * tile, hi = lo + tile < n ? lo + tile : n;
  u32 a[MSM_P1_PER], neg[MSM_P1_PER], rank[MSM_P1_PER];
  _Pragma("unroll") for (int q = 0; q < MSM_P1_PER; q++) {
    const size_t i = lo + tid + (size_t)q * MSM_SORT_THREADS;
    a[q] = 0; neg[q] = 0;
    if (i < hi) a[q] = msm_digit_wm(kp, n, i, mp, w, neg[q]);
  }
  _Pragma("unroll") for (int q = 0; q < MSM_P1_PER; q++) rank[q] = a[q] ? atomicAdd(&cnt[(a[q] - 1) >> MSM_LO_BITS], 1u) : 0u;
  __syncthreads();
  if (tid < 64) {                        // wave 0: exclusive scan of the (at most 128) bin counts, two per lane: the bins' first stage slots
    const u32 v0 = cnt[2 * tid], v1 = cnt[2 * tid + 1], sm = v0 + v1;
    u32 inc = sm;
    _Pragma("unroll") for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d, 64); if ((int)tid >= d) inc += o; }
    const u32 l0 = inc - sm;
    loff[2 * tid] = l0; loff[2 * tid + 1] = l0 + v0;
    if (tcs) {
      gbase[2 * tid] = 2 * tid < HB ? run0[(size_t)(2 * tid) * gridDim.x + blockIdx.x] : 0u;
      gbase[2 * tid + 1] = 2 * tid + 1 < HB ? run0[(size_t)(2 * tid + 1) * gridDim.x + blockIdx.x] : 0u;
    }
    if (tid == 63) live_s = inc;         // entries of the tile (terms with a nonzero digit)
  } else if (!tcs && tid < 128) {        // wave 1, beside it: this tile's run in every bin, reserved with one global atomic per bin
    const u32 j = tid - 64;
    u32* cur = cursor + (size_t)blockIdx.y * HB;
    const u32 v0 = cnt[2 * j], v1 = cnt[2 * j + 1];
    const u32 g0 = v0 ? atomicAdd(&cur[2 * j], v0) : 0u, g1 = v1 ? atomicAdd(&cur[2 * j + 1], v1) : 0u;
    gbase[2 * j] += g0; gbase[2 * j + 1] += g1;
  }
  __syncthreads();
  _Pragma("unroll") for (int q = 0; q < MSM_P1_PER; q++) if (a[q]) {
    const u32 bin = (a[q] - 1) >> MSM_LO_BITS, slot = loff[bin] + rank[q];
    st_rec[slot] = (u32)(lo + tid + (size_t)q * MSM_SORT_THREADS) | (neg[q] << 31);
    st_lo[slot] = (uint8_t)((a[q] - 1) & ((1u << MSM_LO_BITS) - 1u));
    st_bin[slot] = (uint8_t)bin;
  }
  __syncthreads();
  const u32 live = live_s;
  for (u32 j = tid; j < live; j += MSM_SORT_THREADS) {
    const u32 bn = st_bin[j], g = j - loff[bn] + gbase[bn];
    rec[g] = st_rec[j];
    lo8[g] = st_lo[j];
  }
}
// SEGH: the block also does what k_seg_hist would do for its 256 buckets (it holds their counts anyway): the histogram of segment lengths of
// tile s HB + coarse, key-major in bh, and the identity for empty buckets -- one launch and one pass over the offsets fewer (round 5).
constexpr int SEG_PMAX = 1024;
template <bool SEGH>
__global__ void __launch_bounds__(MSM_P2_THREADS) k_msm_part_sort(MsmParams mp, u32 ptiles, const u32* tcs, const u32* rec, const uint8_t* lo8, u32* idx, u32* off, u32 P, ExtAoS buckets, u32* bh, const u32* totals) {
  constexpr u32 NLO = 1u << MSM_LO_BITS;
  constexpr int PER = MSM_P2_CAP / MSM_P2_THREADS;
  __shared__ u32 cnt[NLO];
  __shared__ u32 stage[MSM_P2_CAP];
  __shared__ u32 seg_h[SEGH ? SEG_PMAX + 1 : 1];
  const u32 coarse = blockIdx.x, HB = gridDim.x, s = blockIdx.y, tid = threadIdx.x;
  if constexpr (SEGH) { for (u32 k = tid; k <= P; k += MSM_P2_THREADS) seg_h[k] = 0; }
  u32 gb, ge;
  if (tcs) {
    const u32* run0 = tcs + (size_t)s * ((size_t)HB * ptiles + 1);
    gb = run0[(size_t)coarse * ptiles]; ge = run0[(size_t)(coarse + 1) * ptiles];
  } else {
    // the bin's entries from the totals per (slot, bin): its first entry = the slot's base + the totals of the bins before it (ptiles = terms of the pass here)
    __shared__ u32 range_s[2];
    if (tid < 64) {
      const u32* tot = totals + (size_t)s * HB;
      const u32 t0 = 2 * tid < HB ? tot[2 * tid] : 0u, t1 = 2 * tid + 1 < HB ? tot[2 * tid + 1] : 0u;
      u32 tinc = t0 + t1;
      _Pragma("unroll") for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(tinc, d, 64); if ((int)tid >= d) tinc += o; }
      const u32 b0 = s * ptiles + tinc - (t0 + t1);
      if (2 * tid == coarse) { range_s[0] = b0; range_s[1] = b0 + t0; }
      if (2 * tid + 1 == coarse) { range_s[0] = b0 + t0; range_s[1] = b0 + t0 + t1; }
    }
    __syncthreads();
    gb = range_s[0]; ge = range_s[1];
  }
  const bool staged = ge - gb <= MSM_P2_CAP;
  u32* o = off + (size_t)s * (mp.B + 1);
  if (tid < NLO) cnt[tid] = 0;
  __syncthreads();
  // the usual bin fits the stage: every thread takes its PER entries in one round of loads, keeps them in registers and draws
  // their ranks within the low value from the counting atomics (one LDS atomic per entry)
  u32 r[PER], b[PER], rank[PER];
  if (staged) {
    _Pragma("unroll") for (int q = 0; q < PER; q++) {
      const u32 i = gb + tid + (u32)q * MSM_P2_THREADS;
      b[q] = i < ge ? (u32)lo8[i] : ~0u;
      r[q] = i < ge ? rec[i] : 0u;
    }
    _Pragma("unroll") for (int q = 0; q < PER; q++) rank[q] = b[q] != ~0u ? atomicAdd(&cnt[b[q]], 1u) : 0u;
  } else {
    // oversized bin (skewed scalars): the same rounds of PER loads per thread, stage by stage
    for (u32 base = gb; base < ge; base += MSM_P2_CAP) {
      _Pragma("unroll") for (int q = 0; q < PER; q++) { const u32 i = base + tid + (u32)q * MSM_P2_THREADS; b[q] = i < ge ? (u32)lo8[i] : ~0u; }
      _Pragma("unroll") for (int q = 0; q < PER; q++) if (b[q] != ~0u) atomicAdd(&cnt[b[q]], 1u);
    }
  }
  __syncthreads();
  // exclusive scan of the 256 counters by the first wave: four per lane, then a shuffle scan over the lane sums
  if (tid < 64) {
    u32 v[NLO / 64], sm = 0;
    _Pragma("unroll") for (u32 j = 0; j < NLO / 64; j++) { v[j] = cnt[tid * (NLO / 64) + j]; sm += v[j]; }
    u32 inc = sm;
    _Pragma("unroll") for (int d = 1; d < 64; d <<= 1) { const u32 x = __shfl_up(inc, d, 64); if ((int)tid >= d) inc += x; }
    u32 run = inc - sm;
    _Pragma("unroll") for (u32 j = 0; j < NLO / 64; j++) {
      const u32 k = tid * (NLO / 64) + j;
      cnt[k] = run;                                                    // first slot of the low value, relative to gb
      o[((size_t)coarse << MSM_LO_BITS) + k] = gb + run;
      run += v[j];
    }
    if (coarse + 1 == HB && tid == 0) o[mp.B] = ge;
  }
  __syncthreads();
  if constexpr (SEGH) {
    if (tid < NLO) {                                                    // one bucket per thread: its count from the scanned offsets
      const u32 c = (tid + 1 < NLO ? cnt[tid + 1] : ge - gb) - cnt[tid];
      if (c == 0) aos_put_ext(buckets, (size_t)s * mp.B + ((size_t)coarse << MSM_LO_BITS) + tid, Curve::identity());
      else {
        const u32 full = c / P, rem = c - full * P;
        if (full) atomicAdd(&seg_h[0], full);
        if (rem) atomicAdd(&seg_h[P - rem], 1u);
      }
    }
    __syncthreads();
    const size_t stiles = (size_t)gridDim.x * gridDim.y, tile = (size_t)s * HB + coarse;
    for (u32 k = tid; k <= P; k += MSM_P2_THREADS) bh[(size_t)k * stiles + tile] = seg_h[k];
  }
  if (staged) {
    _Pragma("unroll") for (int q = 0; q < PER; q++) if (b[q] != ~0u) stage[cnt[b[q]] + rank[q]] = r[q];
    __syncthreads();
    for (u32 j = tid; j < ge - gb; j += MSM_P2_THREADS) idx[gb + j] = stage[j];
  } else {
    for (u32 base = gb; base < ge; base += MSM_P2_CAP) {
      _Pragma("unroll") for (int q = 0; q < PER; q++) {
        const u32 i = base + tid + (u32)q * MSM_P2_THREADS;
        b[q] = i < ge ? (u32)lo8[i] : ~0u;
        r[q] = i < ge ? rec[i] : 0u;
      }
      _Pragma("unroll") for (int q = 0; q < PER; q++) rank[q] = b[q] != ~0u ? atomicAdd(&cnt[b[q]], 1u) : 0u;
      _Pragma("unroll") for (int q = 0; q < PER; q++) if (b[q] != ~0u) idx[gb + rank[q]] = r[q];
    }
  }
}

// ================================================================================================ Pippenger: bucket accumulation
// Balanced accumulation over fixed chunks: the sorted entries of a slot are cut into chunks of `chunk` entries, one lane per
// chunk, so every lane performs the same number of mixed additions whatever the bucket-size distribution.  A run that starts at a
// bucket start is written to buckets[s B + b]; the run a chunk inherits from the previous chunk goes to head[s nchunk + t] and is
// merged by k_msm_fixup.
constexpr int MSM_CHUNK_MIN = 16;
// OFF_LDS (round 6; windows of at most MSM_ACC_LDS_BUCKETS buckets, i.e. every layout the chunked path is planned for): the slot's B + 1 bucket
// offsets are staged in LDS first.  The lane's first bucket is then found by a binary search over LDS (ten dependent reads of ~0.1 us instead of
// ten dependent global loads of ~0.8 us at the head of every lane's chain), and a lane that crosses into the next bucket -- some lane of a wave
// does in four iterations of ten at 128 entries per bucket -- no longer stalls its wave on a global load in the middle of the addition chain.
constexpr u32 MSM_ACC_LDS_BUCKETS = 8192;
template <bool OFF_LDS>
__global__ void __launch_bounds__(256) k_msm_accumulate(size_t n, u32 B, u32 chunk, u32 nchunk, const u32* off, const u32* idx, const u32* niels, ExtAoS buckets, ExtAoS head) {
  extern __shared__ __attribute__((aligned(16))) u32 acc_off[];
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 s = blockIdx.y;
  const u32* og = off + (size_t)s * (B + 1);
  if constexpr (OFF_LDS) {
    for (u32 k = threadIdx.x; k <= B; k += 256) acc_off[k] = og[k];
    __syncthreads();
  }
  auto o = [&](size_t k) -> u32 { if constexpr (OFF_LDS) return acc_off[k]; else return og[k]; };
  const size_t M = o(B);                                    // end of the slot's entries
  const size_t start = (size_t)s * n + t * chunk;
  if (t >= nchunk || start >= M) return;
  const size_t end = start + chunk < M ? start + chunk : M;
  // bucket containing `start`: largest b with o[b] <= start
  size_t lo = 0, hi = B;
  while (hi - lo > 1) { const size_t mid = (lo + hi) >> 1; if (o(mid) <= start) lo = mid; else hi = mid; }
  size_t b = lo;
  u32 nxt = o(b + 1);
  while (nxt <= start) { b++; nxt = o(b + 1); }          // skip empty buckets that share the offset
  bool inherited = o(b) < start;                            // first run continues a bucket begun in an earlier chunk
  Ext acc = Curve::identity();
  bool any = false;
  const size_t bk0 = (size_t)s * B, hd = (size_t)s * nchunk + t;
  // software pipeline (as in k_msm_accumulate_seg): the entry of term pos + 1 -- a dependent 4-byte index load, then a random
  // 128-byte gather -- is in flight while term pos is added
  u32 e = idx[start];
  ANiels p = lds_aniels(niels + (size_t)(e & 0x7fffffffu) * GNIELS_WORDS);
  u32 e_next = start + 1 < end ? idx[start + 1] : e;
  #pragma unroll 1
  for (size_t pos = start; pos < end; pos++) {
    if (pos >= nxt) {
      if (inherited) { aos_put_ext(head, hd, acc); inherited = false; } else if (any) aos_put_ext(buckets, bk0 + b, acc);
      acc = Curve::identity(); any = false;
      do { b++; nxt = o(b + 1); } while (nxt <= pos);
    }
    const ANiels p_next = lds_aniels(niels + (size_t)(e_next & 0x7fffffffu) * GNIELS_WORDS);
    const u32 e_next2 = pos + 2 < end ? idx[pos + 2] : e_next;
    acc = Curve::add_signed<true>(acc, p, (e >> 31) ? ~0u : 0u);
    any = true;
    e = e_next; p = p_next; e_next = e_next2;
  }
  if (inherited) aos_put_ext(head, hd, acc); else aos_put_ext(buckets, bk0 + b, acc);
}
// buckets[b] (+)= heads of the chunks that continue bucket b; empty buckets become the identity.
// A bucket with more than FIXUP_SERIAL_MAX heads (heavily skewed digit distribution: repeated scalars) is appended to a work
// list instead and reduced by a whole workgroup in k_msm_fixup_big; if the list is full the pair falls back to the serial loop
// (slow but correct).
constexpr u32 FIXUP_SERIAL_MAX = 32;
constexpr u32 FIXUP_BIG_MAX = 2048;       // work-list capacity
constexpr u32 FIXUP_BIG_QUADS = 64;       // quads (of 4 lanes) per big bucket
struct BigBucket { u32 bucket, t_first, t_last, pad; };
// TWO LANES per bucket, each running whole-lane additions (reference src/lib.rs:992-999) over every other head (the next head is
// fetched while the current one is added), then one more addition folds the odd lane's sum into the even lane's.  A wave on its
// own issues at nearly the SIMD's full rate (experiments/lone_wave), so what counts is instructions per wave and waves per SIMD:
// ~5 + 1 whole-lane additions of ~2100 instructions in 736 waves (2^17 terms) against ~8 four-round quad additions of ~1200 in
// 1472 waves for one quad per bucket (round 2 .. early round 3: 67 us against 52 us).
static JJ_DEV Ext pair_partner(const Ext& e) {      // the point held by the other lane of the pair (lane ^ 1)
  Ext r;
  _Pragma("unroll") for (int l = 0; l < NL; l++) {
    r.u.l[l] = (u32)__builtin_amdgcn_mov_dpp((int)e.u.l[l], 0xB1, 0xf, 0xf, false);
    r.v.l[l] = (u32)__builtin_amdgcn_mov_dpp((int)e.v.l[l], 0xB1, 0xf, 0xf, false);
    r.z.l[l] = (u32)__builtin_amdgcn_mov_dpp((int)e.z.l[l], 0xB1, 0xf, 0xf, false);
    r.t1.l[l] = (u32)__builtin_amdgcn_mov_dpp((int)e.t1.l[l], 0xB1, 0xf, 0xf, false);
    r.t2.l[l] = (u32)__builtin_amdgcn_mov_dpp((int)e.t2.l[l], 0xB1, 0xf, 0xf, false);
  }
  return r;
}
// A bucket with more than FIXUP_SERIAL_MAX heads (repeated scalars) is not walked by its pair: the WHOLE WAVE takes it afterwards -- lane l
// adds up heads t_first + l, + 64, ..., six butterfly steps over the lanes (ds_bpermute) fold the 64 partial sums, lane 0 adds the bucket's
// own run and stores.  (Until round 6 such buckets went to a work list for a second launch, k_msm_fixup_big: 5 us per call for a list that is
// empty unless scalars repeat.  The segment path of large inputs still uses that kernel for its merge list.)
static JJ_DEV Ext wave_xor_partner(const Ext& e, int d) {
  Ext r;
  _Pragma("unroll") for (int l = 0; l < NL; l++) {
    r.u.l[l] = (u32)__shfl_xor((int)e.u.l[l], d, 64); r.v.l[l] = (u32)__shfl_xor((int)e.v.l[l], d, 64); r.z.l[l] = (u32)__shfl_xor((int)e.z.l[l], d, 64);
    r.t1.l[l] = (u32)__shfl_xor((int)e.t1.l[l], d, 64); r.t2.l[l] = (u32)__shfl_xor((int)e.t2.l[l], d, 64);
  }
  return r;
}
__global__ void __launch_bounds__(256) k_msm_fixup(size_t n, u32 B, u32 Ws, u32 chunk, u32 nchunk, const u32* off, ExtAoS buckets, ExtAoS head) {
  const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  const u32 half = threadIdx.x & 1u, lane = threadIdx.x & 63u;
  size_t t_first = 1, t_last = 0;                             // no heads
  if (g < (size_t)Ws * B) {
    const u32 s = (u32)(g / B), j = (u32)(g % B);
    const u32* o = off + (size_t)s * (B + 1);
    const u32 base = (u32)(s * n), lo = o[j] - base, hi = o[j + 1] - base;
    if (lo == hi) { if (half == 0) aos_put_ext(buckets, g, Curve::identity()); }
    else { t_first = (size_t)s * nchunk + lo / chunk + 1; t_last = (size_t)s * nchunk + (hi - 1) / chunk; }
  }
  const bool some = t_first <= t_last, big = some && t_last - t_first + 1 > FIXUP_SERIAL_MAX;
  if (some && !big) {
    // even lane: the bucket's own first run + heads t_first, t_first + 2, ...; odd lane: heads t_first + 1, t_first + 3, ...
    Ext acc = half == 0 ? aos_ext(buckets, g) : Curve::identity();
    size_t t = t_first + half;
    Ext nx = aos_ext(head, t <= t_last ? t : t_last);
    #pragma unroll 1
    for (; t <= t_last; t += 2) {
      const Ext cur = nx;
      if (t + 2 <= t_last) nx = aos_ext(head, t + 2);
      acc = Curve::add(acc, Curve::to_niels(cur));
    }
    acc = Curve::add(acc, Curve::to_niels(pair_partner(acc)));
    if (half == 0) aos_put_ext(buckets, g, acc);
  }
  unsigned long long todo = __ballot(big && half == 0);
  #pragma unroll 1
  while (todo) {                                              // uniform over the wave
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const size_t gb = ((size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u) + (u32)src) >> 1;
    const u32 tf_lo = (u32)__shfl((int)(u32)t_first, src, 64), tf_hi = (u32)__shfl((int)(u32)(t_first >> 32), src, 64);
    const u32 tl_lo = (u32)__shfl((int)(u32)t_last, src, 64), tl_hi = (u32)__shfl((int)(u32)(t_last >> 32), src, 64);
    const size_t tf = ((size_t)tf_hi << 32) | tf_lo, tl = ((size_t)tl_hi << 32) | tl_lo;
    Ext acc = Curve::identity();
    #pragma unroll 1
    for (size_t t = tf + lane; t <= tl; t += 64) acc = Curve::add(acc, Curve::to_niels(aos_ext(head, t)));
    #pragma unroll 1
    for (int d = 32; d >= 1; d >>= 1) acc = Curve::add(acc, Curve::to_niels(wave_xor_partner(acc, d)));
    if (lane == 0) { acc = Curve::add(acc, Curve::to_niels(aos_ext(buckets, gb))); aos_put_ext(buckets, gb, acc); }
  }
}
// ---- Segment-sorted accumulation (large inputs).  Every non-empty bucket is cut into segments of at most P entries, the
// segments are counting-sorted by length (longest first), and each lane adds up one segment: lanes of a wave run the
// same number of iterations, no lane ever switches buckets inside its loop, and a bucket with a single segment (the
// common case) is finished by its lane.  Buckets with several segments (repeated scalars) get their extra segments as `head`
// partials that k_msm_fixup_big folds in (a merge list for few, the big-bucket list for many).
#ifndef JJ_MSM_ACC_MINBLOCKS
#define JJ_MSM_ACC_MINBLOCKS 1        // resident 256-thread blocks per CU the accumulate kernel is compiled for: 1 = no register cap (137 VGPRs,
#endif                                // 3 waves per SIMD); capping at 128 (4 waves) changes nothing, 96 (5 waves) spills (profiles/r2_msm_acc_occupancy.txt)
struct Seg { u32 start, len, dst, pad; };            // dst: bucket index, or 0x80000000 | head index
struct MergeItem { u32 bucket, h0, k, pad; };         // buckets[bucket] += head[h0 .. h0 + k)
// bucket g = s B + j of the pass: its entries are [lo, lo + c)
static JJ_DEV void seg_bucket(const u32* off, u32 B, size_t g, u32& lo, u32& c) {
  const size_t s = g / B, j = g % B;
  const u32* o = off + s * (B + 1);
  lo = o[j]; c = o[j + 1] - lo;
}
// pass 1: per-tile histogram of segment lengths, key = P - len (longer first), stored key-major (bh[key * tiles + tile]: the plan
// scans one key's row); empty buckets become the identity
__global__ void __launch_bounds__(256) k_seg_hist(size_t nb, u32 B, u32 per_tile, u32 P, const u32* off, ExtAoS buckets, u32* bh) {
  __shared__ u32 hist[SEG_PMAX + 1];
  for (u32 k = threadIdx.x; k <= P; k += 256) hist[k] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * per_tile;
  for (u32 j = threadIdx.x; j < per_tile; j += 256) {
    const size_t b = base + j;
    if (b >= nb) break;
    u32 lo, c; seg_bucket(off, B, b, lo, c);
    if (c == 0) { aos_put_ext(buckets, b, Curve::identity()); continue; }
    const u32 full = c / P, rem = c - full * P;
    if (full) atomicAdd(&hist[0], full);
    if (rem) atomicAdd(&hist[P - rem], 1u);
  }
  __syncthreads();
  for (u32 k = threadIdx.x; k <= P; k += 256) bh[(size_t)k * gridDim.x + blockIdx.x] = hist[k];
}
// exclusive scan over `cnt` words of `v` by one 256-thread workgroup (in place); returns the total
static JJ_DEV u32 block_scan_excl(u32* v, u32 cnt, u32* sc /* [256] shared */) {
  const u32 tid = threadIdx.x, per = (cnt + 255) / 256, t0 = tid * per;
  u32 mine = 0;
  for (u32 t = t0; t < t0 + per && t < cnt; t++) mine += v[t];
  sc[tid] = mine;
  __syncthreads();
  for (u32 d = 1; d < 256; d <<= 1) {
    const u32 x = tid >= d ? sc[tid - d] : 0u;
    __syncthreads();
    sc[tid] += x;
    __syncthreads();
  }
  u32 run = sc[tid] - mine;
  for (u32 t = t0; t < t0 + per && t < cnt; t++) { const u32 c = v[t]; v[t] = run; run += c; }
  const u32 total = sc[255];
  __syncthreads();
  return total;
}
// between the passes, one workgroup per key: the key's row becomes each tile's first slot within the key, the key's total goes to
// tot[key].  (Round 2's plan was a single workgroup holding the whole matrix in LDS, which also capped the tiles of the two passes
// at ~230.)  The keys' first slots -- an exclusive scan over at most 1025 totals -- are formed by every workgroup of pass 2 itself.
__global__ void __launch_bounds__(256) k_seg_plan(u32 tiles, u32* bh, u32* tot /* [P + 1] */) {
  __shared__ u32 sc[256];
  const u32 total = block_scan_excl(bh + (size_t)blockIdx.x * tiles, tiles, sc);
  if (threadIdx.x == 0) tot[blockIdx.x] = total;
}
// pass 2: every segment takes the next slot of its key: the key's first slot (scan of the totals) + the tile's first slot within the
// key + a running count
__global__ void __launch_bounds__(256) k_seg_scatter(size_t nb, u32 B, u32 per_tile, u32 P, const u32* off, const u32* bh, const u32* tot, u32* total_out, Seg* seg,
                                                      u32* counters /* [0] heads, [1] merge items, [2] big buckets */, MergeItem* merge, BigBucket* big) {
  __shared__ u32 cur[SEG_PMAX + 1];
  __shared__ u32 sc[256];
  for (u32 k = threadIdx.x; k <= P; k += 256) cur[k] = tot[k];
  __syncthreads();
  const u32 all = block_scan_excl(cur, P + 1, sc);
  if (blockIdx.x == 0 && threadIdx.x == 0) *total_out = all;                   // number of segments, read by the accumulation
  for (u32 k = threadIdx.x; k <= P; k += 256) cur[k] += bh[(size_t)k * gridDim.x + blockIdx.x];
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * per_tile;
  const u32 lane = threadIdx.x & 63u;
  for (u32 j0 = 0; j0 < per_tile; j0 += 256) {                  // the same trips for every lane: the wave-wide sums below need all of them
    const u32 j = j0 + threadIdx.x;
    const size_t b = base + j;
    u32 lo = 0, c = 0;
    if (j < per_tile && b < nb) seg_bucket(off, B, b, lo, c);
    const u32 full = c / P, rem = c - full * P, nseg = full + (rem ? 1u : 0u);
    const u32 extra = nseg > 1 ? nseg - 1 : 0u;                  // heads this bucket needs beyond its own slot
    // ONE global atomic per wave for the heads and one for the merge items, instead of one each per bucket with several segments: uniform
    // scalars give almost none of those (mean bucket 32 entries, P = 64), skewed ones tens of thousands, all on the same two addresses
    u32 incl = extra;
    _Pragma("unroll") for (int d = 1; d < 64; d <<= 1) { const u32 o = (u32)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += o; }
    const u32 total = (u32)__shfl((int)incl, 63, 64);
    u32 wbase = 0;
    if (total) { if (lane == 63) wbase = atomicAdd(&counters[0], total); wbase = (u32)__shfl((int)wbase, 63, 64); }
    const u32 h0 = wbase + incl - extra;
    bool listed = false;
    if (extra > FIXUP_SERIAL_MAX) {                               // (rare: heavily repeated scalars)
      const u32 slot = atomicAdd(&counters[2], 1u);
      if (slot < FIXUP_BIG_MAX) { big[slot].bucket = (u32)b; big[slot].t_first = h0; big[slot].t_last = h0 + nseg - 2; big[slot].pad = 0; listed = true; }
    }
    const bool want = extra && !listed;
    const unsigned long long mball = __ballot(want);
    u32 mbase = 0;
    if (mball) {
      const int leader = __ffsll((long long)mball) - 1;
      if ((int)lane == leader) mbase = atomicAdd(&counters[1], (u32)__popcll(mball));
      mbase = (u32)__shfl((int)mbase, leader, 64);
    }
    if (want) { const u32 mi = mbase + (u32)__popcll(mball & ((1ull << lane) - 1ull)); merge[mi].bucket = (u32)b; merge[mi].h0 = h0; merge[mi].k = nseg - 1; merge[mi].pad = 0; }
    for (u32 sgi = 0; sgi < nseg; sgi++) {
      const u32 len = sgi < full ? P : rem;
      const u32 slot = atomicAdd(&cur[P - len], 1u);
      Seg sg; sg.start = lo + sgi * P; sg.len = len; sg.dst = sgi == 0 ? (u32)b : (0x80000000u | (h0 + sgi - 1)); sg.pad = 0;
      seg[slot] = sg;
    }
  }
}
__global__ void __launch_bounds__(256, JJ_MSM_ACC_MINBLOCKS) k_msm_accumulate_seg(const u32* nseg_total, const Seg* seg, const u32* idx, const u32* niels, ExtAoS buckets, ExtAoS head) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= *nseg_total) return;
  const Seg sg = seg[t];
  Ext acc = Curve::identity();
  const u32* ip = idx + sg.start;
  // software pipeline: the entry of term k+1 (a dependent 4-byte index load, then a random 128-byte gather) is in flight
  // while term k is added
  u32 e = ip[0];
  ANiels p = lds_aniels(niels + (size_t)(e & 0x7fffffffu) * GNIELS_WORDS);
  u32 e_next = sg.len > 1 ? ip[1] : e;
  #pragma unroll 1
  for (u32 k = 0; k < sg.len; k++) {
    const ANiels p_next = lds_aniels(niels + (size_t)(e_next & 0x7fffffffu) * GNIELS_WORDS);
    const u32 e_next2 = k + 2 < sg.len ? ip[k + 2] : e_next;
    acc = Curve::add_signed<true>(acc, p, (e >> 31) ? ~0u : 0u);
    e = e_next; p = p_next; e_next = e_next2;
  }
  if (sg.dst >> 31) aos_put_ext(head, sg.dst & 0x7fffffffu, acc); else aos_put_ext(buckets, sg.dst, acc);
}
// Buckets with a few extra segments (`merge`, segment path only; nullptr otherwise): one quad of lanes folds them in, grid-stride over
// the list -- it is short or empty unless scalars repeat.  (Its own launch until round 4: an empty pass over 1024 workgroups cost 13-16 us
// between the accumulation and the reduce; here it rides in the launch of the big buckets.)
// Big buckets: every workgroup folds a strided share of the listed buckets' heads into FIXUP_BIG_QUADS partials each
// (written to `partial[item][quad]`); the LAST workgroup to finish (a device-side counter) folds the partials of every item
// into its bucket.  The list is empty for anything but heavily repeated scalars.  The two lists name different buckets.
__global__ void __launch_bounds__(256) k_msm_fixup_big(u32* counters, const BigBucket* big, ExtAoS buckets, ExtAoS head, SoA partial, const MergeItem* merge) {
  __shared__ u32 last_s;
  if (merge) {
    const u32 role_m = threadIdx.x & 3u, mcnt = counters[1];
    #pragma unroll 1
    for (size_t m = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; m < mcnt; m += (size_t)gridDim.x * (blockDim.x >> 2)) {
      const MergeItem it = merge[m];
      Ext acc = aos_ext(buckets, it.bucket);
      #pragma unroll 1
      for (u32 j = 0; j < it.k; j++) acc = quad_add_ext(acc, aos_ext(head, (size_t)it.h0 + j), role_m);
      if (role_m == 0) aos_put_ext(buckets, it.bucket, acc);
    }
  }
  u32 cnt = counters[2]; if (cnt > FIXUP_BIG_MAX) cnt = FIXUP_BIG_MAX;
  if (cnt == 0) return;
  const u32 role = threadIdx.x & 3u, quad = threadIdx.x >> 2;
  #pragma unroll 1
  for (u32 item = blockIdx.x; item < cnt; item += gridDim.x) {
    const BigBucket bb = big[item];
    Ext acc = Curve::identity();
    #pragma unroll 1
    for (size_t t = (size_t)bb.t_first + quad; t <= bb.t_last; t += FIXUP_BIG_QUADS) acc = quad_add_ext(acc, aos_ext(head, t), role);
    if (role == 0) soa_put_ext(partial, (size_t)item * FIXUP_BIG_QUADS + quad, acc);
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last_s = atomicAdd(&counters[3], 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!last_s) return;
  __threadfence();
  #pragma unroll 1
  for (u32 item = quad; item < cnt; item += FIXUP_BIG_QUADS) {
    const BigBucket bb = big[item];
    Ext acc = aos_ext(buckets, bb.bucket);
    const u32 nh = bb.t_last - bb.t_first + 1, used = nh < FIXUP_BIG_QUADS ? nh : FIXUP_BIG_QUADS;
    #pragma unroll 1
    for (u32 q = 0; q < used; q++) acc = quad_add_ext(acc, soa_ext(partial, (size_t)item * FIXUP_BIG_QUADS + q), role);
    if (role == 0) aos_put_ext(buckets, bb.bucket, acc);
  }
}

// ================================================================================================ Pippenger: bucket reduction
// sum_j (j + 1) b_j of every window, on quads of lanes (the chains are latency-bound).  A chunk of L consecutive buckets j0 ..
// j0 + L - 1 contributes T + j0 S with T = sum (j - j0 + 1) b_j (running sums) and S = sum b_j; j0 S is a double-and-add over
// the significant bits of j0 (a multiple of L = 2^lb: lb plain doublings at the end).  Block (blk, s) takes the chunks blk * 64 +
// quad (+ 64 nblk ...) of slot s, its 64 quads are folded through LDS, and the last block of the window folds the blocks' sums
// and writes the window's point into the output record (msm_finish_window).  Windows narrower than the widest one use only the
// first 2^(width-1) of their B bucket slots; the chunks above are skipped.
// chunk k of slot s: T + j0 S, and its t1 * t2
static JJ_DEV Ext msm_reduce_chunk(const MsmParams& mp, u32 s, u32 k, u32 L, int lb, int jbits, const ExtAoS& buckets, u32 role, Fe& Tout) {
  const size_t first = (size_t)s * mp.B + (size_t)k * L;       // bucket index of the pass
  const u32 j0 = k * L;                                        // index inside the window
  // Every accumulator travels with T = t1*t2.  `running += bucket` is a TWO-round addition: the bucket enters in extended-Niels form
  // (V - U, V + U, 2Z, 2d T), whose 2d T is prepared two buckets ahead by the idle lanes of `total += running` (a three-round addition
  // between two accumulators: only lane 0 needs its middle round): bucket j - 2's t1*t2 on lane 1, 2d times bucket j - 1's on lane 2.
  // Five multiplication rounds per bucket instead of six (round 3).
  Ext running = Curve::identity(), total = Curve::identity();
  Fe Tr = Fq::zero(), Tt = Fq::zero(), dummy;
  Ext bk = aos_ext(buckets, first + L - 1);
  Ext bn = aos_ext(buckets, first + (L > 1 ? L - 2 : 0));
  // prologue, two rounds: lane 0: t1*t2 of the first bucket, lane 1: of the second; then lane 0: 2d * (the first)
  const Fe pr = Fq::mul(role_select4(bk.t1, bn.t1, bk.t1, bk.t1, role), role_select4(bk.t2, bn.t2, bk.t2, bk.t2, role));   // stored t1, t2 are carried
  Fe Tnext = quad_bcast<1>(pr);                                // t1*t2 of bucket j - 1
  Fe T2d = Fq::mul(quad_bcast<0>(pr), Fq::konst(FqP::D2));     // 2d t1*t2 of bucket j
  #pragma unroll 1
  for (int j = (int)L - 1; j >= 0; j--) {
    const Ext nn = aos_ext(buckets, first + (j > 1 ? j - 2 : 0));
    ENiels en;
    en.vpu = Fq::carry(Fq::add(bk.v, bk.u)); en.vmu = Fq::sub(bk.v, bk.u); en.z2 = Fq::add(bk.z, bk.z); en.t2d = T2d;
    running = quad_add_eniels(running, Tr, en, 0u, role, Tr);
    Fe Tnn, T2dn;
    total = quad_add_ext_t2(total, Tt, running, Tr, role, Tt, nn.t1, nn.t2, Tnn, Tnext, Fq::konst(FqP::D2), T2dn);
    bk = bn; bn = nn; T2d = T2dn; Tnext = Tnn;
  }
  // total += j0 * running   (j0 < B = 2^jbits).  j0 is a multiple of the chunk length L = 2^lb: double-and-add over the
  // jbits - lb significant bits, then lb plain doublings (no additions for bits that are zero by construction).  `running` enters
  // the additions in extended-Niels form too (one product for its 2d T, once): two rounds per addition instead of three.
  if (j0) {
    ENiels rn;
    rn.vpu = Fq::carry(Fq::add(running.v, running.u)); rn.vmu = Fq::sub(running.v, running.u); rn.z2 = Fq::add(running.z, running.z);
    rn.t2d = Fq::mul(Tr, Fq::konst(FqP::D2));
    const ENiels idn = Curve::eniels_identity();
    Ext m = Curve::identity();
    Fe Tm = Fq::zero();
    #pragma unroll 1
    for (int bit = jbits - 1; bit >= lb; bit--) {
      m = quad_dbl_t(m, role, Tm);
      const u32 mask = ((j0 >> bit) & 1u) ? ~0u : 0u;
      ENiels sel;
      sel.vpu = Fq::select(idn.vpu, rn.vpu, mask); sel.vmu = Fq::select(idn.vmu, rn.vmu, mask);
      sel.z2 = Fq::select(idn.z2, rn.z2, mask); sel.t2d = Fq::select(idn.t2d, rn.t2d, mask);
      m = quad_add_eniels(m, Tm, sel, 0u, role, Tm);
    }
    #pragma unroll 1
    for (int bit = 0; bit < lb; bit++) m = quad_dbl_t(m, role, Tm);
    total = quad_add_ext_t(total, Tt, m, Tm, role, Tt, Tr, Tr, dummy);
  }
  Tout = Tt;
  return total;
}
// workgroups of 64 quads that window slot s needs: its chunks of L buckets, 64 to a workgroup, at most nblk
static __host__ __device__ __forceinline__ u32 msm_reduce_blocks(const MsmParams& mp, int s, u32 L, u32 nblk) {
  const int w = mp.w0 + s * mp.wstride, width = mp.c + (w < mp.r ? 1 : 0);
  const u32 Kw = ((1u << (width - 1)) + L - 1) / L, need = (Kw + MSM_TREE_QUADS - 1) / MSM_TREE_QUADS;
  return need < nblk ? need : nblk;
}
// MULTI: a quad may own several chunks (only with tuning overrides that leave more than 64 * 64 chunks per window).
// The grid is ONE-dimensional and holds only workgroups that have chunks (a window one bit narrower than the widest needs half as
// many): the dispatcher places every launched workgroup at once, so workgroups beyond one per CU would share a CU with another
// chain from the start and both would run ~1.6x longer -- 272 launched for 256 that work cost 217 us instead of 130.
template <bool MULTI>
__global__ void __launch_bounds__(4 * MSM_TREE_QUADS) k_msm_reduce_fold(size_t n, MsmParams mp, u32 L, u32 nblk, int jbits, ExtAoS buckets, u32* part, u32* counters, u32* rec) {
  __shared__ __attribute__((aligned(16))) u32 st[MSM_TREE_QUADS * LDS_PT_WORDS];
  const u32 role = threadIdx.x & 3u, quad = threadIdx.x >> 2;
  u32 blk = blockIdx.x, s = 0, nblk_s = msm_reduce_blocks(mp, 0, L, nblk);
  while (blk >= nblk_s) { blk -= nblk_s; s++; nblk_s = msm_reduce_blocks(mp, (int)s, L, nblk); }      // the host launched exactly the sum
  const int w = msm_slot_window(mp, (int)s);
  if (blockIdx.x == 0 && threadIdx.x == 0) msm_write_header(rec, mp, n);
  const u32 Bw = 1u << (msm_win_width(mp, w) - 1);
  const u32 Kw = (Bw + L - 1) / L;                              // chunks of this window that hold buckets
  const int lb = __ffs((int)L) - 1;
  Ext acc = Curve::identity();
  Fe Tacc = Fq::zero();
  const u32 k0 = blk * MSM_TREE_QUADS + quad;
  if (k0 < Kw) acc = msm_reduce_chunk(mp, s, k0, L, lb, jbits, buckets, role, Tacc);
  if constexpr (MULTI) {
    #pragma unroll 1
    for (u32 k = k0 + MSM_TREE_QUADS * nblk_s; k < Kw; k += MSM_TREE_QUADS * nblk_s) {
      Fe Tt, dummy;
      const Ext total = msm_reduce_chunk(mp, s, k, L, lb, jbits, buckets, role, Tt);
      acc = quad_add_ext_t(acc, Tacc, total, Tt, role, Tacc, Tt, Tt, dummy);
    }
  }
  const u32 base = blk * MSM_TREE_QUADS;
  const u32 live = base >= Kw ? 0u : (Kw - base < (u32)MSM_TREE_QUADS ? Kw - base : (u32)MSM_TREE_QUADS);
  quad_tree_sum(st, quad, role, live, acc, Tacc);
  msm_finish_window(st, quad, role, nblk_s, nblk, blk, s, w, acc, Tacc, part, counters, rec);
}

// ---- Two-level reduction for wide windows (round 5).  The quad chains above spend as many instructions on exchanges, role selects
// and idle lanes as on the mathematics (~4x the products' own cost), and with 2^15 buckets per window every quad walks 32 of them.
// Level 1 runs in LANE form, at the throughput of whole-lane additions: the window's Bw buckets are read as an R x M matrix
// (bucket j = i M + m, M = 2^mbits, R = Bw / M rows), lane m takes column m, i.e. the STRIDED buckets m, m + M, ..., and leaves
//     S_m = sum_i b_{i M + m}          T_m = sum_i i b_{i M + m}            (running sums from the top row: 2 (R - 2) + 1 additions)
// so that   sum_j (j + 1) b_j = sum_m (m + 1) S_m + M sum_m T_m.   Level 2 (quads again) sees an R-times smaller weighted sum over
// the S_m and a plain sum of the T_m; the weight M = 2^mbits of that plain sum costs nothing: a chunk's j0 * (sum S) is a
// double-and-add over the bits [lb, mbits) of j0, and the chunk's sum of T_m is simply the value that chain starts from (bit mbits).
// Both arrays leave level 1 as extended-Niels records (144 B), the form level 2's two-round additions consume.
__global__ void __launch_bounds__(256) k_msm_reduce_l1(MsmParams mp, int mbits, ExtAoS buckets, u32* SN, u32* TN) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ((size_t)mp.Ws << mbits)) return;
  const u32 M = 1u << mbits, s = (u32)(t >> mbits), m = (u32)t & (M - 1u);
  const int w = msm_slot_window(mp, (int)s);
  const u32 R = (1u << (msm_win_width(mp, w) - 1)) >> mbits;          // rows of this window (the host keeps M <= the narrowest window's buckets): uniform over a wave
  const size_t base = (size_t)s * mp.B + m;
  // running sums from the top row: running_i = b_i + ... + b_{R-1}, T = sum_{i >= 1} running_i, S = running_0.  T1*T2 of each accumulator
  // is formed once per step and serves both the addition into it and its hand-over as an operand: 19 products per row.
  // (Written as two independent chains per step -- total += running_i beside running_{i-1} = running_i + b_{i-1} -- it needs 272
  // registers and runs no faster: 70.0 against 70.5 us.)
  Ext running = aos_ext(buckets, base + (size_t)(R - 1) * M);
  Fe Tr = Curve::tt<true>(running);                                    // stored t1, t2 are carried
  Ext total = running;
  Fe Tt = Tr;
  if (R == 1) { total = Curve::identity(); Tt = Fq::zero(); }
  else {
    Ext nxt = aos_ext(buckets, base + (size_t)(R - 2) * M);
    #pragma unroll 1
    for (int i = (int)R - 2; i >= 1; i--) {
      const Ext cur = nxt;
      nxt = aos_ext(buckets, base + (size_t)(i - 1) * M);            // the next bucket is in flight while this one is added
      running = Curve::add_t(running, Tr, Curve::to_niels<true>(cur));
      Tr = Curve::tt<true>(running);
      total = Curve::add_t(total, Tt, Curve::to_niels_t(running, Tr));
      Tt = Curve::tt<true>(total);
    }
    running = Curve::add_t(running, Tr, Curve::to_niels<true>(nxt));   // + b_0: S complete
    Tr = Curve::tt<true>(running);
  }
  store_eniels(SN + t * ENIELS_WORDS, Curve::to_niels_t(running, Tr));
  store_eniels(TN + t * ENIELS_WORDS, Curve::to_niels_t(total, Tt));
}
// level 2, chunk k of slot s: elements m = k L .. k L + L - 1;  sum (m + 1) S_m + M sum T_m  =  sum (m - j0 + 1) S_m  +  [2^mbits sum T_m + j0 sum S_m]
static JJ_DEV Ext msm_reduce2_chunk(u32 s, u32 k, u32 L, int lb, int mbits, const u32* SN, const u32* TN, u32 role, Fe& Tout) {
  const size_t first = (((size_t)s << mbits) + (size_t)k * L) * ENIELS_WORDS;
  const u32 j0 = k * L;
  Ext running = Curve::identity(), total = Curve::identity(), plain = Curve::identity();
  Fe Tr = Fq::zero(), Tt = Fq::zero(), Tp = Fq::zero(), dummy;
  ENiels sn = load_eniels(SN + first + (size_t)(L - 1) * ENIELS_WORDS), tn = load_eniels(TN + first + (size_t)(L - 1) * ENIELS_WORDS);
  // each record is fetched again right after its last use, five (S) or seven (T) multiplication rounds before the next one: no second
  // register set for the prefetch
  #pragma unroll 1
  for (int j = (int)L - 1; j >= 0; j--) {
    running = quad_add_eniels(running, Tr, sn, 0u, role, Tr);
    if (j > 0) sn = load_eniels(SN + first + (size_t)(j - 1) * ENIELS_WORDS);
    total = quad_add_ext_t(total, Tt, running, Tr, role, Tt, Tr, Tr, dummy);
    plain = quad_add_eniels(plain, Tp, tn, 0u, role, Tp);
    if (j > 0) tn = load_eniels(TN + first + (size_t)(j - 1) * ENIELS_WORDS);
  }
  // m = 2^mbits plain + j0 running: double-and-add over the bits [lb, mbits) of j0 (a multiple of L = 2^lb) starting from `plain`
  ENiels rn;
  rn.vpu = Fq::carry(Fq::add(running.v, running.u)); rn.vmu = Fq::sub(running.v, running.u); rn.z2 = Fq::add(running.z, running.z);
  rn.t2d = Fq::mul(Tr, Fq::konst(FqP::D2));
  const ENiels idn = Curve::eniels_identity();
  Ext m = plain;
  Fe Tm = Tp;
  #pragma unroll 1
  for (int bit = mbits - 1; bit >= lb; bit--) {
    m = quad_dbl_t(m, role, Tm);
    const u32 mask = ((j0 >> bit) & 1u) ? ~0u : 0u;
    m = quad_add_eniels(m, Tm, Curve::select(idn, rn, mask), 0u, role, Tm);
  }
  #pragma unroll 1
  for (int bit = 0; bit < lb; bit++) m = quad_dbl_t(m, role, Tm);
  total = quad_add_ext_t(total, Tt, m, Tm, role, Tt, Tr, Tr, dummy);
  Tout = Tt;
  return total;
}
// grid: Ws * nblk workgroups of 64 quads, nblk = ceil(M / L / 64) <= 64 per window (the same for every window: M does not depend on it)
__global__ void __launch_bounds__(4 * MSM_TREE_QUADS) k_msm_reduce_l2(size_t n, MsmParams mp, int mbits, u32 L, u32 nblk, const u32* SN, const u32* TN, u32* part, u32* counters, u32* rec) {
  __shared__ __attribute__((aligned(16))) u32 st[MSM_TREE_QUADS * LDS_PT_WORDS];
  const u32 role = threadIdx.x & 3u, quad = threadIdx.x >> 2;
  const u32 s = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const int w = msm_slot_window(mp, (int)s);
  if (blockIdx.x == 0 && threadIdx.x == 0) msm_write_header(rec, mp, n);
  const u32 K = (1u << mbits) / L;
  const int lb = __ffs((int)L) - 1;
  Ext acc = Curve::identity();
  Fe Tacc = Fq::zero();
  const u32 k = blk * MSM_TREE_QUADS + quad;
  if (k < K) acc = msm_reduce2_chunk(s, k, L, lb, mbits, SN, TN, role, Tacc);
  const u32 base = blk * MSM_TREE_QUADS;
  const u32 live = base >= K ? 0u : (K - base < (u32)MSM_TREE_QUADS ? K - base : (u32)MSM_TREE_QUADS);
  quad_tree_sum(st, quad, role, live, acc, Tacc);
  msm_finish_window(st, quad, role, nblk, nblk, blk, s, w, acc, Tacc, part, counters, rec);
}

// ================================================================================================ fold of gathered records
// The multi-rank MSM all-gathers one record per rank (jj_msm_allgather).  Until round 4 every rank copied all G records to
// the host and added them window by window on one CPU thread: the only part of the exchange that grows with G.  Here workgroup w
// sums window w of the G records on quads of lanes (quad g takes records g, g + 64, ...; LDS tree) and writes ONE record whose window
// mask is the union and whose term count is the sum: one 8 KB copy and the single-record host tail follow, whatever G is.  Both
// partitions (terms: every record holds every window; windows: the masks are disjoint) and empty shards (mask 0) pass through the same code.
// Records of DIFFERENT window layouts (a shard small enough for the 64-window small-batch path beside Pippenger shards) cannot be added
// window by window: the output header then carries magic 0 and the caller falls back to the host's combine_records over all G.
static JJ_DEV bool msm_rec_point(const u32* rec, int w, u32 role, Ext& p, Fe& T) {      // window w of one record -> Montgomery form; false if the record lacks it
  const u64 mask = (u64)rec[4] | ((u64)rec[5] << 32);
  if (!((mask >> w) & 1ull)) return false;
  u32 wd[8];
  load8(wd, rec + MSM_REC_HDR_WORDS + (size_t)w * MSM_REC_PT_WORDS, role);               // lane r converts coordinate r (U, V, Z, T)
  const Fe cr = Fq::mul(Fq::unpack(wd), Fq::konst(FqP::FROM_HOST));
  p.u = quad_bcast<0>(cr); p.v = quad_bcast<1>(cr); p.z = quad_bcast<2>(cr); T = quad_bcast<3>(cr);
  p.t1 = p.u; p.t2 = p.v;                                                                // unused by the T-carrying quad operations
  return true;
}
// the same record entry written by the four lanes of a quad together (lane r converts and stores coordinate r): one product round
// instead of four in a row on one lane
static JJ_DEV void msm_store_window_quad(u32* points, int w, const Ext& acc, const Fe& T, u32 role) {
  u32 wd[8];
  Fq::pack(wd, Fq::canon_plain_product(Fq::mul(role_select4(acc.u, acc.v, acc.z, T, role), Fq::konst(FqP::HOST_R))));
  store8(points + (size_t)w * MSM_REC_PT_WORDS, role, wd);
}
__global__ void __launch_bounds__(4 * MSM_TREE_QUADS) k_msm_fold_records(const u32* recs, u32 G, u32 stride_words, u32* out) {
  __shared__ __attribute__((aligned(16))) u32 st[MSM_TREE_QUADS * LDS_PT_WORDS];
  __shared__ u32 hdr_s[8];
  const u32 role = threadIdx.x & 3u, quad = threadIdx.x >> 2;
  const int w = (int)blockIdx.x;
  // every workgroup reads the G headers itself (32 bytes each, one thread per record, G <= 256 at a time): layout, union mask, term count
  if (threadIdx.x < 8) hdr_s[threadIdx.x] = threadIdx.x == 0 ? 1u : 0u;        // [0] ok  [1] Wref  [2..3] mask  [4..5] n
  __syncthreads();
  for (u32 g0 = 0; g0 < G; g0 += blockDim.x) {
    const u32 g = g0 + threadIdx.x;
    if (g < G) {
      const uint4* hp = reinterpret_cast<const uint4*>(recs + (size_t)g * stride_words);
      const uint4 h0 = hp[0], h1 = hp[1];
      if (h0.x != MSM_REC_MAGIC || h0.y != 2u || h0.z == 0 || h0.z > 64u) atomicAnd(&hdr_s[0], 0u);
      else {
        if (h1.x | h1.y) { atomicMax(&hdr_s[1], h0.z); atomicOr(&hdr_s[2], h1.x); atomicOr(&hdr_s[3], h1.y); }      // (an empty shard: any layout)
        const u32 old = atomicAdd(&hdr_s[4], h1.z);
        if (old + h1.z < old) atomicAdd(&hdr_s[5], 1u);
        if (h1.w) atomicAdd(&hdr_s[5], h1.w);
      }
    }
  }
  __syncthreads();
  // all non-empty records must share one layout: the largest W seen is the reference, a second pass finds any other
  const u32 Wmax = hdr_s[1];
  for (u32 g0 = 0; g0 < G; g0 += blockDim.x) {
    const u32 g = g0 + threadIdx.x;
    if (g < G) {
      const u32* h = recs + (size_t)g * stride_words;
      if ((h[4] | h[5]) && h[2] != Wmax) atomicAnd(&hdr_s[0], 0u);
    }
  }
  __syncthreads();
  const u32 ok = hdr_s[0], Wref = Wmax ? Wmax : (G ? recs[2] : (u32)SM_W);
  const u64 mask = (u64)hdr_s[2] | ((u64)hdr_s[3] << 32);
  if (w == 0 && threadIdx.x == 0) {
    out[1] = 2u; out[2] = Wref; out[3] = 1u; out[4] = hdr_s[2]; out[5] = hdr_s[3]; out[6] = hdr_s[4]; out[7] = hdr_s[5];
    for (int j = 8; j < MSM_REC_HDR_WORDS; j++) out[j] = 0;
    out[0] = ok ? MSM_REC_MAGIC : 0u;
  }
  if (!ok || w >= (int)Wref || !((mask >> w) & 1ull)) return;          // uniform over the workgroup
  Ext acc = Curve::identity();
  Fe Tacc = Fq::zero();
  #pragma unroll 1
  for (u32 g = quad; g < G; g += MSM_TREE_QUADS) {
    Ext p; Fe Tp, dummy;
    if (msm_rec_point(recs + (size_t)g * stride_words, w, role, p, Tp)) acc = quad_add_ext_t(acc, Tacc, p, Tp, role, Tacc, Tp, Tp, dummy);
  }
  const u32 live = G < (u32)MSM_TREE_QUADS ? G : (u32)MSM_TREE_QUADS;
  quad_tree_sum(st, quad, role, live, acc, Tacc);
  if (quad == 0) msm_store_window_quad(out + MSM_REC_HDR_WORDS, w, acc, Tacc, role);
}

// ================================================================================================ device-side finish (opt-in)
// jj_msm_dev: the host tail's work -- Horner over the windows of ONE record (252 dependent doublings + one addition per window),
// one inversion, canonical (u, v) -- on one quad of lanes, so that the sum never leaves the device and no host thread waits.
// Same formulas as jj_host_tail.h WindowSums::finish.  A chain: ~2 x 252 multiplication rounds + the inversion's ~330 products on
// a single lane of a single wave, ~0.5 ms -- ten times the host tail (profiles/r4_msm_dev_finish.txt): for pipelines that must
// not synchronise with the host, not for latency.
__global__ void __launch_bounds__(64) k_msm_finish_dev(const u32* rec, void* out64) {
  const u32 role = threadIdx.x & 3u;
  if (threadIdx.x >= 4) return;                                  // one quad; the other lanes of the wave leave
  const int W = (int)rec[2];
  const u64 mask = (u64)rec[4] | ((u64)rec[5] << 32);
  const int c = 253 / W, r = 253 % W;
  Ext acc = Curve::identity();
  Fe T = Fq::zero();
  bool any = false;
  #pragma unroll 1
  for (int w = W - 1; w >= 0; w--) {
    if (any) {
      const int width = c + (w < r ? 1 : 0);
      #pragma unroll 1
      for (int i = 0; i < width; i++) acc = quad_dbl_t(acc, role, T);
    }
    if ((mask >> w) & 1ull) {
      // the window's point: U, V, Z, T as value * 2^256 mod q (plain canonical words) -> Montgomery form
      const u32* src = rec + MSM_REC_HDR_WORDS + (size_t)w * MSM_REC_PT_WORDS;
      u32 wd[8];
      Ext p; Fe Tp;
      load8(wd, src, 0); p.u = Fq::mul(Fq::unpack(wd), Fq::konst(FqP::FROM_HOST));
      load8(wd, src, 1); p.v = Fq::mul(Fq::unpack(wd), Fq::konst(FqP::FROM_HOST));
      load8(wd, src, 2); p.z = Fq::mul(Fq::unpack(wd), Fq::konst(FqP::FROM_HOST));
      load8(wd, src, 3); Tp = Fq::mul(Fq::unpack(wd), Fq::konst(FqP::FROM_HOST));
      p.t1 = p.u; p.t2 = p.v;                                     // unused by the T-carrying quad operations
      if (any) { Fe dummy; acc = quad_add_ext_t(acc, T, p, Tp, role, T, Tp, Tp, dummy); }
      else { acc = p; T = Tp; any = true; }
    }
  }
  if (role == 0) {
    const Fe zi = Fq::invert(acc.z);
    store_affine(out64, 0, Fq::mul(acc.u, zi), Fq::mul(acc.v, zi));
  }
}
