// jubjub_hip.hpp — C++17 host-side mirror of the zkcrypto/jubjub public API for the batched MI355X engine.
//
// Header-only layer above the C ABI (jubjub_hip.h).  It keeps the reference crate's names, argument meaning and
// error behaviour for the scalar-multiplication path, lifted from one element to a batch (std::vector of wire
// encodings).  The reference is Rust; no Rust toolchain exists in this image, so this C++ mirror (plus the Python
// one in jubjub_amd/) is the host side that is compiled and tested here.
//
//   reference (src/lib.rs, src/fr.rs)                          here
//   ---------------------------------------------------------  -----------------------------------------------
//   Fr / Fq   (to_bytes, from_bytes, add, sub, mul, ...)        jubjub::FrBatch / jubjub::FqBatch
//   AffinePoint::{to_bytes, from_bytes, batch_from_bytes,       jubjub::AffineBatch::{to_bytes, from_bytes,
//     from_bytes_pre_zip216_compatibility, to_niels,              from_bytes_pre_zip216_compatibility, to_niels,
//     mul_by_cofactor, is_small_order, is_torsion_free,           mul_by_cofactor, is_small_order, is_torsion_free,
//     is_prime_order, is_identity}                                is_prime_order, is_identity, is_on_curve}
//   &ExtendedPoint * &Fr  (lib.rs:873-879)                      operator*(const AffineBatch&, const FrBatch&)
//   ExtendedPoint::{double, +, -, neg}                          AffineBatch::{double_, operator+, operator-, neg}
//   AffineNielsPoint::multiply_bits (lib.rs:297-301)            FixedBase::multiply_bits / operator*
//   batch_normalize (lib.rs:1084-1107)                          jubjub::batch_normalize
//   iter.sum() (lib.rs:183-193)                                 AffineBatch::sum ; jubjub::msm
//   SubgroupPoint::from_bytes (lib.rs:1427-1429)                AffineBatch::from_bytes(..., DecodeFlags::subgroup())
//
// Results cross the boundary as the crate's own public encodings, so they can be fed straight back into the Rust
// types with Fq::from_bytes / AffinePoint::from_raw_unchecked.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "jubjub_hip.h"

namespace jubjub {

using Bytes32 = std::array<uint8_t, 32>;
using Bytes64 = std::array<uint8_t, 64>;

class Error : public std::runtime_error {
 public:
  Error(int status, const std::string& what) : std::runtime_error(what), status(status) {}
  int status;
};

// Owns one device context (one per GPU; one process per GPU is the intended deployment).
class Context {
 public:
  explicit Context(int device = 0) {
    const int rc = jj_ctx_create(device, &ctx_);
    if (rc != JJ_OK) throw Error(rc, rc == JJ_ERR_NODEVICE ? "no gfx950 GPU visible (there is no CPU fallback)" : "jj_ctx_create failed");
  }
  ~Context() { if (ctx_) jj_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  jj_ctx* raw() const { return ctx_; }
  void check(int rc) const { if (rc != JJ_OK) throw Error(rc, std::string("libjubjub_hip: ") + jj_last_error(ctx_)); }

 private:
  jj_ctx* ctx_ = nullptr;
};

// `CtOption<T>` for a batch: values plus one Choice byte each (1 = Some).  Values of None entries are zero.
template <class T>
struct CtOptionBatch {
  T value;
  std::vector<uint8_t> is_some;
  bool all_some() const { for (auto b : is_some) if (!b) return false; return true; }
};

// ------------------------------------------------------------------------------------------------ fields
template <bool IS_FR>
class FieldBatch {
 public:
  FieldBatch(const Context& c, std::vector<Bytes32> v) : c_(&c), v_(std::move(v)) {}
  static FieldBatch from_u64(const Context& c, const std::vector<uint64_t>& xs) {   // From<u64> for Fr (fr.rs:42-46)
    std::vector<Bytes32> v(xs.size());
    for (size_t i = 0; i < xs.size(); i++) { v[i].fill(0); for (int b = 0; b < 8; b++) v[i][b] = (uint8_t)(xs[i] >> (8 * b)); }
    return FieldBatch(c, std::move(v));
  }
  // from_bytes (fr.rs:268-292): None when the integer is not below the modulus
  static CtOptionBatch<FieldBatch> from_bytes(const Context& c, const std::vector<Bytes32>& in) {
    CtOptionBatch<FieldBatch> r{FieldBatch(c, std::vector<Bytes32>(in.size())), std::vector<uint8_t>(in.size())};
    c.check((IS_FR ? jj_fr_from_bytes : jj_fq_from_bytes)(c.raw(), in.size(), in.data(), r.value.v_.data(), r.is_some.data()));
    return r;
  }
  // from_bytes_wide (fr.rs:312-343)
  static FieldBatch from_bytes_wide(const Context& c, const std::vector<Bytes64>& in) {
    FieldBatch r(c, std::vector<Bytes32>(in.size()));
    c.check((IS_FR ? jj_fr_from_bytes_wide : jj_fq_from_bytes_wide)(c.raw(), in.size(), in.data(), r.v_.data()));
    return r;
  }
  const std::vector<Bytes32>& to_bytes() const { return v_; }   // fr.rs:296-308 (already canonical)
  size_t len() const { return v_.size(); }
  // PrimeFieldBits::to_le_bits (fr.rs:746-773): 256 bytes of 0/1 per element, bit 0 first
  std::vector<std::array<uint8_t, 256>> to_le_bits() const {
    std::vector<std::array<uint8_t, 256>> out(len());
    c_->check((IS_FR ? jj_fr_to_le_bits : jj_fq_to_le_bits)(c_->raw(), len(), v_.data(), out.data()));
    return out;
  }
  // Field::random (fr.rs:684-688; bls12_381::Scalar likewise): 64 PRNG bytes per element through from_bytes_wide, for both
  // fields.  The two 32-byte halves come from two decorrelated streams of the library's counter-based generator (the seed
  // hashed with a domain tag per half; jubjub_amd/group.py random_stream_seeds is the same function).  Test data only: the
  // generator is reproducible by design and not a source of secret scalars.
  static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  static FieldBatch random(const Context& c, size_t n, uint64_t seed, uint64_t first_index = 0) {
    std::vector<Bytes32> lo(n), hi(n);
    c.check(jj_synth_bytes32(c.raw(), n, splitmix64(seed ^ 0x6F4C646E72466A6Aull), first_index, lo.data()));
    c.check(jj_synth_bytes32(c.raw(), n, splitmix64(seed ^ 0x6948646E72466A6Aull), first_index, hi.data()));
    std::vector<std::array<uint8_t, 64>> wide(n);
    for (size_t i = 0; i < n; i++) { std::memcpy(wide[i].data(), lo[i].data(), 32); std::memcpy(wide[i].data() + 32, hi[i].data(), 32); }
    FieldBatch r(c, std::vector<Bytes32>(n));
    c.check((IS_FR ? jj_fr_from_bytes_wide : jj_fq_from_bytes_wide)(c.raw(), n, wide.data(), r.v_.data()));
    return r;
  }

  FieldBatch operator+(const FieldBatch& o) const { return bin(IS_FR ? jj_fr_add : jj_fq_add, o); }
  FieldBatch operator-(const FieldBatch& o) const { return bin(IS_FR ? jj_fr_sub : jj_fq_sub, o); }
  FieldBatch operator*(const FieldBatch& o) const { return bin(IS_FR ? jj_fr_mul : jj_fq_mul, o); }
  FieldBatch operator-() const { return un(IS_FR ? jj_fr_neg : jj_fq_neg); }
  FieldBatch square() const { return un(IS_FR ? jj_fr_square : jj_fq_square); }
  FieldBatch double_() const { return un(IS_FR ? jj_fr_double : jj_fq_double); }
  CtOptionBatch<FieldBatch> invert() const { return opt(IS_FR ? jj_fr_invert : jj_fq_invert); }   // fr.rs:438-540
  CtOptionBatch<FieldBatch> sqrt() const { return opt(IS_FR ? jj_fr_sqrt : jj_fq_sqrt); }         // fr.rs:384-399
  bool operator==(const FieldBatch& o) const { return v_ == o.v_; }

 private:
  using Bin = int (*)(jj_ctx*, size_t, const void*, const void*, void*);
  using Un = int (*)(jj_ctx*, size_t, const void*, void*);
  using Opt = int (*)(jj_ctx*, size_t, const void*, void*, uint8_t*);
  FieldBatch bin(Bin f, const FieldBatch& o) const {
    if (o.len() != len()) throw Error(JJ_ERR_INVALID, "length mismatch");
    FieldBatch r(*c_, std::vector<Bytes32>(len()));
    c_->check(f(c_->raw(), len(), v_.data(), o.v_.data(), r.v_.data()));
    return r;
  }
  FieldBatch un(Un f) const { FieldBatch r(*c_, std::vector<Bytes32>(len())); c_->check(f(c_->raw(), len(), v_.data(), r.v_.data())); return r; }
  CtOptionBatch<FieldBatch> opt(Opt f) const {
    CtOptionBatch<FieldBatch> r{FieldBatch(*c_, std::vector<Bytes32>(len())), std::vector<uint8_t>(len())};
    c_->check(f(c_->raw(), len(), v_.data(), r.value.v_.data(), r.is_some.data()));
    return r;
  }
  const Context* c_;
  std::vector<Bytes32> v_;
};
using FrBatch = FieldBatch<true>;
using FqBatch = FieldBatch<false>;

// ------------------------------------------------------------------------------------------------ points
struct DecodeFlags {
  unsigned bits = JJ_DECOMPRESS_ZIP216;
  static DecodeFlags zip216() { return DecodeFlags{JJ_DECOMPRESS_ZIP216}; }                          // AffinePoint::from_bytes
  static DecodeFlags pre_zip216() { return DecodeFlags{0}; }                                          // from_bytes_pre_zip216_compatibility
  static DecodeFlags subgroup() { return DecodeFlags{JJ_DECOMPRESS_ZIP216 | JJ_DECOMPRESS_TORSION_FREE}; }  // SubgroupPoint::from_bytes
};

class AffineBatch {
 public:
  AffineBatch(const Context& c, std::vector<Bytes64> p) : c_(&c), p_(std::move(p)) {}
  // AffinePoint::from_raw_unchecked (lib.rs:662-664)
  static AffineBatch from_raw_unchecked(const Context& c, std::vector<Bytes64> uv) { return AffineBatch(c, std::move(uv)); }
  static AffineBatch identity(const Context& c, size_t n) {                                           // lib.rs:416-421
    std::vector<Bytes64> p(n); for (auto& e : p) { e.fill(0); e[32] = 1; } return AffineBatch(c, std::move(p));
  }
  static AffineBatch generator(const Context& c, size_t n) {                                          // lib.rs:1380-1396
    static const uint8_t U[32] = {0xfe, 0xad, 0xa7, 0xf1, 0x5d, 0xd3, 0xb3, 0xe4, 0xaf, 0x81, 0xbf, 0x29, 0x1b, 0x5d, 0xf5, 0xca,
                                  0x87, 0x81, 0x0a, 0xd6, 0xdd, 0x03, 0x0f, 0x8b, 0xc8, 0x87, 0x37, 0xbf, 0xb8, 0xcb, 0xed, 0x62};
    std::vector<Bytes64> p(n); for (auto& e : p) { e.fill(0); std::memcpy(e.data(), U, 32); e[32] = 11; } return AffineBatch(c, std::move(p));
  }
  // ExtendedPoint::random / SubgroupPoint::random (lib.rs:1244-1267, 1290-1298) over the library's counter-based stream
  static AffineBatch random(const Context& c, size_t n, uint64_t seed, uint64_t first_index = 0, bool subgroup = false) {
    std::vector<Bytes64> p(n);
    c.check(jj_random_points(c.raw(), n, seed, first_index, subgroup ? 1 : 0, p.data(), nullptr));
    return AffineBatch(c, std::move(p));
  }
  // AffinePoint::from_bytes / batch_from_bytes / from_bytes_pre_zip216_compatibility (lib.rs:469-627)
  static CtOptionBatch<AffineBatch> from_bytes(const Context& c, const std::vector<Bytes32>& enc, DecodeFlags f = DecodeFlags::zip216()) {
    CtOptionBatch<AffineBatch> r{AffineBatch(c, std::vector<Bytes64>(enc.size())), std::vector<uint8_t>(enc.size())};
    c.check(jj_decompress(c.raw(), enc.size(), enc.data(), f.bits, r.value.p_.data(), r.is_some.data()));
    return r;
  }
  static CtOptionBatch<AffineBatch> batch_from_bytes(const Context& c, const std::vector<Bytes32>& enc) { return from_bytes(c, enc); }
  static CtOptionBatch<AffineBatch> from_bytes_pre_zip216_compatibility(const Context& c, const std::vector<Bytes32>& enc) {
    return from_bytes(c, enc, DecodeFlags::pre_zip216());
  }
  std::vector<Bytes32> to_bytes() const {                                                             // lib.rs:455-464
    std::vector<Bytes32> out(len()); c_->check(jj_compress(c_->raw(), len(), p_.data(), out.data())); return out;
  }
  const std::vector<Bytes64>& coords() const { return p_; }                                           // get_u / get_v (lib.rs:630-637)
  size_t len() const { return p_.size(); }

  AffineBatch double_() const { return un(jj_point_double); }                                          // lib.rs:739-828
  AffineBatch neg() const { return un(jj_point_neg); }                                                 // lib.rs:92-104
  AffineBatch mul_by_cofactor() const { return un(jj_point_mul_by_cofactor); }                         // lib.rs:722-724
  AffineBatch operator+(const AffineBatch& o) const { return bin(jj_point_add, o); }                   // lib.rs:1012-1019
  AffineBatch operator-(const AffineBatch& o) const { return bin(jj_point_sub, o); }                   // lib.rs:1021-1028
  std::vector<std::array<uint8_t, 96>> to_niels() const {                                              // lib.rs:652-658
    std::vector<std::array<uint8_t, 96>> out(len()); c_->check(jj_point_to_niels(c_->raw(), len(), p_.data(), out.data())); return out;
  }
  std::vector<uint8_t> is_identity() const { return pred(jj_is_identity); }                            // lib.rs:424-426
  std::vector<uint8_t> is_small_order() const { return pred(jj_is_small_order); }                      // lib.rs:435-437
  std::vector<uint8_t> is_torsion_free() const { return pred(jj_is_torsion_free); }                    // lib.rs:441-443
  std::vector<uint8_t> is_prime_order() const { return pred(jj_is_prime_order); }                      // lib.rs:449-452
  std::vector<uint8_t> is_on_curve() const { return pred(jj_is_on_curve); }                            // lib.rs:670-675
  // ExtendedPoint::multiply / multiply_bits: raw 32-byte patterns, low 252 bits used (lib.rs:357-385, 831-833); constant-time like the reference (jj_varbase_mul)
  AffineBatch multiply_bits(const std::vector<Bytes32>& by) const {
    if (by.size() != len()) throw Error(JJ_ERR_INVALID, "length mismatch");
    AffineBatch r(*c_, std::vector<Bytes64>(len()));
    c_->check(jj_varbase_mul(c_->raw(), len(), by.data(), p_.data(), r.p_.data()));
    return r;
  }
  AffineBatch operator*(const FrBatch& k) const { return multiply_bits(k.to_bytes()); }                // lib.rs:873-879, 1109-1115
  // Sum (lib.rs:183-193): one point
  Bytes64 sum() const { Bytes64 out; c_->check(jj_point_sum(c_->raw(), len(), p_.data(), out.data())); return out; }
  bool operator==(const AffineBatch& o) const { return p_ == o.p_; }

 private:
  using Un = int (*)(jj_ctx*, size_t, const void*, void*);
  using Bin = int (*)(jj_ctx*, size_t, const void*, const void*, void*);
  using Pred = int (*)(jj_ctx*, size_t, const void*, uint8_t*);
  AffineBatch un(Un f) const { AffineBatch r(*c_, std::vector<Bytes64>(len())); c_->check(f(c_->raw(), len(), p_.data(), r.p_.data())); return r; }
  AffineBatch bin(Bin f, const AffineBatch& o) const {
    if (o.len() != len()) throw Error(JJ_ERR_INVALID, "length mismatch");
    AffineBatch r(*c_, std::vector<Bytes64>(len()));
    c_->check(f(c_->raw(), len(), p_.data(), o.p_.data(), r.p_.data()));
    return r;
  }
  std::vector<uint8_t> pred(Pred f) const { std::vector<uint8_t> out(len()); c_->check(f(c_->raw(), len(), p_.data(), out.data())); return out; }
  const Context* c_;
  std::vector<Bytes64> p_;
};

// `AffineNielsPoint * Fr` / multiply_bits for ONE base point (lib.rs:272-310): the window table lives on the device
class FixedBase {
 public:
  FixedBase(const Context& c, const Bytes64& base) : c_(&c) { c.check(jj_fixedbase_table_create(c.raw(), base.data(), 0, &t_)); }
  ~FixedBase() { if (t_) jj_fixedbase_table_destroy(c_->raw(), t_); }
  FixedBase(const FixedBase&) = delete;
  FixedBase& operator=(const FixedBase&) = delete;
  AffineBatch multiply_bits(const std::vector<Bytes32>& by) const {
    std::vector<Bytes64> out(by.size());
    c_->check(jj_fixedbase_mul(c_->raw(), t_, by.size(), by.data(), out.data()));
    return AffineBatch(*c_, std::move(out));
  }
  AffineBatch operator*(const FrBatch& k) const { return multiply_bits(k.to_bytes()); }
  const jj_table* raw() const { return t_; }
  const Context& context() const { return *c_; }

 private:
  const Context* c_;
  jj_table* t_ = nullptr;
};
// sum_j bases[j] * scalars[j][i] for every i: sums of AffineNielsPoint::multiply_bits over several fixed generators
inline AffineBatch fixedbase_multi_mul(const std::vector<const FixedBase*>& bases, const std::vector<std::vector<Bytes32>>& scalars) {
  if (bases.empty() || bases.size() != scalars.size()) throw Error(JJ_ERR_INVALID, "bases / scalars mismatch");
  const size_t n = scalars[0].size();
  std::vector<const jj_table*> tabs;
  std::vector<Bytes32> flat;
  flat.reserve(n * bases.size());
  for (size_t j = 0; j < bases.size(); j++) {
    if (scalars[j].size() != n) throw Error(JJ_ERR_INVALID, "length mismatch");          // cf. lib.rs:841
    tabs.push_back(bases[j]->raw());
    flat.insert(flat.end(), scalars[j].begin(), scalars[j].end());
  }
  const Context& c = bases[0]->context();
  std::vector<Bytes64> out(n);
  c.check(jj_fixedbase_multi_mul(c.raw(), tabs.data(), (int)tabs.size(), n, flat.data(), out.data()));
  return AffineBatch(c, std::move(out));
}

// batch_normalize (lib.rs:1084-1107): (U,V,Z,T1,T2) canonical 160-byte records -> affine
inline AffineBatch batch_normalize(const Context& c, const std::vector<std::array<uint8_t, 160>>& ext) {
  std::vector<Bytes64> out(ext.size());
  c.check(jj_batch_normalize(c.raw(), ext.size(), ext.data(), out.data()));
  return AffineBatch(c, std::move(out));
}
// sum_i points[i] * scalars[i]   (iterator Sum of p * k, lib.rs:183-193 + 873-879)
inline Bytes64 msm(const Context& c, const AffineBatch& points, const FrBatch& scalars) {
  if (points.len() != scalars.len()) throw Error(JJ_ERR_INVALID, "length mismatch");
  Bytes64 out;
  c.check(jj_msm(c.raw(), points.len(), scalars.to_bytes().data(), points.coords().data(), out.data()));
  return out;
}

// The same sum with the host tail of one MSM overlapping the kernels of the next (jj_msm_begin / jj_msm_finish): the job owns copies
// of its inputs until it is finished
struct AllRanks { bool by_windows = false; };      // MsmJob over every rank of the communicator lent with set_comm (jj_msm_allgather_begin)
class MsmJob {
 public:
  MsmJob(const Context& c, const AffineBatch& points, const FrBatch& scalars) : c_(&c), s_(scalars.to_bytes()), p_(points.coords()) {
    if (p_.size() != s_.size()) throw Error(JJ_ERR_INVALID, "length mismatch");
    c.check(jj_msm_begin(c.raw(), p_.size(), s_.data(), p_.data(), &job_));
  }
  // this rank's terms of a sum over all ranks: window sums, ncclAllGather and the fold of the gathered records queued behind one another;
  // every rank constructs the same jobs in the same order
  MsmJob(const Context& c, const AffineBatch& my_points, const FrBatch& my_scalars, AllRanks how) : c_(&c), s_(my_scalars.to_bytes()), p_(my_points.coords()) {
    if (p_.size() != s_.size()) throw Error(JJ_ERR_INVALID, "length mismatch");
    c.check(jj_msm_allgather_begin(c.raw(), p_.size(), s_.data(), p_.data(), how.by_windows ? 1 : 0, &job_));
  }
  MsmJob(const MsmJob&) = delete;
  MsmJob& operator=(const MsmJob&) = delete;
  ~MsmJob() { if (job_) { Bytes64 sink; (void)jj_msm_finish(job_, sink.data()); } }
  Bytes64 finish() {
    if (!job_) throw Error(JJ_ERR_INVALID, "MSM job already finished");
    Bytes64 out;
    jj_msm_job* j = job_;
    job_ = nullptr;
    c_->check(jj_msm_finish(j, out.data()));
    return out;
  }

 private:
  const Context* c_;
  std::vector<Bytes32> s_;
  std::vector<Bytes64> p_;
  jj_msm_job* job_ = nullptr;
};
// MSM cut into parts (jj_msm_partial / jj_msm_combine): part g of G by windows (every part sees all terms) or all windows of a slice
// of the terms; the records are combined in one host tail
using MsmRecord = std::array<uint8_t, JJ_MSM_PARTIAL_BYTES>;
inline MsmRecord msm_partial(const Context& c, const AffineBatch& points, const FrBatch& scalars, int part_index = 0, int part_count = 1) {
  if (points.len() != scalars.len()) throw Error(JJ_ERR_INVALID, "length mismatch");
  MsmRecord rec;
  c.check(jj_msm_partial(c.raw(), points.len(), scalars.to_bytes().data(), points.coords().data(), part_index, part_count, rec.data()));
  return rec;
}
inline Bytes64 msm_combine(const std::vector<MsmRecord>& records) {
  Bytes64 out;
  const int rc = jj_msm_combine(records.size(), records.empty() ? nullptr : records.data(), out.data());
  if (rc) throw Error(rc, "jj_msm_combine: damaged or mismatched records");
  return out;
}
// Round 4: page-locked host memory for batches that come back call after call (jj_host_alloc); the raw entry points then copy straight
// from and to it.  (std::vector arguments are pageable: the library moves them through its own page-locked staging buffers.)
class HostBuffer {
 public:
  explicit HostBuffer(size_t bytes) : n_(bytes) { const int rc = jj_host_alloc(bytes, &p_); if (rc) throw Error(rc, "jj_host_alloc failed"); }
  ~HostBuffer() { (void)jj_host_free(p_); }
  HostBuffer(const HostBuffer&) = delete;
  HostBuffer& operator=(const HostBuffer&) = delete;
  uint8_t* data() { return static_cast<uint8_t*>(p_); }
  const uint8_t* data() const { return static_cast<const uint8_t*>(p_); }
  size_t size() const { return n_; }
 private:
  void* p_ = nullptr;
  size_t n_ = 0;
};
// points[i] * scalars[i] on raw wire-format buffers (any kind of host memory, e.g. HostBuffers kept by the caller): out = n x 64 bytes
inline void multiply_raw(const Context& c, size_t n, const uint8_t* scalars32, const uint8_t* points64, uint8_t* out64) {
  c.check(jj_varbase_mul(c.raw(), n, scalars32, points64, out64));
}
// One process per GPU: the Sum over the terms of every rank of an RCCL communicator the application created (jj_ctx_set_comm +
// jj_msm_allgather; examples/msm_rccl.cpp).  all_gather_fn: address of that RCCL library's ncclAllGather, or nullptr (looked up).
inline void set_comm(const Context& c, void* nccl_comm, int rank, int nranks, void* all_gather_fn = nullptr) {
  c.check(jj_ctx_set_comm(c.raw(), nccl_comm, rank, nranks, all_gather_fn));
}
inline Bytes64 msm_all_ranks(const Context& c, const AffineBatch& my_points, const FrBatch& my_scalars, bool by_windows = false) {
  if (my_points.len() != my_scalars.len()) throw Error(JJ_ERR_INVALID, "length mismatch");
  Bytes64 out;
  c.check(jj_msm_allgather(c.raw(), my_points.len(), my_scalars.to_bytes().data(), my_points.coords().data(), by_windows ? 1 : 0, out.data()));
  return out;
}
// `ExtendedPoint * Fr` as operator* computes it: the reference's constant-time discipline (lib.rs:334-343, 357-379), no scalar-dependent
// address or branch (jj_varbase_mul; multiply_ct is the name rounds 3-4 gave it)
inline AffineBatch multiply_ct(const Context& c, const AffineBatch& points, const FrBatch& scalars) {
  if (points.len() != scalars.len()) throw Error(JJ_ERR_INVALID, "length mismatch");
  std::vector<Bytes64> out(points.len());
  c.check(jj_varbase_mul_ct(c.raw(), points.len(), scalars.to_bytes().data(), points.coords().data(), out.data()));
  return AffineBatch(c, std::move(out));
}
// the same product by the variable-time ladder (per-lane window table in memory, digit-dependent addresses): for PUBLIC scalars only
inline AffineBatch multiply_vartime(const Context& c, const AffineBatch& points, const FrBatch& scalars) {
  if (points.len() != scalars.len()) throw Error(JJ_ERR_INVALID, "length mismatch");
  std::vector<Bytes64> out(points.len());
  c.check(jj_varbase_mul_vartime(c.raw(), points.len(), scalars.to_bytes().data(), points.coords().data(), out.data()));
  return AffineBatch(c, std::move(out));
}
// several fixed bases with short scalars through one LDS table set, one pass (sums of multiply_bits, lib.rs:297-301)
class CompositeBase {
 public:
  CompositeBase(const Context& c, const std::vector<Bytes64>& bases, const std::vector<int>& scalar_bits) : c_(&c), nb_(bases.size()) {
    if (bases.empty() || bases.size() != scalar_bits.size()) throw Error(JJ_ERR_INVALID, "one bit length per base");
    c.check(jj_fixedbase_composite_create(c.raw(), (int)bases.size(), bases.data(), scalar_bits.data(), &t_));
  }
  ~CompositeBase() { if (t_) jj_fixedbase_table_destroy(c_->raw(), t_); }
  CompositeBase(const CompositeBase&) = delete;
  CompositeBase& operator=(const CompositeBase&) = delete;
  // scalars[b][i]: only the low scalar_bits[b] bits are used
  AffineBatch multiply_bits(const std::vector<std::vector<Bytes32>>& scalars) const {
    if (scalars.size() != nb_) throw Error(JJ_ERR_INVALID, "bases / scalars mismatch");
    const size_t n = scalars[0].size();
    std::vector<Bytes32> flat;
    for (const auto& v : scalars) { if (v.size() != n) throw Error(JJ_ERR_INVALID, "length mismatch"); flat.insert(flat.end(), v.begin(), v.end()); }
    std::vector<Bytes64> out(n);
    c_->check(jj_fixedbase_composite_mul(c_->raw(), t_, n, flat.data(), out.data()));
    return AffineBatch(*c_, std::move(out));
  }

 private:
  const Context* c_;
  size_t nb_;
  jj_table* t_ = nullptr;
};

}  // namespace jubjub
