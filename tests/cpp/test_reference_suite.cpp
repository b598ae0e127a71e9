// Replays the reference crate's own tests (zkcrypto/jubjub src/lib.rs:1456-1935, src/fr.rs:787-1244,
// tests/fq_blackbox.rs, tests/fr_blackbox.rs) through the C++ host mirror (include/jubjub_hip.hpp) on the GPU.
// Golden data comes from a text file written by tests/test_cpp_host.py out of tests/golden/reference_vectors.json.
// Usage: test_reference_suite <vectors.txt>      (exit code 0 = all passed)
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

#include "jubjub_hip.hpp"

using namespace jubjub;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("  FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

static std::map<std::string, std::vector<std::vector<uint8_t>>> G;
static std::vector<uint8_t> unhex(const std::string& h) {
  std::vector<uint8_t> o(h.size() / 2);
  for (size_t i = 0; i < o.size(); i++) o[i] = (uint8_t)std::stoi(h.substr(2 * i, 2), nullptr, 16);
  return o;
}
static void load(const char* path) {
  std::ifstream f(path);
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream ss(line);
    std::string key, hex;
    ss >> key;
    while (ss >> hex) G[key].push_back(unhex(hex));
  }
}
template <size_t N>
static std::vector<std::array<uint8_t, N>> arr(const std::string& key) {
  std::vector<std::array<uint8_t, N>> o;
  for (auto& v : G.at(key)) { std::array<uint8_t, N> a; if (v.size() != N) { std::printf("bad size for %s\n", key.c_str()); std::exit(2); } std::memcpy(a.data(), v.data(), N); o.push_back(a); }
  return o;
}

// tests/common.rs:7-9 : XorShiftRng::from_seed([0..=15]) ; rand_xorshift 0.3 next_u32
struct XorShift {
  uint32_t x, y, z, w;
  XorShift() { uint8_t s[16]; for (int i = 0; i < 16; i++) s[i] = (uint8_t)i; auto rd = [&](int o) { return (uint32_t)s[o] | (uint32_t)s[o + 1] << 8 | (uint32_t)s[o + 2] << 16 | (uint32_t)s[o + 3] << 24; };
             x = rd(0); y = rd(4); z = rd(8); w = rd(12); }
  uint32_t next_u32() { uint32_t t = x ^ (x << 11); x = y; y = z; z = w; w = w ^ (w >> 19) ^ (t ^ (t >> 8)); return w; }
  void fill(uint8_t* p, size_t n) { for (size_t i = 0; i < n; i += 4) { uint32_t v = next_u32(); for (size_t b = 0; b < 4 && i + b < n; b++) p[i + b] = (uint8_t)(v >> (8 * b)); } }
};
template <class F>
static F random_elems(const Context& c, XorShift& rng, size_t n) {   // tests/common.rs:15-29: 64 random bytes -> from_bytes_wide
  std::vector<Bytes64> raw(n);
  for (auto& r : raw) rng.fill(r.data(), 64);
  return F::from_bytes_wide(c, raw);
}

// src/lib.rs:1807-1890
static void test_serialization_consistency(const Context& c) {
  std::puts("test_serialization_consistency");
  const auto expected = arr<32>("serialization_16");
  const AffineBatch gen = AffineBatch::generator(c, 1).mul_by_cofactor();
  AffineBatch p = gen;
  const auto batched = AffineBatch::batch_from_bytes(c, expected);
  CHECK(batched.all_some());
  for (size_t i = 0; i < expected.size(); i++) {
    CHECK(p.is_on_curve()[0] == 1);
    const auto ser = p.to_bytes();
    const auto de = AffineBatch::from_bytes(c, ser);
    CHECK(de.all_some() && de.value == p);
    CHECK(batched.value.coords()[i] == p.coords()[0]);
    CHECK(ser[0] == expected[i]);
    p = p + gen;
  }
}
// src/lib.rs:1893-1935
static void test_zip_216(const Context& c) {
  std::puts("test_zip_216");
  for (auto& b : arr<32>("zip216_noncanonical")) {
    CHECK(AffineBatch::from_bytes(c, {b}).is_some[0] == 0);
    Bytes32 enc = b; enc[31] &= 0x7f;
    CHECK(AffineBatch::from_bytes(c, {enc}).is_some[0] == 1);
    const auto parsed = AffineBatch::from_bytes_pre_zip216_compatibility(c, {b});
    CHECK(parsed.is_some[0] == 1);
    Bytes32 re = parsed.value.to_bytes()[0];
    CHECK(re != b);
    re[31] |= 0x80;
    CHECK(re == b);
  }
}
// src/lib.rs:1679-1696, 1730-1754
static void find_eight_torsion(const Context& c) {
  std::puts("find_eight_torsion / test_small_order / test_is_identity");
  const auto r_bytes = arr<32>("FR_MODULUS_BYTES");
  const AffineBatch g0 = AffineBatch::generator(c, 1);
  CHECK(g0.is_small_order()[0] == 0);
  const AffineBatch g = g0.multiply_bits(r_bytes);
  CHECK(g.is_small_order()[0] == 1);
  AffineBatch cur = g;
  const auto tors = arr<64>("EIGHT_TORSION");
  for (auto& t : tors) { CHECK(cur.coords()[0] == t); cur = cur + g; }
  const AffineBatch T(c, tors);
  for (auto b : T.is_small_order()) CHECK(b == 1);
  for (auto b : T.mul_by_cofactor().is_identity()) CHECK(b == 1);
  const auto tf = T.is_torsion_free();
  for (size_t i = 0; i < 8; i++) CHECK(tf[i] == (i == 7));
  CHECK(g0.mul_by_cofactor().is_torsion_free()[0] == 1);     // find_curve_generator, lib.rs:1719
  CHECK(g0.is_prime_order()[0] == 0 && g0.mul_by_cofactor().is_prime_order()[0] == 1);
}
// src/lib.rs:1504-1527, 1756-1804
static void test_mul_consistency(const Context& c) {
  std::puts("test_assoc / test_mul_consistency");
  const AffineBatch p = AffineBatch(c, arr<64>("TEST_POINT")).mul_by_cofactor();
  CHECK(p.is_on_curve()[0] == 1);
  const FrBatch k1000 = FrBatch::from_u64(c, {1000}), k3938 = FrBatch::from_u64(c, {3938});
  CHECK((p * k1000) * k3938 == p * (k1000 * k3938));
  const FrBatch a(c, {arr<32>("fr_mul_a")[0]}), b(c, {arr<32>("fr_mul_b")[0]}), cc(c, {arr<32>("fr_mul_c")[0]});
  CHECK(a * b == cc);
  CHECK(p * cc == (p * a) * b);
  const FixedBase fb(c, p.coords()[0]);                        // AffineNielsPoint path
  CHECK(fb * cc == (p * a) * b);
  CHECK(p * cc == (fb * a) * b);
}
// src/lib.rs:1529-1575 (through to_bytes-level observables)
static void test_doubling_chain(const Context& c) {
  std::puts("test_batch_normalize (doubling chain)");
  AffineBatch p = AffineBatch(c, arr<64>("TEST_POINT")).mul_by_cofactor();
  std::vector<Bytes64> chain;
  for (int i = 0; i < 10; i++) { chain.push_back(p.coords()[0]); p = p.double_(); }
  const AffineBatch all(c, chain);
  for (auto b : all.is_on_curve()) CHECK(b == 1);
  const AffineBatch k2 = AffineBatch(c, std::vector<Bytes64>(chain.begin(), chain.end() - 1)) * FrBatch::from_u64(c, std::vector<uint64_t>(9, 2));
  for (int i = 0; i < 9; i++) CHECK(k2.coords()[i] == chain[i + 1]);
}
// src/fr.rs:855-1244 (the vector-valued ones)
static void test_fr_vectors(const Context& c) {
  std::puts("fr: test_to_bytes / test_from_bytes / test_addition / test_negation / test_inversion / test_sqrt");
  const Bytes32 neg_one = arr<32>("fr_neg_one")[0];
  const FrBatch one = FrBatch::from_u64(c, {1}), zero = FrBatch::from_u64(c, {0});
  CHECK((-one).to_bytes()[0] == neg_one);
  for (auto& bad : arr<32>("fr_from_bytes_invalid")) CHECK(FrBatch::from_bytes(c, {bad}).is_some[0] == 0);
  CHECK(FrBatch::from_bytes(c, {neg_one}).is_some[0] == 1);
  const FrBatch largest(c, {neg_one});                         // LARGEST = r - 1 (fr.rs:1045-1050)
  CHECK((largest + one) == zero);
  CHECK((-largest) == one && (-zero) == zero);
  CHECK(zero.invert().is_some[0] == 0);
  CHECK(one.invert().value == one && largest.invert().value == largest);
  // test_sqrt (fr.rs:1204-1227): exactly 47 non-residues among r-2 ... r-101
  // (the start value is given by its Montgomery limbs r-2 in the reference; `square -= Fr::one()` each round)
  std::vector<uint64_t> ks(100); for (int i = 0; i < 100; i++) ks[i] = i;
  const FrBatch squares = FrBatch(c, std::vector<Bytes32>(100, arr<32>("fr_sqrt_start")[0])) - FrBatch::from_u64(c, ks);
  const auto roots = squares.sqrt();
  int none = 0; for (auto s : roots.is_some) none += !s;
  CHECK(none == 47);
  const FrBatch back = roots.value * roots.value;
  for (int i = 0; i < 100; i++) if (roots.is_some[i]) CHECK(back.to_bytes()[i] == squares.to_bytes()[i]);
  CHECK(FrBatch::from_bytes_wide(c, {arr<64>("fr_wide_max_in")[0]}).to_bytes()[0] == arr<32>("fr_wide_max_out")[0]);
}
// tests/fq_blackbox.rs / tests/fr_blackbox.rs: the 11 properties x NUM_BLACK_BOX_CHECKS = 2000
template <class F>
static void blackbox(const Context& c, const char* name) {
  std::printf("%s_blackbox (11 properties x 2000)\n", name);
  const size_t N = 2000;
  const F zero = F::from_u64(c, std::vector<uint64_t>(N, 0)), one = F::from_u64(c, std::vector<uint64_t>(N, 1));
  { XorShift r; const F a = random_elems<F>(c, r, N); const auto rt = F::from_bytes(c, a.to_bytes()); CHECK(rt.all_some() && rt.value == a); }
  { XorShift r; const F a = random_elems<F>(c, r, N), b = random_elems<F>(c, r, N), d = random_elems<F>(c, r, N);
    CHECK((a + b) + d == a + (b + d)); CHECK((a * b) * d == a * (b * d)); CHECK(a + b == b + a); CHECK(a * b == b * a); }
  { XorShift r; const F a = random_elems<F>(c, r, N);
    CHECK(a + zero == a && zero + a == a); CHECK(a - zero == a && zero - (-a) == a); CHECK(a + (-a) == zero && (-a) + a == zero);
    CHECK(a * one == a && one * a == a); CHECK(a * zero == zero && zero * a == zero);
    const auto inv = a.invert(); CHECK(inv.all_some()); CHECK(a * inv.value == one && inv.value * a == one); }
}

int main(int argc, char** argv) {
  if (argc < 2) { std::puts("usage: test_reference_suite vectors.txt"); return 2; }
  load(argv[1]);
  try {
    Context c(0);
    test_serialization_consistency(c);
    test_zip_216(c);
    find_eight_torsion(c);
    test_mul_consistency(c);
    test_doubling_chain(c);
    test_fr_vectors(c);
    blackbox<FqBatch>(c, "fq");
    blackbox<FrBatch>(c, "fr");
    // Group::random / Field::random / to_le_bits through the mirror (lib.rs:1244-1267, fr.rs:684-688, 746-773): sampled points are
    // on the curve, in the prime-order subgroup when asked, never the identity; bits recompose the canonical bytes; and
    // (k * P) over random inputs agrees between the ladder and the sum of its bit-decomposed doublings for k = 2^j.
    {
      std::puts("random / to_le_bits");
      const AffineBatch p = AffineBatch::random(c, 300, 0x4a55, 5, true), q = AffineBatch::random(c, 300, 0x4a55, 5, false);
      for (auto b : p.is_prime_order()) CHECK(b == 1);
      for (auto b : q.is_on_curve()) CHECK(b == 1);
      for (auto b : q.is_identity()) CHECK(b == 0);
      const FrBatch k = FrBatch::random(c, 300, 0x4a55, 9);
      const auto bits = k.to_le_bits();
      for (size_t i = 0; i < k.len(); i++) {
        Bytes32 re{}; for (int b = 0; b < 256; b++) re[b >> 3] |= (uint8_t)(bits[i][b] << (b & 7));
        CHECK(re == k.to_bytes()[i]);
      }
      CHECK(FrBatch::from_bytes(c, k.to_bytes()).all_some());
      CHECK(AffineBatch::random(c, 50, 0x4a55, 105, true) == AffineBatch(c, std::vector<Bytes64>(p.coords().begin() + 100, p.coords().begin() + 150)));
    }
    // round-3 entry points through the mirror: the constant-time ladder, the MSM in two halves / in parts, several short-scalar bases
    {
      std::puts("multiply_ct / MsmJob / msm_partial + msm_combine / CompositeBase");
      const AffineBatch p = AffineBatch::random(c, 500, 0x77, 0, false);
      const FrBatch k = FrBatch::random(c, 500, 0x78, 0);
      CHECK(multiply_ct(c, p, k) == p * k);
      const Bytes64 want = msm(c, p, k);
      MsmJob j1(c, p, k), j2(c, p, k);
      CHECK(j2.finish() == want);
      CHECK(j1.finish() == want);
      std::vector<MsmRecord> recs;
      for (int g = 0; g < 3; g++) recs.push_back(msm_partial(c, p, k, g, 3));
      CHECK(msm_combine(recs) == want);
      CHECK((p * k).sum() == want);
      std::vector<Bytes64> bases(p.coords().begin(), p.coords().begin() + 3);
      CompositeBase cb(c, bases, {64, 64, 64});
      std::vector<std::vector<Bytes32>> ks(3, std::vector<Bytes32>(200));
      for (int b = 0; b < 3; b++) for (size_t i = 0; i < 200; i++) { ks[b][i] = Bytes32{}; for (int q = 0; q < 8; q++) ks[b][i][q] = k.to_bytes()[(b * 100 + i) % 500][q]; }
      FixedBase f0(c, bases[0]), f1(c, bases[1]), f2(c, bases[2]);
      CHECK(cb.multiply_bits(ks) == fixedbase_multi_mul({&f0, &f1, &f2}, ks));
    }
    // round-4 entry points through the mirror: page-locked host buffers on the chunked copy / compute pipeline (2^19 + 5 units: pipelined),
    // against the same batch in std::vectors (pageable: the bounce path)
    {
      std::puts("HostBuffer + multiply_raw (page-locked, pipelined) against operator* (pageable vectors, bounce path)");
      const size_t n = ((size_t)1 << 19) + 5;
      const AffineBatch p = AffineBatch::random(c, n, 0x91, 0, false);
      const FrBatch k = FrBatch::random(c, n, 0x92, 0);
      HostBuffer hs(32 * n), hp(64 * n), ho(64 * n);
      std::memcpy(hs.data(), k.to_bytes().data(), 32 * n);
      std::memcpy(hp.data(), p.coords().data(), 64 * n);
      multiply_raw(c, n, hs.data(), hp.data(), ho.data());
      const AffineBatch want = p * k;
      CHECK(std::memcmp(ho.data(), want.coords().data(), 64 * n) == 0);
      bool refused = false;                                   // no communicator lent: the multi-rank sum is refused, not computed on one rank
      try { (void)msm_all_ranks(c, p, k); } catch (const Error&) { refused = true; }
      CHECK(refused);
    }
    // error behaviour: length mismatch is rejected like the assert at src/lib.rs:841
    bool threw = false;
    try { AffineBatch::generator(c, 2) * FrBatch::from_u64(c, {1}); } catch (const Error&) { threw = true; }
    CHECK(threw);
  } catch (const Error& e) { std::printf("ERROR %d: %s\n", e.status, e.what()); return 3; }
  std::printf("%s (%d failures)\n", failures ? "FAILED" : "ALL PASSED", failures);
  return failures ? 1 : 0;
}
