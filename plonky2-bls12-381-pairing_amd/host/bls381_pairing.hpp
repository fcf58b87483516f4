// bls381_pairing.hpp -- header-only C++ mirror of the reference's native pairing interface on top of the C ABI
// (include/bls381_hip.h).  The reference is Rust; no Rust toolchain exists in the build image, so the host side
// above the C ABI is C++ with the reference's names, argument meaning and error behaviour:
//
//   optimized_miller_loop(P: G1Projective, Q: G2Projective) -> Fq12          src/miller_loop_native_optimized.rs:81
//   multi_miller_loop(terms: &[(&G1Affine, &G2Affine)]) -> MillerLoopResult   src/miller_loop_native.rs:154
//   struct MillerLoopResult(pub(crate) Fq12), Default = one                   src/miller_loop_native.rs:7-14
//   MillerLoopResult::final_exponentiation -> Gt                              src/fields_as_trees/miller_loop.rs:166-178
//
// Types are plain limb structs in arkworks' in-memory layout (Fq = 6 x u64 little-endian Montgomery limbs),
// so a Rust caller's values map onto them field for field.
//
// Everything executes on the GPU through libbls381_hip.so; there is no CPU path -- constructing a Backend
// without a HIP device throws.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bls381_hip.h"

namespace bls381 {

struct Fq { std::array<uint64_t, 6> limbs{}; };                  // ark_bls12_381::Fq (Montgomery form)
struct Fq2 { Fq c0, c1; };
struct Fq6 { Fq2 c0, c1, c2; };
struct Fq12 { Fq6 c0, c1; };
struct G1Projective { Fq x, y, z; };                             // Jacobian, as arkworks stores it
struct G2Projective { Fq2 x, y, z; };
struct G1Affine { Fq x, y; bool infinity = false; };
struct G2Affine { Fq2 x, y; bool infinity = false; };
static_assert(sizeof(Fq12) == 576 && sizeof(G1Projective) == 144 && sizeof(G2Projective) == 288, "ABI layout");

struct PairingError : std::runtime_error { using std::runtime_error::runtime_error; };

/// One GPU context (`bls381_ctx`).  Thread-compatible: use one Backend per host thread or serialise calls.
class Backend {
 public:
  explicit Backend(int device_id = 0) {
    const int rc = bls381_create(&ctx_, device_id);
    if (rc != BLS381_OK) throw PairingError(std::string("bls381_create: ") + bls381_strerror(rc));
  }
  ~Backend() { bls381_destroy(ctx_); }
  Backend(const Backend&) = delete;
  Backend& operator=(const Backend&) = delete;
  bls381_ctx* raw() const { return ctx_; }
  void check(int rc, const char* what) const {
    if (rc != BLS381_OK)
      throw PairingError(std::string(what) + ": " + bls381_strerror(rc) + " (hip " + std::to_string(bls381_last_hip_error(ctx_)) + ")");
  }

 private:
  bls381_ctx* ctx_ = nullptr;
};

/// src/miller_loop_native.rs:7-14
struct MillerLoopResult {
  Fq12 value;
  /// `impl Default`: Fq12::one() -- fetched from the device so the encoding matches the context's I/O form
  static MillerLoopResult one(const Backend& be) {
    MillerLoopResult r;
    be.check(bls381_multi_miller_literal(be.raw(), reinterpret_cast<uint64_t*>(&r.value), 1), "bls381_multi_miller_literal");
    return r;
  }
  /// src/fields_as_trees/miller_loop.rs:166-178
  Fq12 final_exponentiation(const Backend& be) const {
    Fq12 out;
    be.check(bls381_final_exp_batch(be.raw(), reinterpret_cast<const uint64_t*>(&value), reinterpret_cast<uint64_t*>(&out), nullptr, 1),
             "bls381_final_exp_batch");
    return out;
  }
};

/// Batched optimized_miller_loop: element i of the result is optimized_miller_loop(pairs[i].first, pairs[i].second).
/// Throws where the reference panics (f_den == 0, opt.rs:102).
inline std::vector<Fq12> optimized_miller_loop_batch(const Backend& be, const std::vector<std::pair<G1Projective, G2Projective>>& pairs) {
  const size_t n = pairs.size();
  std::vector<G1Projective> p(n);
  std::vector<G2Projective> q(n);
  for (size_t i = 0; i < n; ++i) { p[i] = pairs[i].first; q[i] = pairs[i].second; }
  std::vector<Fq12> out(n);
  std::vector<uint8_t> flags(n);
  be.check(bls381_miller_ref_opt_batch(be.raw(), reinterpret_cast<const uint64_t*>(p.data()), reinterpret_cast<const uint64_t*>(q.data()),
                                       reinterpret_cast<uint64_t*>(out.data()), flags.data(), n),
           "bls381_miller_ref_opt_batch");
  for (size_t i = 0; i < n; ++i)
    if (flags[i] & BLS381_FLAG_DEN_ZERO) throw PairingError("optimized_miller_loop: f_den == 0 (the reference panics here, opt.rs:102)");
  return out;
}

/// src/miller_loop_native_optimized.rs:81
inline Fq12 optimized_miller_loop(const Backend& be, const G1Projective& P, const G2Projective& Q) {
  return optimized_miller_loop_batch(be, {{P, Q}})[0];
}

namespace detail {
inline void pack_terms(const std::vector<std::pair<const G1Affine*, const G2Affine*>>& terms, std::vector<uint64_t>& g1, std::vector<uint64_t>& g2) {
  for (const auto& t : terms) {
    const size_t o1 = g1.size(), o2 = g2.size();
    g1.resize(o1 + 12, 0);
    g2.resize(o2 + 24, 0);
    if (!t.first->infinity) {
      for (int i = 0; i < 6; ++i) { g1[o1 + i] = t.first->x.limbs[i]; g1[o1 + 6 + i] = t.first->y.limbs[i]; }
    }
    if (!t.second->infinity) {
      const Fq* c[4] = {&t.second->x.c0, &t.second->x.c1, &t.second->y.c0, &t.second->y.c1};
      for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 6; ++i) g2[o2 + 6 * k + i] = c[k]->limbs[i];
    }
  }
}
}  // namespace detail

/// src/miller_loop_native.rs:154 with the line evaluation live (the reference's body is commented out and returns
/// one; `MillerLoopResult::one` reproduces that literal behaviour).  Any number of terms, like the reference's slice.
inline MillerLoopResult multi_miller_loop(const Backend& be, const std::vector<std::pair<const G1Affine*, const G2Affine*>>& terms) {
  if (terms.empty()) return MillerLoopResult::one(be);
  std::vector<uint64_t> g1, g2;
  detail::pack_terms(terms, g1, g2);
  MillerLoopResult r;
  be.check(bls381_multi_miller_batch(be.raw(), g1.data(), g2.data(), terms.size(), reinterpret_cast<uint64_t*>(&r.value), nullptr, 1),
           "bls381_multi_miller_batch");
  return r;
}

/// Product of pairings e(P_1, Q_1) ... e(P_k, Q_k) for n independent k-tuples (Groth16 shape: k = 3).
inline std::vector<Fq12> multi_pairing_batch(const Backend& be, const std::vector<uint64_t>& g1_aff, const std::vector<uint64_t>& g2_aff, size_t k) {
  if (k == 0 || g1_aff.size() % (12 * k) != 0) throw PairingError("multi_pairing_batch: ragged input");
  const size_t n = g1_aff.size() / (12 * k);
  if (g2_aff.size() != n * 24 * k) throw PairingError("multi_pairing_batch: ragged input");
  std::vector<Fq12> out(n);
  be.check(bls381_multi_pairing_batch(be.raw(), g1_aff.data(), g2_aff.data(), k, reinterpret_cast<uint64_t*>(out.data()), nullptr, n),
           "bls381_multi_pairing_batch");
  return out;
}

}  // namespace bls381
