// Fp layer of the MI355X pairing engine: GF(p) for the BLS12-381 base field on gfx950.
//
// Replaces (behind the C ABI in include/bls381_hip.h) the Fp arithmetic that the reference takes
// from ark-ff `Fq` [third party] and restates in src/fields/bls12_381base.rs:174-268.
//
// Representation (chosen from the instruction-rate measurements in profiles/r01_valu_rates_mi355x.txt:
// on gfx950 a carry-writing add costs as much as a 32x32+64 multiply-add, so carries are designed out):
//   * 14 limbs x 28 bits in 32-bit registers ("U28"), value = sum l[i] << 28 i.  Limbs may exceed 28 bits
//     between multiplications (additions never propagate carries).
//   * Montgomery form with R = 2^392: 11 bits above p, so products of unreduced operands (values up to
//     tens of p) still reduce to < 2p with no conditional subtraction anywhere on the hot path.
//   * One multiplication = finely-integrated product scanning: every column is a chain of
//     v_mad_u64_u32 into one 64-bit accumulator that cannot overflow (bounds below), then one
//     v_mul_lo/v_and (Montgomery quotient digit) and one 64-bit shift.  No carry flag is ever used.
//   * `mont2` accumulates TWO products before the single reduction (a*b + c*d): an Fp2 multiplication
//     is two of those -- schoolbook with lazy reduction -- with no Karatsuba pre/post additions.
//
// Bounds are part of the type: Fp<L, V> promises  limb < L * 2^28 (+ a few units)  and  value < V * p.
// Every operation static_asserts that its accumulators cannot overflow, so an unsafe lazy-reduction
// chain is a compile error, not a wrong answer.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BLS_HD __host__ __device__ __forceinline__
#define BLS_HD_NOINLINE inline __host__ __device__ __noinline__
#else
#define BLS_HD inline __attribute__((always_inline))
#define BLS_HD_NOINLINE inline __attribute__((noinline))
#endif
#define BLS_CONST static constexpr

#include "fp_consts.h"

namespace bls {

constexpr int NL = 14;                 // limbs
constexpr int LW = 28;                 // limb width
constexpr uint32_t LMASK = (1u << LW) - 1u;

// Limit for the product terms of one accumulator column:
//   14 * (sum of La*Lb over the fused products) * 2^56  +  14 * 2^56 (m * p)  +  carry-in (< 2^37)  <  2^64
// <=> sum(La*Lb) + 1 <= 18 (252/256 used; the slack covers the "+ few units" of almost-normalised limbs).
constexpr int MONT_LL_MAX = 17;
// Value bounds carry 1% slack: "V" promises value < V * 1.01 p.  A product of bounds Va, Vb reduces to
//   < ((sum Va*Vb) * 1.01^2 * p / 2^392 + 1) p   with  p / 2^392 < 1 / 2520.3,
// which stays < 2.02 p (i.e. V = 2) while sum(Va*Vb) <= 2500.  The slack absorbs the quotient estimate of
// weak_reduce() below.
constexpr int MONT_VV_MAX = 2500;

#ifdef BLS_BOUND_CHECK
// Host-only instrumentation (tests/hostsim): verifies at run time that the compile-time bounds hold.
void bound_check_fail(const char* what);
void bound_check(const uint32_t* l, int L, int V);
extern long g_op_count[8];   // 0 mul, 1 mul2, 2 sqr, 3 weak_reduce, 4 norm, 5 add/sub/neg/dbl, 6 dot (lazy sums)
#define BLS_COUNT(k) (++::bls::g_op_count[k])
#else
#define BLS_COUNT(k) ((void)0)
#endif

template <int L, int V>
struct Fp {
  static constexpr int kL = L, kV = V;
  uint32_t l[NL];

  // widening conversion: a tighter bound is always a valid looser bound
  template <int L2, int V2>
  BLS_HD Fp(const Fp<L2, V2>& o) {
    static_assert(L2 <= L && V2 <= V, "cannot narrow Fp bounds implicitly");
#pragma unroll
    for (int i = 0; i < NL; ++i) l[i] = o.l[i];
  }
  Fp() = default;
  BLS_HD void chk() const {
#ifdef BLS_BOUND_CHECK
    bound_check(l, L, V);
#endif
  }
};

using FpR = Fp<1, 2>;   // "reduced": what every Montgomery multiplication returns

// ------------------------------------------------------------------------------------------------
// Montgomery multiplication core on raw limb arrays.  N_PROD = 1: a*b ; N_PROD = 2: a*b + c*d.
// Fully unrolled: 196 MADs per product + 196 for the reduction.
template <int N_PROD>
BLS_HD void mont_core(uint32_t* __restrict__ out, const uint32_t* a, const uint32_t* b,
                      const uint32_t* c, const uint32_t* d) {
  uint32_t m[NL];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) {
      acc += (uint64_t)a[i] * b[k - i];
      if (N_PROD == 2) acc += (uint64_t)c[i] * d[k - i];
    }
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * BLS_P28[k - i];
    m[k] = ((uint32_t)acc * BLS_PINV28) & LMASK;
    acc += (uint64_t)m[k] * BLS_P28[0];
    acc >>= LW;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) {
      acc += (uint64_t)a[i] * b[k - i];
      if (N_PROD == 2) acc += (uint64_t)c[i] * d[k - i];
    }
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc += (uint64_t)m[i] * BLS_P28[k - i];
    out[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LW;
  }
  out[NL - 1] = (uint32_t)acc;
}

// Montgomery squaring core: a*a with the cross terms doubled (105 + 196 MADs instead of 392).
BLS_HD void mont_sqr_core(uint32_t* __restrict__ out, const uint32_t* a) {
  uint32_t m[NL], a2[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) a2[i] = a[i] << 1;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; ++k) {
    const int lo = (k < NL) ? 0 : k - NL + 1;
    const int hi = (k < NL) ? k : NL - 1;
#pragma unroll
    for (int i = lo; i <= hi; ++i) {
      const int j = k - i;
      if (i < j) acc += (uint64_t)a2[i] * a[j];
      else if (i == j) acc += (uint64_t)a[i] * a[i];
    }
    if (k < NL) {
#pragma unroll
      for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * BLS_P28[k - i];
      m[k] = ((uint32_t)acc * BLS_PINV28) & LMASK;
      acc += (uint64_t)m[k] * BLS_P28[0];
    } else {
#pragma unroll
      for (int i = k - NL + 1; i < NL; ++i) acc += (uint64_t)m[i] * BLS_P28[k - i];
      out[k - NL] = (uint32_t)acc & LMASK;
    }
    acc >>= LW;
  }
  out[NL - 1] = (uint32_t)acc;
}

#ifndef BLS_MONT_INLINE
#define BLS_MONT_INLINE 1
#endif
#if BLS_MONT_INLINE
#define BLS_MONT_FN BLS_HD
#else
#define BLS_MONT_FN BLS_HD_NOINLINE
#endif

struct Raw { uint32_t l[NL]; };
BLS_MONT_FN Raw mont1_raw(Raw a, Raw b) { Raw r; mont_core<1>(r.l, a.l, b.l, nullptr, nullptr); return r; }
BLS_MONT_FN Raw mont2_raw(Raw a, Raw b, Raw c, Raw d) { Raw r; mont_core<2>(r.l, a.l, b.l, c.l, d.l); return r; }
BLS_MONT_FN Raw montsqr_raw(Raw a) { Raw r; mont_sqr_core(r.l, a.l); return r; }

template <int L, int V> BLS_HD Raw raw_of(const Fp<L, V>& a) { Raw r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i];
  return r; }
BLS_HD FpR fp_of(const Raw& a) { FpR r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i];
  return r; }

// a * b / R  (mod p), result < 2p, limbs normalised
template <int La, int Va, int Lb, int Vb>
BLS_HD FpR mul(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) {
  static_assert(La * Lb <= MONT_LL_MAX, "mul: limb bounds overflow the 64-bit column accumulator");
  static_assert(Va * Vb <= MONT_VV_MAX, "mul: value bounds break the < 2p result guarantee");
  BLS_COUNT(0);
  FpR r = fp_of(mont1_raw(raw_of(a), raw_of(b)));
  r.chk();
  return r;
}

template <int La, int Va>
BLS_HD FpR sqr(const Fp<La, Va>& a) {
  // cross terms use the doubled operand (2 a_i) * a_j once instead of a_i a_j twice: same column total as a * a
  static_assert(La * La <= MONT_LL_MAX, "sqr: limb bounds overflow the 64-bit column accumulator");
  static_assert(Va * Va <= MONT_VV_MAX, "sqr: value bounds");
  BLS_COUNT(2);
  FpR r = fp_of(montsqr_raw(raw_of(a)));
  r.chk();
  return r;
}

// (a*b + c*d) / R  (mod p), result < 2p -- one reduction for two products
template <int La, int Va, int Lb, int Vb, int Lc, int Vc, int Ld, int Vd>
BLS_HD FpR mul2(const Fp<La, Va>& a, const Fp<Lb, Vb>& b, const Fp<Lc, Vc>& c, const Fp<Ld, Vd>& d) {
  static_assert(La * Lb + Lc * Ld <= MONT_LL_MAX, "mul2: limb bounds overflow the 64-bit column accumulator");
  static_assert(Va * Vb + Vc * Vd <= MONT_VV_MAX, "mul2: value bounds break the < 2p result guarantee");
  BLS_COUNT(1);
  FpR r = fp_of(mont2_raw(raw_of(a), raw_of(b), raw_of(c), raw_of(d)));
  r.chk();
  return r;
}

// ------------------------------------------------------------------------------------------------
// Lazy reduction: (a_1 b_1 + ... + a_N b_N) / R with ONE Montgomery reduction.  The products are accumulated by operand
// scanning into 27 column accumulators (64-bit each: the same headroom argument as above, 14 * sum(La Lb) * 2^56 + 14 * 2^56
// + carry < 2^64), so only ONE pair of operands has to be live at a time -- a column-scanning form of N products would
// need all 2 N operands in registers at once.  Every reduction saved is 196 multiply-adds and ~120 other instructions;
// at one wave per SIMD every instruction costs the same issue slot, so the sparse line multiplication written as six sums of
// three products (tower.hip.h: mul_by_014_lazy) comes out 13 % shorter than Karatsuba with a reduction per product.
BLS_HD void mont_acc_product(uint64_t (&acc)[2 * NL - 1], const uint32_t (&a)[NL], const uint32_t (&b)[NL]) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
#pragma unroll
    for (int j = 0; j < NL; ++j) acc[i + j] += (uint64_t)a[i] * b[j];
  }
}
BLS_HD void mont_acc_reduce(uint32_t (&out)[NL], uint64_t (&acc)[2 * NL - 1]) {
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const uint64_t t = acc[k] + carry;
    const uint32_t m = ((uint32_t)t * BLS_PINV28) & LMASK;
#pragma unroll
    for (int j = 1; j < NL; ++j) acc[k + j] += (uint64_t)m * BLS_P28[j];
    carry = (t + (uint64_t)m * BLS_P28[0]) >> LW;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
    const uint64_t t = acc[k] + carry;
    out[k - NL] = (uint32_t)t & LMASK;
    carry = t >> LW;
  }
  out[NL - 1] = (uint32_t)carry;
}

// A term of a lazy sum is anything with compile-time bounds `ll` (sum of La Lb) and `vv` (sum of Va Vb) of the products it adds
// and accumulate<C>(acc), C = the coefficient being computed (used by the serial Fp2 backend; 0 here).
// One Fp product: references to its two factors (bounds travel in the types).
template <class A, class B> struct FpProd {
  const A& a; const B& b;
  static constexpr int ll = A::kL * B::kL, vv = A::kV * B::kV;
  template <int C> BLS_HD void accumulate(uint64_t (&acc)[2 * NL - 1]) const { mont_acc_product(acc, a.l, b.l); }
};
template <int La, int Va, int Lb, int Vb>
BLS_HD FpProd<Fp<La, Va>, Fp<Lb, Vb>> pr(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) { return FpProd<Fp<La, Va>, Fp<Lb, Vb>>{a, b}; }

template <int C, class... T>
BLS_HD FpR dot_terms(const T&... t) {
  static_assert((0 + ... + T::ll) <= MONT_LL_MAX, "dot: limb bounds overflow the 64-bit column accumulators");
  static_assert((0 + ... + T::vv) <= MONT_VV_MAX, "dot: value bounds break the < 2p result guarantee");
  BLS_COUNT(6);
  uint64_t acc[2 * NL - 1];
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; ++k) acc[k] = 0;
  (t.template accumulate<C>(acc), ...);
  FpR r;
  mont_acc_reduce(r.l, acc);
  r.chk();
  return r;
}
// dot(pr(a1, b1), ..., pr(aN, bN)) = (a1 b1 + ... + aN bN) / R  (mod p), result < 2p, limbs normalised
template <class... P>
BLS_HD FpR dot(const P&... p) { return dot_terms<0>(p...); }

// ------------------------------------------------------------------------------------------------
// additions: limb-wise, no carries.  When the result would leave the representable bounds (limb < 15 * 2^28,
// value < 64 * 1.01 p) the larger operand is conditioned first (norm / weak_reduce), decided at compile time.
constexpr int L_CAP = 15;
constexpr int V_CAP = 64;

template <int L, int V> BLS_HD Fp<1, V> norm(const Fp<L, V>& a);
template <int L, int V> BLS_HD Fp<1, 2> weak_reduce(const Fp<L, V>& a);

template <int La, int Va, int Lb, int Vb>
BLS_HD auto add(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) {
  if constexpr (Va + Vb > V_CAP) {
    if constexpr (Va >= Vb) return add(weak_reduce(a), b); else return add(a, weak_reduce(b));
  } else if constexpr (La + Lb > L_CAP) {
    if constexpr (La >= Lb) return add(norm(a), b); else return add(a, norm(b));
  } else {
    Fp<La + Lb, Va + Vb> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] + b.l[i];
    r.chk();
    return r;
  }
}

template <int La, int Va>
BLS_HD auto dbl(const Fp<La, Va>& a) {
  if constexpr (2 * Va > V_CAP) return dbl(weak_reduce(a));
  else if constexpr (2 * La > L_CAP) return dbl(norm(a));
  else {
    Fp<2 * La, 2 * Va> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] << 1;
    r.chk();
    return r;
  }
}

// a - b computed as a + (K p, spread so no limb borrows) - b ; K = Vb + 1 multiples of p are added.
template <int La, int Va, int Lb, int Vb>
BLS_HD auto sub(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) {
  static_assert(BLS_SPREAD_MIN_LIMB >= BLS_SPREAD_LMAX, "spread table assumption");
  if constexpr (Vb + 1 > BLS_SPREAD_KMAX || Va + Vb + 1 > V_CAP) {
    if constexpr (Va > Vb) return sub(weak_reduce(a), b); else return sub(a, weak_reduce(b));
  } else if constexpr (Lb > BLS_SPREAD_LMAX || La + Lb + 1 > L_CAP) {
    if constexpr (La > Lb) return sub(norm(a), b); else return sub(a, norm(b));
  } else {
    Fp<La + Lb + 1, Va + Vb + 1> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] + (BLS_SPREAD[Vb][Lb - 1][i] - b.l[i]);
    r.chk();
    return r;
  }
}

template <int Lb, int Vb>
BLS_HD auto neg(const Fp<Lb, Vb>& b) {
  if constexpr (Vb + 1 > BLS_SPREAD_KMAX) return neg(weak_reduce(b));
  else if constexpr (Lb > BLS_SPREAD_LMAX) return neg(norm(b));
  else {
    Fp<Lb + 1, Vb + 1> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = BLS_SPREAD[Vb][Lb - 1][i] - b.l[i];
    r.chk();
    return r;
  }
}

// One parallel carry pass: limbs come back to 28 bits (+ at most L units), value unchanged.
template <int L, int V>
BLS_HD Fp<1, V> norm(const Fp<L, V>& a) {
  BLS_COUNT(4);
  Fp<1, V> r;
  r.l[0] = a.l[0] & LMASK;
#pragma unroll
  for (int i = 1; i < NL - 1; ++i) r.l[i] = (a.l[i] & LMASK) + (a.l[i - 1] >> LW);
  r.l[NL - 1] = a.l[NL - 1] + (a.l[NL - 2] >> LW);
  r.chk();
  return r;
}

// Cheap value reduction (about a tenth of a multiplication): subtract q*p with q estimated from the top
// limb, computed as x + q * (2^392 - p) with the 2^392 overflow dropped.  q = floor(x_hi * M / 2^40),
// M = floor(2^40 / (p_hi + 1)), never exceeds x / p and misses it by less than 1.01, so the result is
// < 2.01 p (V = 2 with the slack above).  Requires V <= 64 so that x_hi * M fits 64 bits comfortably.
constexpr uint32_t P_HI = BLS_P28[NL - 1];                               // p >> 364
constexpr uint64_t WR_M = ((uint64_t)1 << 40) / (uint64_t)(P_HI + 1);

template <int L, int V>
BLS_HD FpR weak_reduce(const Fp<L, V>& a) {
  static_assert(V <= 64, "weak_reduce: value bound too large for the quotient estimate");
  static_assert(L <= L_CAP, "weak_reduce: limb bound");
  BLS_COUNT(3);
  // top 28-bit digit of the value, short by at most one unit (carries still pending below limb 12)
  const uint32_t top = a.l[NL - 1] + (a.l[NL - 2] >> LW);
  const uint32_t q = (uint32_t)(((uint64_t)top * WR_M) >> 40);
  // x + q * (2^392 - p) with an exact carry chain; the 2^392 multiples fall off the top limb.
  // Four instructions per limb: one MAD (carry as the 64-bit addend), one 64-bit add of the limb, mask, shift.
  FpR r;
  uint64_t carry = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const uint32_t ci = (i < NL - 1) ? (LMASK - BLS_P28[i] + (i == 0 ? 1u : 0u)) : (LMASK - BLS_P28[i]);
    uint64_t acc = (uint64_t)q * ci + carry;
    acc += a.l[i];
    r.l[i] = (uint32_t)acc & LMASK;
    carry = acc >> LW;
  }
  r.chk();
  return r;
}

// Make an element fit declared bounds with the cheapest sufficient step.
template <int LT, int VT, int L, int V>
BLS_HD auto fit(const Fp<L, V>& a) {
  static_assert(LT >= 1 && VT >= 2, "fit: target must admit a reduced element");
  if constexpr (V > VT) return weak_reduce(a);
  else if constexpr (L > LT) return norm(a);
  else return a;
}

// Bring any element back to < 2p with normalised limbs: one Montgomery multiplication by "one".
template <int L, int V>
BLS_HD FpR reduce(const Fp<L, V>& a) {
  Fp<1, 1> one;
#pragma unroll
  for (int i = 0; i < NL; ++i) one.l[i] = BLS_ONE_INT[i];
  return mul(a, one);
}

// Re-label an element with bounds the type system cannot derive (checked at run time in the host
// bound-check build, tests/hostsim).  Use only with a comment that proves the claim.
template <int L2, int V2, int L, int V>
BLS_HD Fp<L2, V2> assume_bounds(const Fp<L, V>& a) {
  Fp<L2, V2> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i];
  r.chk();
  return r;
}

template <int L, int V>
BLS_HD Fp<L, V> select(bool c, const Fp<L, V>& a, const Fp<L, V>& b) {   // c ? a : b
  Fp<L, V> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

BLS_HD Fp<1, 1> fp_const(const uint32_t (&c)[NL]) {
  Fp<1, 1> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = c[i];
  return r;
}
BLS_HD Fp<1, 1> fp_one() { return fp_const(BLS_ONE_INT); }
BLS_HD Fp<1, 1> fp_zero() { Fp<1, 1> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = 0;
  return r; }

// ------------------------------------------------------------------------------------------------
// Canonical form (only at the ABI boundary and for the exact zero tests of Mode A).
// Input: normalised limbs, value < 2p.  Output: the unique representative in [0, p).
BLS_HD Fp<1, 1> canonical(const FpR& a) {
  // full carry propagation (FIPS output limbs are already < 2^28; this makes it exact for any FpR)
  uint32_t t[NL];
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) { uint32_t v = a.l[i] + carry; t[i] = v & LMASK; carry = v >> LW; }
  t[NL - 1] = a.l[NL - 1] + carry;
  // s = t - p with borrow
  uint32_t s[NL];
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    uint32_t v = t[i] - BLS_P28[i] - borrow;
    borrow = (v >> 31) & 1u;              // limbs < 2^29, so a wrap shows in bit 31
    s[i] = (i < NL - 1) ? (v & LMASK) : v;
  }
  Fp<1, 1> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = borrow ? t[i] : s[i];
  return r;
}

// Exact zero test (the reference compares canonical arkworks values, e.g. `lambda_denominator != zero`,
// src/miller_loop_native_optimized.rs:16).  A value < 2.02 p that is 0 mod p is one of 0, p, 2p.
template <int L, int V>
BLS_HD bool is_zero(const Fp<L, V>& a) {
  const FpR r = fit<1, 2>(a);
  uint32_t t[NL];
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) { uint32_t v = r.l[i] + carry; t[i] = v & LMASK; carry = v >> LW; }
  t[NL - 1] = r.l[NL - 1] + carry;
  uint32_t d0 = 0, d1 = 0, d2 = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) { d0 |= t[i]; d1 |= t[i] ^ BLS_P28[i]; d2 |= t[i] ^ BLS_2P28[i]; }
  return (d0 == 0) | (d1 == 0) | (d2 == 0);
}

// ------------------------------------------------------------------------------------------------
// ABI conversions.  External Fp = 6 x u64 little-endian (48 bytes): either arkworks' in-memory
// Montgomery form a * 2^384 mod p [third party] or the canonical limbs of `Bls12_381Base([u64; 6])`
// (src/fields/bls12_381base.rs:20-21).
BLS_HD Fp<1, 1> unpack384(const uint64_t* w) {          // 384-bit integer -> 28-bit limbs (no reduction)
  Fp<1, 1> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int bit = LW * i, word = bit >> 6, sh = bit & 63;
    uint64_t v = w[word] >> sh;
    if (sh > 64 - LW && word + 1 < 6) v |= w[word + 1] << (64 - sh);
    r.l[i] = (uint32_t)v & LMASK;
  }
  return r;
}

BLS_HD void pack384(uint64_t* w, const Fp<1, 1>& a) {    // canonical 28-bit limbs -> 6 x u64
#pragma unroll
  for (int k = 0; k < 6; ++k) w[k] = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int bit = LW * i, word = bit >> 6, sh = bit & 63;
    w[word] |= (uint64_t)a.l[i] << sh;
    if (sh > 64 - LW && word + 1 < 6) w[word + 1] |= (uint64_t)a.l[i] >> (64 - sh);
  }
}

enum IoForm { IO_ARK_MONT = 0, IO_CANONICAL = 1 };

BLS_HD FpR fp_load(const uint64_t* w, int form) {
  Fp<1, 1> x = unpack384(w);    // any 384-bit pattern: value < 2^384 < 10p
  Fp<1, 10> xv;
#pragma unroll
  for (int i = 0; i < NL; ++i) xv.l[i] = x.l[i];
  Fp<1, 1> k = (form == IO_ARK_MONT) ? fp_const(BLS_C_IN_ARK) : fp_const(BLS_C_IN_CANON);
  return mul(xv, k);
}

template <int L, int V>
BLS_HD void fp_store(uint64_t* w, const Fp<L, V>& a, int form) {
  Fp<1, 1> k = (form == IO_ARK_MONT) ? fp_const(BLS_C_OUT_ARK) : fp_const(BLS_RAW_ONE);
  pack384(w, canonical(mul(a, k)));
}

// a^e for a compile-time-known exponent given as 6 little-endian u64 words.  The exponent is a constant, so the
// 2-bit window walk is wave-uniform control flow: ~380 squarings + ~143 multiplications, table of 3 elements.
template <int L, int V>
BLS_HD_NOINLINE FpR fp_pow_const(const Fp<L, V>& a_in, const uint64_t (&e)[6]) {
  const FpR a1 = reduce(a_in);
  const FpR a2 = sqr(a1);
  const FpR a3 = mul(a2, a1);
  FpR r = reduce(fp_one());                  // e == 0 -> one
  bool started = false;
  for (int w = 191; w >= 0; --w) {           // 192 windows cover 384 bits
    const uint32_t d = (uint32_t)(e[w >> 5] >> ((w & 31) * 2)) & 3u;
    if (started) { r = sqr(r); r = sqr(r); }
    if (d) {
      const FpR t = (d == 1) ? a1 : (d == 2) ? a2 : a3;
      if (started) r = mul(r, t); else { r = t; started = true; }
    }
  }
  return r;
}

// a^e for a run-time exponent of `nwords` little-endian u64 words (wave-uniform: the exponent is a kernel argument),
// plain MSB-first square-and-multiply.  Value-equivalent to the reference's NAF walk `pow_fq`
// (src/fields/helpers.rs:176-195), including its quirk that an all-zero exponent returns the base itself.
template <int L, int V>
BLS_HD_NOINLINE FpR fp_pow_words(const Fp<L, V>& a_in, const uint64_t* e, int nwords) {
  const FpR a = reduce(a_in);
  FpR r = a;
  bool started = false;
  for (int b = nwords * 64 - 1; b >= 0; --b) {
    const bool bit = (e[b >> 6] >> (b & 63)) & 1ull;
    if (started) r = sqr(r);
    if (bit) { if (started) r = mul(r, a); else started = true; }
  }
  return r;                                  // exponent 0: `res` is never touched and stays a (helpers.rs:177)
}

// Fermat inversion a^(p-2) (380 squarings + 143 multiplications); zero maps to zero.  Kept as the cross-check of fp_inv below.
template <int L, int V>
BLS_HD FpR fp_inv_fermat(const Fp<L, V>& a) { return fp_pow_const(a, BLS_P_MINUS_2); }

// Inversion by the optimized binary GCD (T. Pornin, "Optimized Binary GCD for Modular Inversion", ePrint 2020/972, Alg. 2):
// the plain binary GCD on (a, b) = (x, p) with cofactors (u, v), a = x u, b = x v (mod p), needs 2 * 381 - 1 halvings; they are
// done 30 at a time on 62-bit approximations of a and b (their low 30 bits, which decide every parity exactly, and their top
// 32 bits, which decide the comparisons well enough -- a wrong comparison can only make the new a or b negative, which is fixed by
// negating it together with its update factors).  Each round then applies the four 31-bit update factors to the full-width
// values: a, b by exact division by 2^30, u, v by a Montgomery-style division mod p.  Values live in 13 x 30-bit limbs so that the
// division is a limb shift; accumulators are signed 64-bit (v_mad_i64_i32).  26 rounds cover any x < 2^390; no data-dependent
// branch.  ~31 k instructions against ~209 k for Fermat (tools/microbench measurements in DESIGN 4).  Zero maps to zero.
constexpr int GCD_L = 13, GCD_W = 30, GCD_ROUNDS = 26;
constexpr uint32_t GCD_MASK = (1u << GCD_W) - 1u;
static_assert(GCD_ROUNDS * GCD_W >= 2 * 390 - 1, "rounds cover every value the 13-limb form can hold");

// r = (f x + g y [+ q p]) / 2^30 over 13 limbs, top limb signed; MOD adds the multiple of p that makes the sum divisible
template <bool MOD>
BLS_HD void gcd_apply(int32_t (&r)[GCD_L], int32_t f, int32_t g, const int32_t (&x)[GCD_L], const int32_t (&y)[GCD_L]) {
  int64_t acc = (int64_t)f * x[0] + (int64_t)g * y[0];
  uint32_t q = 0;
  if (MOD) {
    q = ((uint32_t)acc * BLS_PINV30) & GCD_MASK;
    acc += (int64_t)(int32_t)q * (int32_t)BLS_P30[0];
  }
  acc >>= GCD_W;                                           // the low limb is zero by construction
#pragma unroll
  for (int i = 1; i < GCD_L; ++i) {
    acc += (int64_t)f * x[i] + (int64_t)g * y[i];
    if (MOD) acc += (int64_t)(int32_t)q * (int32_t)BLS_P30[i];
    r[i - 1] = (int32_t)((uint32_t)acc & GCD_MASK);
    acc >>= GCD_W;
  }
  r[GCD_L - 1] = (int32_t)acc;
}
// x = -x when neg (limbs 0..11 unsigned 30-bit, limb 12 signed)
BLS_HD void gcd_cond_negate(int32_t (&x)[GCD_L], bool neg) {
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < GCD_L - 1; ++i) {
    const uint32_t d = 0u - (uint32_t)x[i] - borrow;
    borrow = d >> 31;                                      // limbs are < 2^30, so a wrap shows in bit 31
    x[i] = neg ? (int32_t)(d & GCD_MASK) : x[i];
  }
  x[GCD_L - 1] = neg ? (int32_t)(0u - (uint32_t)x[GCD_L - 1] - borrow) : x[GCD_L - 1];
}

template <int L, int V>
BLS_HD_NOINLINE FpR fp_inv(const Fp<L, V>& a_in) {
  // exact 28-bit limbs of the input (value < 2.02 p), then 13 x 30-bit limbs: limb i = bits [30 i, 30 i + 30)
  const FpR xr = fit<1, 2>(a_in);
  uint32_t t[NL + 1];
  {
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) { const uint32_t w = xr.l[i] + carry; t[i] = w & LMASK; carry = w >> LW; }
    t[NL - 1] = xr.l[NL - 1] + carry;
    t[NL] = 0;
  }
  int32_t a[GCD_L], b[GCD_L], u[GCD_L], v[GCD_L];
#pragma unroll
  for (int i = 0; i < GCD_L; ++i) {
    a[i] = (int32_t)(((t[i] >> (2 * i)) | (t[i + 1] << (LW - 2 * i))) & GCD_MASK);
    b[i] = (int32_t)BLS_P30[i];
    u[i] = (i == 0) ? 1 : 0;
    v[i] = 0;
  }
  for (int round = 0; round < GCD_ROUNDS; ++round) {
    // 62-bit approximations: low 30 bits | top 32 bits of the window that ends at bit n = max(len a, len b, 62)
    uint32_t ha = (uint32_t)a[2], ma = (uint32_t)a[1], la = (uint32_t)a[0];
    uint32_t hb = (uint32_t)b[2], mb = (uint32_t)b[1], lb = (uint32_t)b[0];
    bool found = false;
#pragma unroll
    for (int i = GCD_L - 1; i >= 3; --i) {
      const bool take = !found && (((uint32_t)a[i] | (uint32_t)b[i]) != 0);
      ha = take ? (uint32_t)a[i] : ha; ma = take ? (uint32_t)a[i - 1] : ma; la = take ? (uint32_t)a[i - 2] : la;
      hb = take ? (uint32_t)b[i] : hb; mb = take ? (uint32_t)b[i - 1] : mb; lb = take ? (uint32_t)b[i - 2] : lb;
      found = found || take;
    }
    const uint32_t top = ha | hb;
    int s = top ? 32 - __builtin_clz(top) : 0;            // bit length of the window's top limb
    // no limb above limb 2 in use: the window is bits [30, 90) and n = max(len, 62), i.e. s >= 2.  Otherwise s as it is: wa >> s then
    // holds exactly the 32 bits below bit n that Alg. 2 asks for (with the clamp applied everywhere a top limb of bit length 1 kept
    // only 31 of them -- still convergent with ~17 spare halvings in 26 rounds, but not what the convergence argument covers).
    s = (!found && s < 2) ? 2 : s;
    // floor(value / 2^(30 t - 32)) >> s = the 32 bits below bit n (t = index of the window's top limb)
    const uint64_t wa = ((((uint64_t)ha << GCD_W) | ma) << 2) | (la >> 28);
    const uint64_t wb = ((((uint64_t)hb << GCD_W) | mb) << 2) | (lb >> 28);
    uint64_t xa = ((wa >> s) << GCD_W) | (uint32_t)a[0];
    uint64_t xb = ((wb >> s) << GCD_W) | (uint32_t)b[0];
    int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll 2
    for (int j = 0; j < GCD_W; ++j) {
      const bool odd = (xa & 1ull) != 0;
      const bool sw = odd && (xa < xb);
      const uint64_t ta = sw ? xb : xa; xb = sw ? xa : xb; xa = ta;
      const int32_t tf = sw ? f1 : f0; f1 = sw ? f0 : f1; f0 = tf;
      const int32_t tg = sw ? g1 : g0; g1 = sw ? g0 : g1; g0 = tg;
      xa -= odd ? xb : 0ull;
      f0 -= odd ? f1 : 0;
      g0 -= odd ? g1 : 0;
      xa >>= 1;
      f1 = (int32_t)((uint32_t)f1 << 1); g1 = (int32_t)((uint32_t)g1 << 1);   // doubling of a possibly negative factor: shift as unsigned
    }
    int32_t na[GCD_L], nb[GCD_L];
    gcd_apply<false>(na, f0, g0, a, b);
    gcd_apply<false>(nb, f1, g1, a, b);
    const bool nega = na[GCD_L - 1] < 0, negb = nb[GCD_L - 1] < 0;
    gcd_cond_negate(na, nega); gcd_cond_negate(nb, negb);
    f0 = nega ? -f0 : f0; g0 = nega ? -g0 : g0;
    f1 = negb ? -f1 : f1; g1 = negb ? -g1 : g1;
    int32_t nu[GCD_L], nv[GCD_L];
    gcd_apply<true>(nu, f0, g0, u, v);
    gcd_apply<true>(nv, f1, g1, u, v);
#pragma unroll
    for (int i = 0; i < GCD_L; ++i) { a[i] = na[i]; b[i] = nb[i]; u[i] = nu[i]; v[i] = nv[i]; }
  }
  // now a = 0, b = gcd = 1 (or x = 0 and v = 0) and v = x^-1 mod p as an integer with |v| < 27 p (every round adds at most p to
  // max(|u|, |v|): |f| + |g| <= 2^30 and q < 2^30; observed < 3 p): add 32 p, back to 28-bit limbs
  uint32_t w[GCD_L + 1];
  {
    int32_t carry = 0;
#pragma unroll
    for (int i = 0; i < GCD_L - 1; ++i) { const int32_t q = v[i] + (int32_t)BLS_32P30[i] + carry; w[i] = (uint32_t)q & GCD_MASK; carry = q >> GCD_W; }
    w[GCD_L - 1] = (uint32_t)(v[GCD_L - 1] + (int32_t)BLS_32P30[GCD_L - 1] + carry);
    w[GCD_L] = 0;
  }
  Fp<1, 60> r;                                             // 5 p < value < 59 p
  r.l[0] = w[0] & LMASK;
#pragma unroll
  for (int j = 1; j < NL; ++j) r.l[j] = ((w[j - 1] >> (GCD_W - 2 * j)) | (w[j] << (2 * j))) & LMASK;
  r.chk();
  // integer inverse of the internal form x = y R  ->  internal form of 1/y:  (1 / (y R)) R^3 / R = R / y
  return mul(r, fp_const(BLS_R3_INT));
}

// Simultaneous inversion (Montgomery's trick) of N values held by ONE lane: 3 N multiplications and a single
// fp_inv.  Zeros stay zero and do not disturb their neighbours.  Used to amortise the one Fp inversion under every
// final exponentiation across elements (k_fp_batch_inv, k_misc.hip).
constexpr int FP_BATCH_INV_CHUNK = 16;       // values per lane in k_fp_batch_inv (k_misc.hip)
template <int N>
BLS_HD void fp_batch_inv(FpR (&x)[N]) {
  FpR pre[N];                                  // pre[i] = x_0 ... x_(i-1), zeros read as one
  bool z[N];
  FpR acc = FpR(fp_one());
#pragma unroll
  for (int i = 0; i < N; ++i) {
    z[i] = is_zero(x[i]);
    x[i] = select(z[i], FpR(fp_one()), x[i]);
    pre[i] = acc;
    acc = mul(acc, x[i]);
  }
  FpR inv = fp_inv(acc);                       // (x_0 ... x_(N-1))^-1
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    const FpR o = mul(inv, pre[i]);            // x_i^-1
    inv = mul(inv, x[i]);
    x[i] = select(z[i], FpR(fp_zero()), o);
  }
}

// Internal-form Fp values parked in device scratch between kernels: the 14 limbs as they are (normalised, value < 2p),
// 16 dwords apart so that the accesses are four aligned dwordx4.
constexpr int RAW_FP_STRIDE = 16;
BLS_HD FpR raw_fp_load(const uint32_t* p) { FpR r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = p[i];
  return r; }
BLS_HD void raw_fp_store(uint32_t* p, const FpR& a) {
#pragma unroll
  for (int i = 0; i < NL; ++i) p[i] = a.l[i];
}

// Lane-sliced scratch for multi-kernel pipelines (final exponentiation stages): value k of lane `lane` keeps its 14 limbs as three
// 16-byte quarters at  base[((3 k + q) * lanes + lane) * 4 ..]  and the two top limbs as one 8-byte piece at  top[(k * lanes + lane) * 2 ..]
// (the `top` plane follows the quarter planes of ALL values), so that one store instruction of a wave writes 64 consecutive 16-byte
// (8-byte) pieces -- fully coalesced, whatever the number of values per lane -- and a parked value costs 56 bytes, not 64.
constexpr int LANE_SCRATCH_DWORDS = NL;      // per parked Fp
struct LaneScratch {
  uint32_t* base;
  uint32_t* top;
  size_t lanes, lane;
  BLS_HD void put(int k, const FpR& a) const {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      uint32_t* d = base + (((size_t)(3 * k + q)) * lanes + lane) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = a.l[4 * q + j];
    }
    uint32_t* t = top + ((size_t)k * lanes + lane) * 2;
    t[0] = a.l[12]; t[1] = a.l[13];
  }
  BLS_HD FpR get(int k) const {
    FpR r;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const uint32_t* d = base + (((size_t)(3 * k + q)) * lanes + lane) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) r.l[4 * q + j] = d[j];
    }
    const uint32_t* t = top + ((size_t)k * lanes + lane) * 2;
    r.l[12] = t[0]; r.l[13] = t[1];
    return r;
  }
};
static_assert(NL == 14, "LaneScratch layout: three quarters and a pair");
// nvals = values per lane the buffer was sized for (LANE_SCRATCH_DWORDS * nvals * lanes dwords)
BLS_HD LaneScratch lane_scratch(uint32_t* buf, size_t lanes, size_t lane, size_t nvals) {
  return LaneScratch{buf, buf + 12 * nvals * lanes, lanes, lane};
}

// Square-root candidate a^((p+1)/4) (p = 3 mod 4); the caller checks candidate^2 == a.
template <int L, int V>
BLS_HD FpR fp_sqrt_candidate(const Fp<L, V>& a) { return fp_pow_const(a, BLS_P_PLUS_1_DIV_4); }

template <int La, int Va, int Lb, int Vb>
BLS_HD bool fp_eq(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) { return is_zero(sub(a, b)); }

// parity of the canonical representative: sgn0_fq, src/fields/helpers.rs:159-167
template <int L, int V>
BLS_HD bool fp_sgn0(const Fp<L, V>& a) {
  const Fp<1, 1> c = canonical(mul(a, fp_const(BLS_RAW_ONE)));     // leave Montgomery form, then canonicalise
  return (c.l[0] & 1u) != 0;
}

}  // namespace bls
