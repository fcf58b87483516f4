// Fp2 = Fp[u]/(u^2+1), SERIAL backend: both coefficients live in one lane (one pairing per lane).
// Used by the host simulation (tests/hostsim) and selectable on the GPU with -DBLS_LANES=1.
// Formulas: src/fields_as_trees/fq2_target_tree.rs:54-142 (reference).
#pragma once
#include "fp.hip.h"

namespace bls {

// ================================================================================================ Fp2
template <int L, int V>
struct Fp2 {
  Fp<L, V> c0, c1;
  Fp2() = default;
  BLS_HD Fp2(const Fp<L, V>& a, const Fp<L, V>& b) : c0(a), c1(b) {}
  template <int L2, int V2>
  BLS_HD Fp2(const Fp2<L2, V2>& o) : c0(o.c0), c1(o.c1) {}
};
using Fp2R = Fp2<1, 2>;

// build an Fp2 from two Fp of (possibly) different bounds
template <int L0, int V0, int L1, int V1>
BLS_HD auto make_fp2(const Fp<L0, V0>& a, const Fp<L1, V1>& b) {
  constexpr int L = L0 > L1 ? L0 : L1;
  constexpr int V = V0 > V1 ? V0 : V1;
  return Fp2<L, V>(Fp<L, V>(a), Fp<L, V>(b));
}
template <int La, int Va, int Lb, int Vb>
BLS_HD auto add(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return make_fp2(add(a.c0, b.c0), add(a.c1, b.c1)); }
template <int La, int Va, int Lb, int Vb>
BLS_HD auto sub(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return make_fp2(sub(a.c0, b.c0), sub(a.c1, b.c1)); }
template <int L, int V>
BLS_HD auto neg(const Fp2<L, V>& a) { return make_fp2(neg(a.c0), neg(a.c1)); }
template <int L, int V>
BLS_HD auto dbl(const Fp2<L, V>& a) { return make_fp2(dbl(a.c0), dbl(a.c1)); }
template <int LT, int VT, int L, int V>
BLS_HD auto fit(const Fp2<L, V>& a) { return make_fp2(fit<LT, VT>(a.c0), fit<LT, VT>(a.c1)); }
template <int L, int V>
BLS_HD Fp2<L, V> select(bool c, const Fp2<L, V>& a, const Fp2<L, V>& b) {
  return Fp2<L, V>(select(c, a.c0, b.c0), select(c, a.c1, b.c1));
}

// Largest operand bounds the fused schoolbook product accepts: with the c1 operand of the lower-L side
// negated (L + 1, V + 1):  La*Lb + (Lmin + 1)*Lmax <= 17  and  Va*Vb + (Vneg + 1)*Vother <= 2500.
template <int La, int Va, int Lb, int Vb>
struct Fp2MulFits {
  static constexpr bool neg_a = (La <= Lb);
  static constexpr int ll = La * Lb + (neg_a ? (La + 1) * Lb : (Lb + 1) * La);
  static constexpr int vv = Va * Vb + (neg_a ? (Va + 1) * Vb : (Vb + 1) * Va);
  static constexpr bool ok = ll <= MONT_LL_MAX && vv <= MONT_VV_MAX;
};

// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u   -- fq2_target_tree.rs:97-115
template <int La, int Va, int Lb, int Vb>
BLS_HD Fp2R mul_raw(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) {
  using F = Fp2MulFits<La, Va, Lb, Vb>;
  static_assert(F::ok, "fp2 mul: operand bounds too large (fit<> them first)");
  if constexpr (F::neg_a)
    return Fp2R(mul2(a.c0, b.c0, neg(a.c1), b.c1), mul2(a.c0, b.c1, a.c1, b.c0));
  else
    return Fp2R(mul2(a.c0, b.c0, a.c1, neg(b.c1)), mul2(a.c0, b.c1, a.c1, b.c0));
}

// Multiplication with automatic operand conditioning: normalise / weakly reduce whichever operand makes
// the fused product overflow its static bounds.
template <int La, int Va, int Lb, int Vb>
BLS_HD Fp2R mul(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) {
  if constexpr (Fp2MulFits<La, Va, Lb, Vb>::ok) return mul_raw(a, b);
  else if constexpr (Va * Vb + (Va + 1) * Vb > MONT_VV_MAX) {
    if constexpr (Va >= Vb) return mul(fit<1, 2>(a), b); else return mul(a, fit<1, 2>(b));
  } else {
    if constexpr (La >= Lb) return mul(fit<1, Va>(a), b); else return mul(a, fit<1, Vb>(b));
  }
}

// Lazy sums of Fp2 products and squares (fp.hip.h `dot_terms`): one reduction per coefficient for
//     dot2(pr2(A1, B1), sq2(A2), ...) = A1 B1 + A2^2 + ...
template <class A, class B> struct Fp2Prod { const A& a; const B& b; };
template <int La, int Va, int Lb, int Vb>
BLS_HD Fp2Prod<Fp2<La, Va>, Fp2<Lb, Vb>> pr2(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return Fp2Prod<Fp2<La, Va>, Fp2<Lb, Vb>>{a, b}; }
template <class A> struct Fp2Sq { const A& a; };
template <int L, int V>
BLS_HD Fp2Sq<Fp2<L, V>> sq2(const Fp2<L, V>& a) { return Fp2Sq<Fp2<L, V>>{a}; }

template <class A0, class A1N, class B0> struct Fp2MulTerm {      // C = 0: a0 b0 + (-a1) b1 ; C = 1: a0 b1 + a1 b0
  A0 a0, a1; A1N a1n; B0 b0, b1;
  static constexpr int ll = A0::kL * B0::kL + A1N::kL * B0::kL, vv = A0::kV * B0::kV + A1N::kV * B0::kV;
  template <int C> BLS_HD void accumulate(uint64_t (&acc)[2 * NL - 1]) const {
    if (C == 0) { mont_acc_product(acc, a0.l, b0.l); mont_acc_product(acc, a1n.l, b1.l); }
    else { mont_acc_product(acc, a0.l, b1.l); mont_acc_product(acc, a1.l, b0.l); }
  }
};
template <class S, class D, class A> struct Fp2SqTerm {           // C = 0: (a0 + a1)(a0 - a1) ; C = 1: (2 a0) a1
  S s, t; D d; A a1;
  static constexpr int ll = S::kL * D::kL, vv = S::kV * D::kV;
  template <int C> BLS_HD void accumulate(uint64_t (&acc)[2 * NL - 1]) const {
    if (C == 0) mont_acc_product(acc, s.l, d.l); else mont_acc_product(acc, t.l, a1.l);
  }
};
template <int La, int Va, int Lb, int Vb>
BLS_HD auto fp2_term(const Fp2Prod<Fp2<La, Va>, Fp2<Lb, Vb>>& p) {
  return Fp2MulTerm<Fp<La, Va>, decltype(neg(p.a.c1)), Fp<Lb, Vb>>{p.a.c0, p.a.c1, neg(p.a.c1), p.b.c0, p.b.c1};
}
template <int L, int V>
BLS_HD auto fp2_term(const Fp2Sq<Fp2<L, V>>& q) {
  return Fp2SqTerm<Fp<2 * L, 2 * V>, decltype(sub(q.a.c0, q.a.c1)), Fp<L, V>>{add(q.a.c0, q.a.c1), dbl(q.a.c0), sub(q.a.c0, q.a.c1), q.a.c1};
}
template <class... T>
BLS_HD Fp2R dot2_terms(const T&... t) { return Fp2R(dot_terms<0>(t...), dot_terms<1>(t...)); }
template <class... T>
BLS_HD Fp2R dot2(const T&... t) { return dot2_terms(fp2_term(t)...); }

// complex squaring (a0+a1)(a0-a1), 2 a0 a1   -- fq2_target_tree.rs:80-91
template <int L, int V>
BLS_HD Fp2R sqr(const Fp2<L, V>& a) {
  if constexpr ((2 * L) * (2 * L + 1) <= MONT_LL_MAX && (2 * V) * (2 * V + 1) <= MONT_VV_MAX) {
    return Fp2R(mul(add(a.c0, a.c1), sub(a.c0, a.c1)), mul(dbl(a.c0), a.c1));
  } else if constexpr ((2 * V) * (2 * V + 1) > MONT_VV_MAX) {
    return sqr(fit<1, 2>(a));
  } else {
    return sqr(fit<1, V>(a));
  }
}

// Fp2 x Fp (line evaluation scaling, src/miller_loop_native.rs:143-147 commented `ell`)
template <int L, int V, int Lk, int Vk>
BLS_HD Fp2R mul_fp(const Fp2<L, V>& a, const Fp<Lk, Vk>& k) { return Fp2R(mul(a.c0, k), mul(a.c1, k)); }

template <int L, int V>
BLS_HD auto conj(const Fp2<L, V>& a) { return make_fp2(a.c0, neg(a.c1)); }   // fq2_target_tree.rs:117-122

// multiply by xi = 1 + u: (c0 - c1, c0 + c1)   -- fq2_target_tree.rs:137-142
template <int L, int V>
BLS_HD auto mul_xi(const Fp2<L, V>& a) { return make_fp2(sub(a.c0, a.c1), add(a.c0, a.c1)); }

// norm a0^2 + a1^2 and the inverse conj(a) / norm given 1 / norm (the two halves of fq2_target_tree.rs:66-78)
BLS_HD FpR fp2_norm(const Fp2R& a) { return mul2(a.c0, a.c0, a.c1, a.c1); }
BLS_HD Fp2R fp2_inv_given(const Fp2R& a, const FpR& norm_inv) { return Fp2R(mul(a.c0, norm_inv), mul(neg(a.c1), norm_inv)); }

// inverse via the norm (fq2_target_tree.rs:66-78); zero maps to zero.
template <int L, int V>
BLS_HD_NOINLINE Fp2R inv(const Fp2<L, V>& a_in) {
  const Fp2R a = fit<1, 2>(a_in);
  return fp2_inv_given(a, fp_inv(fp2_norm(a)));
}

// Fp2 number `idx` of this element in a LaneScratch (two Fp per Fp2 on the one lane)
constexpr int SCRATCH_FP_PER_FP2 = 2;
template <class Store> BLS_HD void fp2_slot_put(const Store& s, int idx, const Fp2R& a) { s.put(2 * idx, a.c0); s.put(2 * idx + 1, a.c1); }
template <class Store> BLS_HD Fp2R fp2_slot_get(const Store& s, int idx) { return Fp2R(s.get(2 * idx), s.get(2 * idx + 1)); }
BLS_HD void fp2_scratch_put(const LaneScratch& s, int idx, const Fp2R& a) { s.put(2 * idx, a.c0); s.put(2 * idx + 1, a.c1); }
BLS_HD Fp2R fp2_scratch_get(const LaneScratch& s, int idx) { return Fp2R(s.get(2 * idx), s.get(2 * idx + 1)); }

template <int L, int V>
BLS_HD bool is_zero(const Fp2<L, V>& a) { return is_zero(a.c0) & is_zero(a.c1); }

BLS_HD Fp2<1, 1> fp2_zero() { return Fp2<1, 1>(fp_zero(), fp_zero()); }
BLS_HD Fp2<1, 1> fp2_one() { return Fp2<1, 1>(fp_one(), fp_zero()); }
BLS_HD Fp2<1, 1> fp2_const(const uint32_t (&c0)[NL], const uint32_t (&c1)[NL]) { return Fp2<1, 1>(fp_const(c0), fp_const(c1)); }
// words between consecutive elements handled by one lane / lanes cooperating on one element
constexpr int LANES_PER_ELEM = 1;
BLS_HD int lane_part() { return 0; }

BLS_HD Fp2R fp2_load(const uint64_t* w, int form) { return Fp2R(fp_load(w, form), fp_load(w + 6, form)); }
template <int L, int V>
BLS_HD void fp2_store(uint64_t* w, const Fp2<L, V>& a, int form) { fp_store(w, a.c0, form); fp_store(w + 6, a.c1, form); }

// both coefficients as Fp values usable by this lane (serial: trivially the members)
template <int L, int V> BLS_HD Fp<L, V> coeff0(const Fp2<L, V>& a) { return a.c0; }
template <int L, int V> BLS_HD Fp<L, V> coeff1(const Fp2<L, V>& a) { return a.c1; }
template <int L0, int V0, int L1, int V1>
BLS_HD auto fp2_from_coeffs(const Fp<L0, V0>& c0, const Fp<L1, V1>& c1) { return make_fp2(c0, c1); }
// coefficients stored at two unrelated addresses (MyFq12 / MyFq6 flat orders, src/fields/helpers.rs:39-41)
BLS_HD Fp2R fp2_load2(const uint64_t* w0, const uint64_t* w1, int form) { return Fp2R(fp_load(w0, form), fp_load(w1, form)); }
template <int L, int V>
BLS_HD void fp2_store2(uint64_t* w0, uint64_t* w1, const Fp2<L, V>& a, int form) { fp_store(w0, a.c0, form); fp_store(w1, a.c1, form); }

}  // namespace bls
