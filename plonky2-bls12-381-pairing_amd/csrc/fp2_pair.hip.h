// Fp2 = Fp[u]/(u^2+1), LANE-PAIR backend (gfx950): one Fp2 element is split across two adjacent lanes of a
// wave -- the even lane holds c0, the odd lane holds c1 -- so one pairing occupies a lane pair and a 64-lane wave
// carries 32 pairings.
//
// Why: the whole Miller-loop / final-exponentiation state of one pairing is ~340 Fp-sized registers in the serial
// layout, more than a lane owns even at one wave per SIMD (512 VGPRs); it spills to scratch and the kernel ends
// up bound by HBM traffic (profiles/r01_v1_baseline).  Splitting Fp2 halves the per-lane state and the per-lane code
// (each lane executes ONE fused multiply-pair per Fp2 product instead of two), so the state stays in registers.
//
// An Fp2 product (a0 + a1 u)(b0 + b1 u) maps onto the pair with no idle lane:
//     even lane:  c0 = a0*b0 + (-a1)*b1         odd lane:  c1 = a0*b1 + a1*b0
// i.e. both lanes run  mul2(a0, b_self, +-a1, b_other)  after fetching the partner's coefficients with two DPP
// quad_perm moves per limb (pair swap [1,0,3,2], pair broadcasts [0,0,2,2] / [1,1,3,3]; full-rate VALU, no
// LDS).  Squaring is (a0+a1)(a0-a1) on the even lane and
// (2 a0) a1 on the odd lane: one single-product multiplication each.
//
// Formulas: src/fields_as_trees/fq2_target_tree.rs:54-142 (reference).  Same API as fp2_serial.hip.h.
#pragma once
#include "fp.hip.h"

namespace bls {

constexpr int LANES_PER_ELEM = 2;

// 0 on the lane holding c0, 1 on the lane holding c1
BLS_HD int lane_part() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)(threadIdx.x & 1u);
#else
  return 0;
#endif
}

// value held by the partner lane (lane ^ 1)
BLS_HD uint32_t dpp_swap(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true);
#else
  return x;   // host pass only type-checks this backend; the host simulation uses fp2_serial.hip.h
#endif
}
// c0 of the pair on both lanes / c1 of the pair on both lanes: DPP broadcasts inside each lane pair
BLS_HD uint32_t dpp_bcast0(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xA0 /* quad_perm:[0,0,2,2] */, 0xF, 0xF, true);
#else
  return x;
#endif
}
BLS_HD uint32_t dpp_bcast1(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xF5 /* quad_perm:[1,1,3,3] */, 0xF, 0xF, true);
#else
  return x;
#endif
}
template <int L, int V>
BLS_HD Fp<L, V> bcast0(const Fp<L, V>& a) { Fp<L, V> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = dpp_bcast0(a.l[i]);
  return r; }
template <int L, int V>
BLS_HD Fp<L, V> bcast1(const Fp<L, V>& a) { Fp<L, V> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = dpp_bcast1(a.l[i]);
  return r; }

template <int L, int V>
BLS_HD Fp<L, V> swap(const Fp<L, V>& a) {
  Fp<L, V> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = dpp_swap(a.l[i]);
  return r;
}

template <int L, int V>
struct Fp2 {
  Fp<L, V> v;                          // this lane's coefficient
  Fp2() = default;
  BLS_HD explicit Fp2(const Fp<L, V>& a) : v(a) {}
  template <int L2, int V2>
  BLS_HD Fp2(const Fp2<L2, V2>& o) : v(o.v) {}
};
using Fp2R = Fp2<1, 2>;

template <int L, int V> BLS_HD Fp2<L, V> wrap2(const Fp<L, V>& a) { return Fp2<L, V>(a); }

// select between two Fp of different bounds (result carries the looser bounds)
template <int La, int Va, int Lb, int Vb>
BLS_HD auto sel(bool c, const Fp<La, Va>& a, const Fp<Lb, Vb>& b) {
  constexpr int L = La > Lb ? La : Lb;
  constexpr int V = Va > Vb ? Va : Vb;
  Fp<L, V> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

template <int La, int Va, int Lb, int Vb>
BLS_HD auto add(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return wrap2(add(a.v, b.v)); }
template <int La, int Va, int Lb, int Vb>
BLS_HD auto sub(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return wrap2(sub(a.v, b.v)); }
template <int L, int V>
BLS_HD auto neg(const Fp2<L, V>& a) { return wrap2(neg(a.v)); }
template <int L, int V>
BLS_HD auto dbl(const Fp2<L, V>& a) { return wrap2(dbl(a.v)); }
template <int LT, int VT, int L, int V>
BLS_HD auto fit(const Fp2<L, V>& a) { return wrap2(fit<LT, VT>(a.v)); }
template <int L, int V>
BLS_HD Fp2<L, V> select(bool c, const Fp2<L, V>& a, const Fp2<L, V>& b) { return Fp2<L, V>(select(c, a.v, b.v)); }

// Operand bounds the fused product accepts: the a-side c1 coefficient is negated on the even lane (L+1, V+1).
template <int La, int Va, int Lb, int Vb>
struct Fp2MulFits {
  static constexpr int ll = La * Lb + (La + 1) * Lb;
  static constexpr int vv = Va * Vb + (Va + 1) * Vb;
  static constexpr bool ok = ll <= MONT_LL_MAX && vv <= MONT_VV_MAX;
};

template <int La, int Va, int Lb, int Vb>
BLS_HD Fp2R mul_raw(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) {
  static_assert(Fp2MulFits<La, Va, Lb, Vb>::ok, "fp2 mul: operand bounds too large (fit<> them first)");
  const bool even = lane_part() == 0;
  const Fp<Lb, Vb> bo = swap(b.v);
  const Fp<La, Va> a0 = bcast0(a.v);                      // a0 on both lanes (one DPP move per limb, no select)
  const Fp<La, Va> a1 = bcast1(a.v);                      // a1 on both lanes
  const auto a1s = sel(even, neg(a1), a1);                // -a1 (even) / +a1 (odd)
  return Fp2R(mul2(a0, b.v, a1s, bo));                    // even: a0 b0 - a1 b1 ; odd: a0 b1 + a1 b0
}

template <int La, int Va, int Lb, int Vb>
BLS_HD Fp2R mul(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) {
  if constexpr (Fp2MulFits<La, Va, Lb, Vb>::ok) return mul_raw(a, b);
  else if constexpr (Fp2MulFits<Lb, Vb, La, Va>::ok) return mul_raw(b, a);      // negate the other side instead
  else if constexpr (Va * Vb + (Va + 1) * Vb > MONT_VV_MAX) {
    if constexpr (Va >= Vb) return mul(fit<1, 2>(a), b); else return mul(a, fit<1, 2>(b));
  } else {
    if constexpr (La >= Lb) return mul(fit<1, Va>(a), b); else return mul(a, fit<1, Vb>(b));
  }
}

// Lazy sums of Fp2 products and squares (fp.hip.h `dot_terms`): one reduction per lane for
//     dot2(pr2(A1, B1), sq2(A2), ...) = A1 B1 + A2^2 + ...
// A product contributes the two Fp products of mul_raw on each lane -- even: a0 b0 + (-a1) b1 ; odd: a0 b1 + a1 b0 --, a square
// the one of sqr -- even: (a0 + a1)(a0 - a1) ; odd: (2 a0) a1.
template <class A, class B> struct Fp2Prod { const A& a; const B& b; };
template <int La, int Va, int Lb, int Vb>
BLS_HD Fp2Prod<Fp2<La, Va>, Fp2<Lb, Vb>> pr2(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return Fp2Prod<Fp2<La, Va>, Fp2<Lb, Vb>>{a, b}; }
template <class A> struct Fp2Sq { const A& a; };
template <int L, int V>
BLS_HD Fp2Sq<Fp2<L, V>> sq2(const Fp2<L, V>& a) { return Fp2Sq<Fp2<L, V>>{a}; }

template <class A0, class A1, class B0> struct Fp2MulTerm {
  A0 a0; A1 a1s; B0 b, bo;
  static constexpr int ll = A0::kL * B0::kL + A1::kL * B0::kL, vv = A0::kV * B0::kV + A1::kV * B0::kV;
  template <int C> BLS_HD void accumulate(uint64_t (&acc)[2 * NL - 1]) const {
    mont_acc_product(acc, a0.l, b.l);
    mont_acc_product(acc, a1s.l, bo.l);
  }
};
template <class X, class Y> struct Fp2SqTerm {
  X x; Y y;
  static constexpr int ll = X::kL * Y::kL, vv = X::kV * Y::kV;
  template <int C> BLS_HD void accumulate(uint64_t (&acc)[2 * NL - 1]) const { mont_acc_product(acc, x.l, y.l); }
};
template <int La, int Va, int Lb, int Vb>
BLS_HD auto fp2_term(const Fp2Prod<Fp2<La, Va>, Fp2<Lb, Vb>>& p) {
  const bool even = lane_part() == 0;
  const Fp<La, Va> a1 = bcast1(p.a.v);
  const auto a1s = sel(even, neg(a1), a1);
  return Fp2MulTerm<Fp<La, Va>, decltype(sel(even, neg(a1), a1)), Fp<Lb, Vb>>{bcast0(p.a.v), a1s, p.b.v, swap(p.b.v)};
}
template <int L, int V>
BLS_HD auto fp2_term(const Fp2Sq<Fp2<L, V>>& s) {
  const bool even = lane_part() == 0;
  const Fp<L, V> o = swap(s.a.v);
  const auto x = add(bcast0(s.a.v), o);
  const auto y = sel(even, sub(s.a.v, o), s.a.v);
  return Fp2SqTerm<decltype(add(bcast0(s.a.v), o)), decltype(sel(even, sub(s.a.v, o), s.a.v))>{x, y};
}
template <class... T>
BLS_HD Fp2R dot2(const T&... t) { return Fp2R(dot_terms<0>(fp2_term(t)...)); }

// even: (a0 + a1)(a0 - a1) ; odd: (a0 + a0) a1      -- fq2_target_tree.rs:80-91
template <int L, int V>
BLS_HD Fp2R sqr(const Fp2<L, V>& a) {
  if constexpr ((2 * L) * (2 * L + 1) <= MONT_LL_MAX && (2 * V) * (2 * V + 1) <= MONT_VV_MAX) {
    const bool even = lane_part() == 0;
    const Fp<L, V> o = swap(a.v);                          // even: a1 ; odd: a0
    const auto x = add(bcast0(a.v), o);                    // even: a0 + a1 ; odd: a0 + a0
    const auto y = sel(even, sub(a.v, o), a.v);            // even: a0 - a1 ; odd: a1
    return Fp2R(mul(x, y));
  } else if constexpr ((2 * V) * (2 * V + 1) > MONT_VV_MAX) {
    return sqr(fit<1, 2>(a));
  } else {
    return sqr(fit<1, V>(a));
  }
}

template <int L, int V, int Lk, int Vk>
BLS_HD Fp2R mul_fp(const Fp2<L, V>& a, const Fp<Lk, Vk>& k) { return Fp2R(mul(a.v, k)); }

template <int L, int V>
BLS_HD auto conj(const Fp2<L, V>& a) { return wrap2(sel(lane_part() == 0, a.v, neg(a.v))); }

// xi = 1 + u: (c0 - c1, c0 + c1)
template <int L, int V>
BLS_HD auto mul_xi(const Fp2<L, V>& a) {
  const Fp<L, V> o = swap(a.v);
  return wrap2(sel(lane_part() == 0, sub(a.v, o), add(a.v, o)));
}

// norm a0^2 + a1^2 (an Fp value, identical on both lanes) and the inverse conj(a) / norm given 1 / norm: the two halves
// of fq2_target_tree.rs:66-78, split so that the one Fp inversion underneath can be batched across elements
// (final exponentiation, k_fexp.hip).
BLS_HD FpR fp2_norm(const Fp2R& a) {
  const FpR o = swap(a.v);
  return mul2(a.v, a.v, o, o);
}
BLS_HD Fp2R fp2_inv_given(const Fp2R& a, const FpR& norm_inv) {
  return Fp2R(mul(sel(lane_part() == 0, a.v, neg(a.v)), norm_inv));
}

// inverse via the norm; both lanes compute the (identical) Fp inverse.  Zero maps to zero.
template <int L, int V>
BLS_HD_NOINLINE Fp2R inv(const Fp2<L, V>& a_in) {
  const Fp2R a = fit<1, 2>(a_in);
  return fp2_inv_given(a, fp_inv(fp2_norm(a)));
}

// Fp2 number `idx` of this element in a LaneScratch: each lane of the pair keeps its own coefficient (one Fp per Fp2)
constexpr int SCRATCH_FP_PER_FP2 = 1;
BLS_HD void fp2_scratch_put(const LaneScratch& s, int idx, const Fp2R& a) { s.put(idx, a.v); }
BLS_HD Fp2R fp2_scratch_get(const LaneScratch& s, int idx) { return Fp2R(s.get(idx)); }

// Fp2 number `idx` of this lane in a per-lane slot store with put(slot, FpR) / get(slot) (LaneLds, pairing.hip.h): one Fp per lane
template <class Store> BLS_HD void fp2_slot_put(const Store& s, int idx, const Fp2R& a) { s.put(idx, a.v); }
template <class Store> BLS_HD Fp2R fp2_slot_get(const Store& s, int idx) { return Fp2R(s.get(idx)); }

// pair-wide AND of a per-lane predicate
BLS_HD bool pair_all(bool z) { return (dpp_swap(z ? 1u : 0u) & (z ? 1u : 0u)) != 0; }

template <int L, int V>
BLS_HD bool is_zero(const Fp2<L, V>& a) { return pair_all(is_zero(a.v)); }

BLS_HD Fp2<1, 1> fp2_zero() { return Fp2<1, 1>(fp_zero()); }
BLS_HD Fp2<1, 1> fp2_one() { return Fp2<1, 1>(select(lane_part() == 0, fp_one(), fp_zero())); }
BLS_HD Fp2<1, 1> fp2_const(const uint32_t (&c0)[NL], const uint32_t (&c1)[NL]) {
  return Fp2<1, 1>(select(lane_part() == 0, fp_const(c0), fp_const(c1)));
}

// each lane moves its own coefficient (48 contiguous bytes)
BLS_HD Fp2R fp2_load(const uint64_t* w, int form) { return Fp2R(fp_load(w + 6 * lane_part(), form)); }
template <int L, int V>
BLS_HD void fp2_store(uint64_t* w, const Fp2<L, V>& a, int form) { fp_store(w + 6 * lane_part(), a.v, form); }

// both coefficients as Fp values on BOTH lanes of the pair (for Fp-only sub-computations such as norms and roots)
template <int L, int V> BLS_HD Fp<L, V> coeff0(const Fp2<L, V>& a) { return select(lane_part() == 0, a.v, swap(a.v)); }
template <int L, int V> BLS_HD Fp<L, V> coeff1(const Fp2<L, V>& a) { return select(lane_part() == 0, swap(a.v), a.v); }
template <int L0, int V0, int L1, int V1>
BLS_HD auto fp2_from_coeffs(const Fp<L0, V0>& c0, const Fp<L1, V1>& c1) { return wrap2(sel(lane_part() == 0, c0, c1)); }
// coefficients stored at two unrelated addresses (MyFq12 / MyFq6 flat orders, src/fields/helpers.rs:39-41)
BLS_HD Fp2R fp2_load2(const uint64_t* w0, const uint64_t* w1, int form) { return Fp2R(fp_load(lane_part() == 0 ? w0 : w1, form)); }
template <int L, int V>
BLS_HD void fp2_store2(uint64_t* w0, uint64_t* w1, const Fp2<L, V>& a, int form) { fp_store(lane_part() == 0 ? w0 : w1, a.v, form); }

}  // namespace bls
