// Fp6 -> Fp12 tower for the MI355X pairing engine, on top of an Fp2 backend (fp2_pair.hip.h: one pairing per lane
// pair, the GPU default; fp2_serial.hip.h: one pairing per lane, used by the host build).
//
// Replaces the tower arithmetic the reference takes from ark-ff `Fq2/Fq6/Fq12` [third party]; formulas
// follow the reference's written statements of the tower:
//   Fp2  = Fp[u]/(u^2+1)        src/fields_as_trees/fq2_target_tree.rs:54-142
//   Fp6  = Fp2[v]/(v^3-(1+u))   src/fields_as_trees/fq6_target_tree.rs:59-293
//   Fp12 = Fp6[w]/(w^2-v)       src/fields_as_trees/fq12_target_tree.rs:53-176
// Field values are canonical at the ABI, so only the algebraic value matters for bit-exactness; inside,
// every Fp2 product is two fused multiply-pairs with a single Montgomery reduction each (fp.hip.h) -- one per lane
// of the pair in the lane-pair layout.
//
// Bounds <L, V> (limb / value, see fp.hip.h) are carried through every type.  Functions whose results are
// kept as loop state return "R" types (<1, 2>) by ending in fit<1, 2>().
#pragma once
#include "fp.hip.h"
#ifndef BLS_LANES
#define BLS_LANES 1
#endif
#if BLS_LANES == 2
#include "fp2_pair.hip.h"
#else
#include "fp2_serial.hip.h"
#endif

// Inlining granularity of the tower (tunable: code size vs call traffic).  Serial backend: Fp2 products inline
// into Fp6-level routines, Fp6/Fp12-level routines are real functions.  Lane-pair backend: everything inlines
// into the (single-copy) hot loops; cold glue goes through the *_call wrappers below.
#ifndef BLS_INLINE_FP6
#define BLS_INLINE_FP6 (BLS_LANES == 2)
#endif
#ifndef BLS_INLINE_FP12
#define BLS_INLINE_FP12 (BLS_LANES == 2)
#endif
#if BLS_INLINE_FP6
#define BLS_FN_FP6 BLS_HD
#else
#define BLS_FN_FP6 BLS_HD_NOINLINE
#endif
#if BLS_INLINE_FP12
#define BLS_FN_FP12 BLS_HD
#else
#define BLS_FN_FP12 BLS_HD_NOINLINE
#endif
// The *_call wrappers of cold glue: real functions by default (one shared copy; arguments and results travel through the
// stack, i.e. scratch).  A translation unit compiled for one wave per SIMD can afford to inline them (BLS_COLD_INLINE=1): the
// Fq12 values then stay in registers (k_fexp_wide.hip).
#ifndef BLS_COLD_INLINE
#define BLS_COLD_INLINE 0
#endif
#if BLS_COLD_INLINE
#define BLS_FN_COLD BLS_HD
#else
#define BLS_FN_COLD BLS_HD_NOINLINE
#endif

namespace bls {

// ================================================================================================ Fp6
template <int L, int V>
struct Fp6 {
  Fp2<L, V> c0, c1, c2;
  Fp6() = default;
  BLS_HD Fp6(const Fp2<L, V>& a, const Fp2<L, V>& b, const Fp2<L, V>& c) : c0(a), c1(b), c2(c) {}
  template <int L2, int V2>
  BLS_HD Fp6(const Fp6<L2, V2>& o) : c0(o.c0), c1(o.c1), c2(o.c2) {}
};
using Fp6R = Fp6<1, 2>;

// helper: build an Fp6 from three Fp2 of different bounds (bounds = component-wise max)
template <int L0, int V0, int L1, int V1, int L2, int V2>
BLS_HD auto make_fp6(const Fp2<L0, V0>& a, const Fp2<L1, V1>& b, const Fp2<L2, V2>& c) {
  constexpr int L = (L0 > L1 ? (L0 > L2 ? L0 : L2) : (L1 > L2 ? L1 : L2));
  constexpr int V = (V0 > V1 ? (V0 > V2 ? V0 : V2) : (V1 > V2 ? V1 : V2));
  return Fp6<L, V>(Fp2<L, V>(a), Fp2<L, V>(b), Fp2<L, V>(c));
}
template <int La, int Va, int Lb, int Vb>
BLS_HD auto add(const Fp6<La, Va>& a, const Fp6<Lb, Vb>& b) {
  return make_fp6(add(a.c0, b.c0), add(a.c1, b.c1), add(a.c2, b.c2));
}
template <int La, int Va, int Lb, int Vb>
BLS_HD auto sub(const Fp6<La, Va>& a, const Fp6<Lb, Vb>& b) {
  return make_fp6(sub(a.c0, b.c0), sub(a.c1, b.c1), sub(a.c2, b.c2));
}
template <int L, int V>
BLS_HD auto neg(const Fp6<L, V>& a) { return make_fp6(neg(a.c0), neg(a.c1), neg(a.c2)); }
template <int L, int V>
BLS_HD auto dbl(const Fp6<L, V>& a) { return make_fp6(dbl(a.c0), dbl(a.c1), dbl(a.c2)); }
template <int LT, int VT, int L, int V>
BLS_HD auto fit(const Fp6<L, V>& a) { return make_fp6(fit<LT, VT>(a.c0), fit<LT, VT>(a.c1), fit<LT, VT>(a.c2)); }

// multiply by v: (a, b, c) -> (c xi, a, b)   -- fq6_target_tree.rs:219-230
template <int L, int V>
BLS_HD auto mul_v(const Fp6<L, V>& a) { return make_fp6(mul_xi(a.c2), a.c0, a.c1); }

// Karatsuba, 6 Fp2 products   -- fq6_target_tree.rs:172-214
template <int La, int Va, int Lb, int Vb>
BLS_FN_FP6 auto mul(const Fp6<La, Va>& a, const Fp6<Lb, Vb>& b) {
  const Fp2R aa = mul(a.c0, b.c0), bb = mul(a.c1, b.c1), cc = mul(a.c2, b.c2);
  const Fp2R x12 = mul(add(a.c1, a.c2), add(b.c1, b.c2));
  const Fp2R x02 = mul(add(a.c0, a.c2), add(b.c0, b.c2));
  const Fp2R x01 = mul(add(a.c0, a.c1), add(b.c0, b.c1));
  auto t1 = add(mul_xi(fit<1, 64>(sub(sub(x12, bb), cc))), aa);
  auto t3 = sub(add(sub(x02, aa), bb), cc);
  auto t2 = add(sub(sub(x01, aa), bb), mul_xi(cc));
  return fit<1, 64>(make_fp6(t1, t2, t3));
}

// CH-SQR2   -- fq6_target_tree.rs:270-293
template <int L, int V>
BLS_FN_FP6 auto sqr(const Fp6<L, V>& a) {
  const Fp2R s0 = sqr(a.c0);
  const Fp2R ab = mul(a.c0, a.c1);
  const auto s1 = dbl(ab);
  const Fp2R s2 = sqr(add(sub(a.c0, a.c1), a.c2));
  const Fp2R bc = mul(a.c1, a.c2);
  const auto s3 = dbl(bc);
  const Fp2R s4 = sqr(a.c2);
  auto c0 = add(mul_xi(s3), s0);
  auto c1 = add(mul_xi(s4), s1);
  auto c2 = sub(sub(add(add(s1, s2), s3), s0), s4);
  return fit<1, 64>(make_fp6(c0, c1, c2));
}

// sparse (c0, c1, 0)   -- fq6_target_tree.rs:232-259
template <int L, int V, int L0, int V0, int L1, int V1>
BLS_FN_FP6 auto mul_by_01(const Fp6<L, V>& s, const Fp2<L0, V0>& c0, const Fp2<L1, V1>& c1) {
  const Fp2R aa = mul(s.c0, c0), bb = mul(s.c1, c1);
  auto t1 = add(mul_xi(mul(s.c2, c1)), aa);
  auto t2 = sub(sub(mul(add(c0, c1), add(s.c0, s.c1)), aa), bb);
  auto t3 = add(mul(s.c2, c0), bb);
  return fit<1, 64>(make_fp6(t1, t2, t3));
}

// sparse (0, c1, 0)   -- fq6_target_tree.rs:261-268
template <int L, int V, int L1, int V1>
BLS_FN_FP6 auto mul_by_1(const Fp6<L, V>& s, const Fp2<L1, V1>& c1) {
  return fit<1, 64>(make_fp6(mul_xi(mul(s.c2, c1)), mul(s.c0, c1), mul(s.c1, c1)));
}

// inverse   -- fq6_target_tree.rs:59-89 ; zero maps to zero.  Written as begin / end around the single Fp inversion
// it bottoms out in (the norm of `t`), so that callers can supply that inverse from a batched inversion.
struct Fp6InvState { Fp2R c0, c1, c2, t; };
BLS_HD FpR fp6_inv_begin(const Fp6R& a, Fp6InvState& st) {
  st.c0 = fit<1, 2>(sub(sqr(a.c0), mul_xi(mul(a.c1, a.c2))));
  st.c1 = fit<1, 2>(sub(mul_xi(sqr(a.c2)), mul(a.c0, a.c1)));
  st.c2 = fit<1, 2>(sub(sqr(a.c1), mul(a.c0, a.c2)));
  auto tmp = add(mul(a.c1, st.c2), mul(a.c2, st.c1));
  st.t = fit<1, 2>(add(mul_xi(tmp), mul(a.c0, st.c0)));
  return fp2_norm(st.t);
}
BLS_HD Fp6R fp6_inv_end(const Fp6InvState& st, const FpR& norm_inv) {
  const Fp2R ti = fp2_inv_given(st.t, norm_inv);
  return Fp6R(mul(st.c0, ti), mul(st.c1, ti), mul(st.c2, ti));
}
template <int L, int V>
BLS_HD_NOINLINE Fp6R inv(const Fp6<L, V>& a_in) {
  Fp6InvState st;
  const FpR n = fp6_inv_begin(fit<1, 2>(a_in), st);
  return fp6_inv_end(st, fp_inv(n));
}

BLS_HD Fp6<1, 1> fp6_zero() { return Fp6<1, 1>(fp2_zero(), fp2_zero(), fp2_zero()); }
BLS_HD Fp6<1, 1> fp6_one() { return Fp6<1, 1>(fp2_one(), fp2_zero(), fp2_zero()); }

// Frobenius   -- fq6_target_tree.rs:129-169, constants canonical (fp_consts.h)
template <int L, int V>
BLS_HD Fp6R frobenius(const Fp6<L, V>& a) {
  const Fp2<1, 1> k1 = fp2_const(BLS_FROB6_C1_C0, BLS_FROB6_C1_C1);
  const Fp2<1, 1> k2 = fp2_const(BLS_FROB6_C2_C0, BLS_FROB6_C2_C1);
  return Fp6R(fit<1, 2>(conj(a.c0)), mul(conj(a.c1), k1), mul(conj(a.c2), k2));
}

// =============================================================================================== Fp12
template <int L, int V>
struct Fp12 {
  Fp6<L, V> c0, c1;
  Fp12() = default;
  BLS_HD Fp12(const Fp6<L, V>& a, const Fp6<L, V>& b) : c0(a), c1(b) {}
  template <int L2, int V2>
  BLS_HD Fp12(const Fp12<L2, V2>& o) : c0(o.c0), c1(o.c1) {}
};
using Fp12R = Fp12<1, 2>;

template <int L0, int V0, int L1, int V1>
BLS_HD auto make_fp12(const Fp6<L0, V0>& a, const Fp6<L1, V1>& b) {
  constexpr int L = L0 > L1 ? L0 : L1;
  constexpr int V = V0 > V1 ? V0 : V1;
  return Fp12<L, V>(Fp6<L, V>(a), Fp6<L, V>(b));
}
template <int LT, int VT, int L, int V>
BLS_HD auto fit(const Fp12<L, V>& a) {
  auto x = fit<LT, VT>(a.c0); auto y = fit<LT, VT>(a.c1);
  return make_fp12(x, y);
}

BLS_HD Fp12<1, 1> fp12_one() { return Fp12<1, 1>(fp6_one(), fp6_zero()); }

// Karatsuba over Fp6   -- fq12_target_tree.rs:130-141
template <int La, int Va, int Lb, int Vb>
BLS_FN_FP12 Fp12R mul(const Fp12<La, Va>& a, const Fp12<Lb, Vb>& b) {
  const auto aa = mul(a.c0, b.c0);
  const auto bb = mul(a.c1, b.c1);
  const auto x = mul(fit<1, 64>(add(a.c1, a.c0)), fit<1, 64>(add(b.c0, b.c1)));
  const auto c1 = sub(sub(x, aa), bb);          // value bounds stay <= 64: no intermediate reduction needed
  const auto c0 = add(mul_v(bb), aa);
  return Fp12R(fit<1, 2>(c0), fit<1, 2>(c1));
}

// complex squaring   -- fq12_target_tree.rs:143-155
template <int L, int V>
BLS_FN_FP12 Fp12R sqr(const Fp12<L, V>& a) {
  const auto ab = fit<1, 2>(mul(a.c0, a.c1));
  const auto s = fit<1, 64>(add(a.c0, a.c1));
  const auto t = fit<1, 64>(add(mul_v(a.c1), a.c0));
  const auto m = mul(t, s);
  const auto c0 = sub(sub(m, ab), mul_v(ab));
  const auto c1 = dbl(ab);
  return Fp12R(fit<1, 2>(c0), fit<1, 2>(c1));
}

template <int L, int V>
BLS_HD auto conj(const Fp12<L, V>& a) { return make_fp12(a.c0, neg(a.c1)); }    // fq12_target_tree.rs:53-58

// inverse   -- fq12_target_tree.rs:77-90 ; zero maps to zero.  Same begin / end split as Fp6: `fp12_inv_norm` is the
// Fp value whose inverse the whole computation needs (N_{Fp2/Fp} . N_{Fp6/Fp2} . N_{Fp12/Fp6} of a).
BLS_HD FpR fp12_inv_begin(const Fp12R& a, Fp6InvState& st) {
  const auto t = sub(fit<1, 2>(sqr(a.c0)), fit<1, 2>(mul_v(fit<1, 2>(sqr(a.c1)))));
  return fp6_inv_begin(fit<1, 2>(t), st);
}
BLS_HD Fp12R fp12_inv_end(const Fp12R& a, const Fp6InvState& st, const FpR& norm_inv) {
  const Fp6R ti = fp6_inv_end(st, norm_inv);
  return Fp12R(fit<1, 2>(mul(a.c0, ti)), fit<1, 2>(mul(a.c1, neg(ti))));
}
template <int L, int V>
BLS_HD_NOINLINE Fp12R inv(const Fp12<L, V>& a_in) {
  const Fp12R a = fit<1, 2>(a_in);
  Fp6InvState st;
  const FpR n = fp12_inv_begin(a, st);
  return fp12_inv_end(a, st, fp_inv(n));
}
// out-of-line halves for the split final exponentiation (k_fexp.hip)
BLS_HD_NOINLINE FpR fp12_inv_norm_call(const Fp12R& a) { Fp6InvState st; return fp12_inv_begin(a, st); }
BLS_HD_NOINLINE Fp12R fp12_inv_given_call(const Fp12R& a, const FpR& norm_inv) {
  Fp6InvState st;
  (void)fp12_inv_begin(a, st);
  return fp12_inv_end(a, st, norm_inv);
}

// Frobenius   -- fq12_target_tree.rs:92-128
template <int L, int V>
BLS_FN_FP12 Fp12R frobenius(const Fp12<L, V>& a) {
  const Fp2<1, 1> k = fp2_const(BLS_FROB12_C1_C0, BLS_FROB12_C1_C1);
  const Fp6R c0 = frobenius(a.c0);
  const Fp6R c1 = frobenius(a.c1);
  return Fp12R(c0, Fp6R(mul(c1.c0, k), mul(c1.c1, k), mul(c1.c2, k)));
}

// sparse multiplication by (c0, c1, 0 | 0, c4, 0)   -- fq12_target_tree.rs:157-176 (commented twin at
// src/miller_loop_native.rs:118-137)
// BLS_LAZY_014 = 1: the same product written coefficient by coefficient as sums of three Fp2 products with ONE reduction
// each (fp.hip.h `dot`): 18 products / 6 reductions per lane instead of Karatsuba's 13 / 13, no pre/post additions and no final
// conditioning (every coefficient leaves its reduction < 2p).  8 % more multiply-adds, ~15-19 % fewer instructions: a win where
// every instruction costs the same issue slot (the Miller kernels, one wave per SIMD), not at two waves per SIMD.
//   (f0 + f1 w)(A + B w),  A = c0 + c1 v,  B = c4 v,  w^2 = v,  v^3 = xi:
//   r0 = f0 A + v f1 B = [f00 c0 + xi f02 c1 + xi f11 c4,  f00 c1 + f01 c0 + xi f12 c4,  f01 c1 + f02 c0 + f10 c4]
//   r1 = f1 A +   f0 B = [f10 c0 + xi f12 c1 + xi f02 c4,  f10 c1 + f11 c0 + f00 c4,     f11 c1 + f12 c0 + f01 c4]
#ifndef BLS_LAZY_014
#define BLS_LAZY_014 0
#endif
#ifndef BLS_LAZY_014_ORDER      // order in which the six coefficients are computed (register pressure; tools/kbench A/B)
#define BLS_LAZY_014_ORDER 0
#endif
template <int L, int V, int L0, int V0, int L1, int V1, int L4, int V4>
BLS_HD Fp12R mul_by_014_lazy(const Fp12<L, V>& f, const Fp2<L0, V0>& c0, const Fp2<L1, V1>& c1, const Fp2<L4, V4>& c4) {
  const auto xc1 = fit<1, 64>(mul_xi(c1));
  const auto xc4 = fit<1, 64>(mul_xi(c4));
  const auto& f00 = f.c0.c0; const auto& f01 = f.c0.c1; const auto& f02 = f.c0.c2;
  const auto& f10 = f.c1.c0; const auto& f11 = f.c1.c1; const auto& f12 = f.c1.c2;
  Fp12R r;
#if BLS_LAZY_014_ORDER == 0
  // the three coefficients that read xi c1 / xi c4 first, then the three that read c4: four line operands live at a time
  r.c0.c0 = dot2(pr2(c0, f00), pr2(xc1, f02), pr2(xc4, f11));
  r.c1.c0 = dot2(pr2(c0, f10), pr2(xc1, f12), pr2(xc4, f02));
  r.c0.c1 = dot2(pr2(c1, f00), pr2(c0, f01), pr2(xc4, f12));
  r.c0.c2 = dot2(pr2(c1, f01), pr2(c0, f02), pr2(c4, f10));
  r.c1.c1 = dot2(pr2(c1, f10), pr2(c0, f11), pr2(c4, f00));
  r.c1.c2 = dot2(pr2(c1, f11), pr2(c0, f12), pr2(c4, f01));
#else
  r.c0.c0 = dot2(pr2(c0, f00), pr2(xc1, f02), pr2(xc4, f11));
  r.c0.c1 = dot2(pr2(c1, f00), pr2(c0, f01), pr2(xc4, f12));
  r.c0.c2 = dot2(pr2(c1, f01), pr2(c0, f02), pr2(c4, f10));
  r.c1.c0 = dot2(pr2(c0, f10), pr2(xc1, f12), pr2(xc4, f02));
  r.c1.c1 = dot2(pr2(c1, f10), pr2(c0, f11), pr2(c4, f00));
  r.c1.c2 = dot2(pr2(c1, f11), pr2(c0, f12), pr2(c4, f01));
#endif
  return r;
}

template <int L, int V, int L0, int V0, int L1, int V1, int L4, int V4>
BLS_FN_FP12 Fp12R mul_by_014(const Fp12<L, V>& f, const Fp2<L0, V0>& c0, const Fp2<L1, V1>& c1, const Fp2<L4, V4>& c4) {
#if BLS_LAZY_014
  return mul_by_014_lazy(f, c0, c1, c4);
#else
  const auto aa = mul_by_01(f.c0, c0, c1);
  const auto bb = mul_by_1(f.c1, c4);
  const auto o = add(c1, c4);
  const auto x = mul_by_01(fit<1, 64>(add(f.c1, f.c0)), c0, o);
  const auto r1 = sub(sub(x, aa), bb);
  const auto r0 = add(mul_v(bb), aa);
  return Fp12R(fit<1, 2>(r0), fit<1, 2>(r1));
#endif
}

// Out-of-line entry points for cold call sites (final-exponentiation glue, reference-mode accumulators): one
// shared copy of each routine however BLS_FN_FP12 is set.
BLS_FN_COLD Fp12R mul_call(const Fp12R& a, const Fp12R& b) { return mul(a, b); }
BLS_FN_COLD Fp12R sqr_call(const Fp12R& a) { return sqr(a); }
BLS_FN_COLD Fp12R frobenius_call(const Fp12R& a) { return frobenius(a); }

// ---- cyclotomic subgroup (final exponentiation)   -- src/fields_as_trees/miller_loop.rs:29-126
template <class A, class B> struct Pair2 { A a; B b; };
template <class A, class B> BLS_HD Pair2<A, B> pair2(const A& a, const B& b) { return Pair2<A, B>{a, b}; }

// (a + b s)^2 in Fp4 = Fp2[s]/(s^2 - xi)   -- miller_loop.rs:29-44
template <int La, int Va, int Lb, int Vb>
BLS_HD auto fp4_square(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) {
  const Fp2R t0 = sqr(a), t1 = sqr(b);
  const auto c0 = add(mul_xi(t1), t0);
  const Fp2R t2 = sqr(add(a, b));
  const auto c1 = sub(sub(t2, t0), t1);
  return pair2(c0, c1);
}

// (x0 + x1 s)(y0 + y1 s) in Fp4, Karatsuba: three Fp2 products
template <class X0, class X1, class Y0, class Y1>
BLS_HD auto fp4_mul(const X0& x0, const X1& x1, const Y0& y0, const Y1& y1) {
  const Fp2R v0 = mul(x0, y0), v1 = mul(x1, y1);
  const auto c0 = add(mul_xi(v1), v0);
  const Fp2R m = mul(add(x0, x1), add(y0, y1));
  const auto c1 = sub(sub(m, v0), v1);
  return pair2(c0, c1);
}

// TWICE the square, 2 f^2, by the Chung-Hasan SQR3 formula over Fp4: with s = w^3 (s^2 = xi) and t = w (t^3 = s),
//   f = A + B t + C t^2,   A = g0 + g3 s,  B = g1 + g4 s,  C = g2 + g5 s      (g_i = coefficient of w^i: c0 = (g0, g2, g4), c1 = (g1, g3, g5))
//   S0 = A^2,  S1 = (A + B + C)^2,  S2 = (A - B + C)^2,  S3 = B C,  S4 = C^2
//   2 f^2 = (2 S0 + 4 s S3) + (S1 - S2 - 4 S3 + 2 s S4) t + (S1 + S2 - 2 S0 - 2 S4) t^2
// Four Fp4 squarings (three Fp2 squarings each) and one Fp4 product (three Fp2 products): 12 x 392 + 3 x 588 = 6 468 multiply-adds per
// lane against 12 x 588 = 7 056 for the complex squaring over two Karatsuba Fq6 products.  The factor 2 (SQR3's interpolation
// halves S1 + S2) is an element of Fp: a Miller value that goes through the final exponentiation may carry it, an exact square
// may not.  Measured against sqr() in the homogeneous-step Miller loop (pairing.hip.h BLS_HOM_SQR3): fewer multiply-adds, more
// additions and carry passes, 3 % slower -- kept as a checked routine (host build: tests/hostsim) and an A/B switch, not used by default.
template <int L, int V>
BLS_HD Fp12R sqr_twice(const Fp12<L, V>& f) {
  const auto& g0 = f.c0.c0; const auto& g2 = f.c0.c1; const auto& g4 = f.c0.c2;
  const auto& g1 = f.c1.c0; const auto& g3 = f.c1.c1; const auto& g5 = f.c1.c2;
  const auto S0 = fp4_square(g0, g3);
  const auto S4 = fp4_square(g2, g5);
  const auto ac0 = add(g0, g2);                                            // A + C
  const auto ac1 = add(g3, g5);
  const auto S1 = fp4_square(add(ac0, g1), add(ac1, g4));
  const auto S2 = fp4_square(sub(ac0, g1), sub(ac1, g4));
  const auto S3 = fp4_mul(g1, g4, g2, g5);
  // (add / sub / dbl condition their operands by themselves when a bound would leave the representable range, fp.hip.h)
  const auto s3a4 = dbl(dbl(S3.a));                                        // 4 S3
  const auto s3b4 = dbl(dbl(S3.b));
  // coefficient of 1:   2 S0 + 4 s S3 = (2 S0.a + 4 xi S3.b) + (2 S0.b + 4 S3.a) s
  const auto h0 = add(dbl(S0.a), mul_xi(s3b4));
  const auto h3 = add(dbl(S0.b), s3a4);
  // coefficient of t:   S1 - S2 - 4 S3 + 2 s S4 = (S1.a - S2.a - 4 S3.a + 2 xi S4.b) + (S1.b - S2.b - 4 S3.b + 2 S4.a) s
  const auto h1 = add(sub(sub(S1.a, S2.a), s3a4), mul_xi(dbl(S4.b)));
  const auto h4 = add(sub(sub(S1.b, S2.b), s3b4), dbl(S4.a));
  // coefficient of t^2: S1 + S2 - 2 S0 - 2 S4
  const auto h2 = sub(sub(add(S1.a, S2.a), dbl(S0.a)), dbl(S4.a));
  const auto h5 = sub(sub(add(S1.b, S2.b), dbl(S0.b)), dbl(S4.b));
  return Fp12R(Fp6R(fit<1, 2>(h0), fit<1, 2>(h2), fit<1, 2>(h4)), Fp6R(fit<1, 2>(h1), fit<1, 2>(h3), fit<1, 2>(h5)));
}

// Granger-Scott squaring   -- miller_loop.rs:46-104
template <int L, int V>
BLS_FN_FP12 Fp12R cyclotomic_square(const Fp12<L, V>& f) {
  const auto& z0 = f.c0.c0; const auto& z4 = f.c0.c1; const auto& z3 = f.c0.c2;
  const auto& z2 = f.c1.c0; const auto& z1 = f.c1.c1; const auto& z5 = f.c1.c2;
  const auto A = fp4_square(z0, z1);
  const auto n0 = add(dbl(sub(A.a, z0)), A.a);          // 2(t0 - z0) + t0
  const auto n1 = add(dbl(add(A.b, z1)), A.b);          // 2(t1 + z1) + t1
  const auto B = fp4_square(z2, z3);
  const auto C = fp4_square(z4, z5);
  const auto n4 = add(dbl(sub(B.a, z4)), B.a);
  const auto n5 = add(dbl(add(B.b, z5)), B.b);
  const auto t0 = mul_xi(fit<1, 2>(C.b));
  const auto n2 = add(dbl(add(t0, z2)), t0);
  const auto n3 = add(dbl(sub(C.a, z3)), C.a);
  return Fp12R(Fp6R(fit<1, 2>(n0), fit<1, 2>(n4), fit<1, 2>(n3)),
               Fp6R(fit<1, 2>(n2), fit<1, 2>(n1), fit<1, 2>(n5)));
}

BLS_FN_COLD Fp12R cyclotomic_square_call(const Fp12R& a) { return cyclotomic_square(a); }

// conj(f^|x|), MSB-first square-and-multiply over BLS_X   -- miller_loop.rs:106-126, with the product the
// reference drops at :121 kept.
BLS_HD_NOINLINE Fp12R cyclotomic_exp(const Fp12R& f) {
  // |x| = 0xd201000000010000 has its one bits at 63, 62, 60, 57, 48, 16: after the leading bit the walk is
  // 1, 2, 3, 9, 32 squarings each followed by a multiplication, then 16 squarings.  Writing it as runs keeps the
  // hot inner loop down to the cyclotomic squaring alone (small live set); the five multiplications sit outside.
  static_assert(BLS_X_PARAM == ((1ull << 63) | (1ull << 62) | (1ull << 60) | (1ull << 57) | (1ull << 48) | (1ull << 16)), "run table");
  constexpr uint64_t RUNS = 1ull | (2ull << 8) | (3ull << 16) | (9ull << 24) | (32ull << 32) | (16ull << 40);   // packed run lengths
  Fp12R tmp = f;                                           // leading one bit
  for (int s = 0; s < 6; ++s) {
    const int run = (int)((RUNS >> (8 * s)) & 0xffu);
    for (int i = 0; i < run; ++i) tmp = cyclotomic_square(tmp);
    // Out of line on purpose: (1) five calls per exponentiation cost nothing, (2) it keeps this function's body
    // within short-branch range and makes it a non-leaf.  hipcc (ROCm 7.2) relaxes an out-of-range branch in a LEAF
    // device function through s[30:31] -- the return address -- and the function then never returns
    // (tools/check_device_asm.py guards the build against that pattern).
    if (s < 5) tmp = mul_call(tmp, f);
  }
  return fit<1, 2>(conj(tmp));
}

// easy part + hard part   -- miller_loop.rs:166-178, :128-164 ; `ok` is false when f == 0 (no inverse)
// `f_inv` = f^-1: computed by the caller, either in place (final_exponentiation below) or from a batched inversion
BLS_HD_NOINLINE Fp12R final_exponentiation_given_inverse(const Fp12R& f, const Fp12R& f_inv) {
  const Fp12R t0f = fit<1, 2>(conj(f));                   // frobenius^6 == conjugation (:168-174)
  Fp12R t1 = f_inv;                                       // :175
  Fp12R t2 = mul_call(t0f, t1);                                // :133
  t1 = t2;
  t2 = mul_call(frobenius_call(frobenius_call(t2)), t1);                 // :135-136
  t1 = fit<1, 2>(conj(cyclotomic_square_call(t2)));       // :137
  Fp12R t3 = cyclotomic_exp(t2);
  Fp12R t4 = cyclotomic_square_call(t3);
  Fp12R t5 = mul_call(t1, t3);
  t1 = cyclotomic_exp(t5);
  Fp12R t0 = cyclotomic_exp(t1);
  Fp12R t6 = cyclotomic_exp(t0);
  t6 = mul_call(t6, t4);
  t4 = cyclotomic_exp(t6);
  t5 = fit<1, 2>(conj(t5));
  t4 = mul_call(t4, mul_call(t5, t2));
  t5 = fit<1, 2>(conj(t2));
  t1 = mul_call(t1, t2);
  t1 = frobenius_call(frobenius_call(frobenius_call(t1)));
  t6 = mul_call(t6, t5);
  t6 = frobenius_call(t6);
  t3 = mul_call(t3, t0);
  t3 = frobenius_call(frobenius_call(t3));
  t3 = mul_call(t3, t1);
  t3 = mul_call(t3, t6);
  return mul_call(t3, t4);
}
BLS_HD_NOINLINE Fp12R final_exponentiation(const Fp12R& f) { return final_exponentiation_given_inverse(f, inv(f)); }

template <int L, int V>
BLS_HD bool is_zero(const Fp6<L, V>& a) { return is_zero(a.c0) & is_zero(a.c1) & is_zero(a.c2); }
template <int L, int V>
BLS_HD bool is_zero(const Fp12<L, V>& a) { return is_zero(a.c0) & is_zero(a.c1); }

BLS_HD Fp12R fp12_load(const uint64_t* w, int form) {
  Fp12R r;
  r.c0.c0 = fp2_load(w, form);      r.c0.c1 = fp2_load(w + 12, form); r.c0.c2 = fp2_load(w + 24, form);
  r.c1.c0 = fp2_load(w + 36, form); r.c1.c1 = fp2_load(w + 48, form); r.c1.c2 = fp2_load(w + 60, form);
  return r;
}
// arkworks in-memory coefficient order c0.c0.c0, c0.c0.c1, c0.c1.c0, ..., c1.c2.c1 (12 x 48 bytes)
template <int L, int V>
BLS_HD void fp12_store(uint64_t* w, const Fp12<L, V>& a, int form) {
  fp2_store(w, a.c0.c0, form);      fp2_store(w + 12, a.c0.c1, form); fp2_store(w + 24, a.c0.c2, form);
  fp2_store(w + 36, a.c1.c0, form); fp2_store(w + 48, a.c1.c1, form); fp2_store(w + 60, a.c1.c2, form);
}

}  // namespace bls
