// Curve steps, Miller loops and pairings for the MI355X engine (one pairing per lane pair, or per lane with the
// serial Fp2 backend).
//
// Two loop schedules over the same field kernels:
//   Mode A  `miller_ref_opt`  bit-exact restatement of the reference's
//           optimized_miller_loop(P: G1Projective, Q: G2Projective) -> Fq12
//           (src/miller_loop_native_optimized.rs:8-127) including its quirks: LSB-first walk of the loop
//           parameter, numerator/denominator pair, Jacobian coordinates read as homogeneous, values living
//           in the c0.c0 slot, "final exponentiation" = one squaring.
//   Mode B  `miller_ate` / `pairing`  the optimal-ate Miller loop the reference's
//           src/miller_loop_native.rs:27-116 (+ commented `ell`, :118-152) specifies, and the final
//           exponentiation of src/fields_as_trees/miller_loop.rs:128-178.
#pragma once
#include "tower.hip.h"

// up to this many pairs the multi-pairing loop inlines one copy of the step routines per pair (state stays in
// registers); beyond it the steps are calls
#ifndef BLS_INLINE_PAIRS_MAX
#define BLS_INLINE_PAIRS_MAX (BLS_LANES == 2 ? 3 : 1)
#endif

// Hook at the top of every Miller-loop iteration (experiments: e.g. a workgroup barrier that keeps the waves of a CU
// within one iteration of each other so that they share instruction-cache lines).  Empty by default.
#ifndef BLS_MILLER_PEEL          // 1: the first line of a Miller loop is copied into f instead of multiplied into f = 1
#define BLS_MILLER_PEEL 1
#endif
#ifndef BLS_MILLER_ITERATION_HOOK
#define BLS_MILLER_ITERATION_HOOK() ((void)0)
#endif

namespace bls {

enum : uint8_t { FLAG_DEN_ZERO = 1, FLAG_INFINITY = 2 };

template <class F> struct Jac { F x, y, z; };     // arkworks `Projective`: Jacobian (X, Y, Z), x = X/Z^2, y = Y/Z^3
using G2Jac = Jac<Fp2R>;
using G1Jac = Jac<FpR>;
struct G2Aff { Fp2R x, y; };
struct G1Aff { FpR x, y; };

// Exact zero / equality tests on field values (the reference compares canonical arkworks elements).
template <int La, int Va, int Lb, int Vb>
BLS_HD bool eq(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return is_zero(sub(a, b)); }
template <int La, int Va, int Lb, int Vb>
BLS_HD bool eq(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) { return fp_eq(a, b); }
template <int L, int V>
BLS_HD FpR inv(const Fp<L, V>& a) { return fp_inv(a); }

// field-generic constants for the curve routines below (F = FpR for G1, Fp2R for G2)
BLS_HD void set_one(FpR& r) { r = FpR(fp_one()); }
BLS_HD void set_one(Fp2R& r) { r = Fp2R(fp2_one()); }
BLS_HD void set_zero(FpR& r) { r = FpR(fp_zero()); }
BLS_HD void set_zero(Fp2R& r) { r = Fp2R(fp2_zero()); }
// curve constant times v: b = 4 on G1 (y^2 = x^3 + 4), b = 4 (1 + u) on the twist
template <int L, int V> BLS_HD auto mul_curve_b(const Fp<L, V>& v) { return dbl(dbl(v)); }
template <int L, int V> BLS_HD auto mul_curve_b(const Fp2<L, V>& v) { return dbl(dbl(mul_xi(v))); }

// products with automatic operand conditioning for either field (the Fp2 backends already condition; plain Fp
// products static_assert instead, so the field-generic curve routines go through these)
template <int La, int Va, int Lb, int Vb>
BLS_HD FpR fmul(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) {
  if constexpr (La * Lb <= MONT_LL_MAX && Va * Vb <= MONT_VV_MAX) return mul(a, b);
  else if constexpr (Va * Vb > MONT_VV_MAX) { if constexpr (Va >= Vb) return fmul(fit<1, 2>(a), b); else return fmul(a, fit<1, 2>(b)); }
  else { if constexpr (La >= Lb) return fmul(fit<1, Va>(a), b); else return fmul(a, fit<1, Vb>(b)); }
}
template <int L, int V>
BLS_HD FpR fsqr(const Fp<L, V>& a) {
  if constexpr (L * L <= MONT_LL_MAX && V * V <= MONT_VV_MAX) return sqr(a);
  else if constexpr (V * V > MONT_VV_MAX) return fsqr(fit<1, 2>(a));
  else return fsqr(fit<1, V>(a));
}
template <int La, int Va, int Lb, int Vb>
BLS_HD Fp2R fmul(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return mul(a, b); }
template <int L, int V>
BLS_HD Fp2R fsqr(const Fp2<L, V>& a) { return sqr(a); }

// ---------------------------------------------------------------------- Jacobian G1 / G2 (field-generic)
// ark-ec 0.4.2 `Projective::double_in_place`, a = 0 (EFD dbl-2009-l) [third party; SURVEY App. A.3]
template <class F>
BLS_HD Jac<F> jac_double(const Jac<F>& p) {
  if (is_zero(p.z)) return p;
  const F A = fsqr(p.x), B = fsqr(p.y), C = fsqr(B);
  const auto D = dbl(sub(sub(fsqr(add(p.x, B)), A), C));
  const auto E = add(dbl(A), A);
  const F Fv = fsqr(E);
  Jac<F> r;
  r.z = fit<1, 2>(dbl(fmul(p.y, p.z)));
  const auto X3 = fit<1, 2>(sub(sub(Fv, D), D));
  r.x = X3;
  r.y = fit<1, 2>(sub(fmul(E, sub(D, X3)), dbl(dbl(dbl(C)))));
  return r;
}

// BLS_FN_COLD (tower.hip.h): a real function by default; inlined in translation units compiled with BLS_COLD_INLINE=1, where arguments
// and results then stay in registers instead of travelling through call frames in the private segment
BLS_FN_COLD G2Jac jac_double_call(const G2Jac& p) { return jac_double(p); }
BLS_HD_NOINLINE G1Jac jac_double_call(const G1Jac& p) { return jac_double(p); }

// ark-ec 0.4.2 `impl AddAssign<&Self> for Projective` (EFD add-2007-bl) [third party; SURVEY App. A.3]
template <class F>
BLS_HD Jac<F> jac_add(const Jac<F>& p1, const Jac<F>& p2) {
  if (is_zero(p1.z)) return p2;
  if (is_zero(p2.z)) return p1;
  const F Z1Z1 = fsqr(p1.z), Z2Z2 = fsqr(p2.z);
  const F U1 = fmul(p1.x, Z2Z2), U2 = fmul(p2.x, Z1Z1);
  const F S1 = fmul(fmul(p1.y, p2.z), Z2Z2), S2 = fmul(fmul(p2.y, p1.z), Z1Z1);
  if (eq(U1, U2) && eq(S1, S2)) return jac_double_call(p1);
  const auto H = sub(U2, U1);
  const F I = fsqr(dbl(H));
  const F J = fmul(H, I);
  const auto r = dbl(sub(S2, S1));
  const F V = fmul(U1, I);
  Jac<F> o;
  const auto X3 = fit<1, 2>(sub(sub(fsqr(r), J), dbl(V)));
  o.x = X3;
  o.y = fit<1, 2>(sub(fmul(r, sub(V, X3)), dbl(fmul(S1, J))));
  o.z = fit<1, 2>(fmul(sub(sub(fsqr(add(p1.z, p2.z)), Z1Z1), Z2Z2), H));
  return o;
}
// `R + R` (opt.rs:93): arkworks' add_assign finds U1 == U2, S1 == S2 and falls through to double_in_place (infinity stays infinity on
// either path), so the doubling formulas ARE what the reference computes for this call; the addition formulas' common part
// (two squarings, six products spent on discovering that the operands are equal) is not repeated here.
BLS_HD G2Jac g2_jac_double(const G2Jac& p) { return jac_double_call(p); }
BLS_FN_COLD G2Jac g2_jac_add(const G2Jac& p1, const G2Jac& p2) { return jac_add(p1, p2); }

// Point helpers around the pairing (the reference's curve types: src/fields_as_trees/g2_curve.rs:29-166,
// g1_curve.rs:22-110; arkworks `Projective` equality / normalisation [third party]).
enum : int { POINT_ADD = 0, POINT_DOUBLE = 1, POINT_NORMALIZE = 2, POINT_EQ = 3, POINT_ON_CURVE = 4, POINT_IN_SUBGROUP = 5 };
// group order r (KAT-6: the decimal at src/miller_loop_native_optimized.rs:110), little-endian u64 words
BLS_CONST uint64_t BLS_R_ORDER[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
enum : uint8_t { FLAG_TRUE = 32 };

// `Projective == Projective`: both at infinity, or neither and X1 Z2^2 == X2 Z1^2, Y1 Z2^3 == Y2 Z1^3
template <class F>
BLS_HD bool jac_eq(const Jac<F>& a, const Jac<F>& b) {
  const bool za = is_zero(a.z), zb = is_zero(b.z);
  if (za || zb) return za && zb;
  const F z1z1 = fsqr(a.z), z2z2 = fsqr(b.z);
  return eq(fmul(a.x, z2z2), fmul(b.x, z1z1)) && eq(fmul(a.y, fmul(z2z2, b.z)), fmul(b.y, fmul(z1z1, a.z)));
}
// Y^2 == X^3 + b Z^6 (infinity counts as on the curve)
template <class F>
BLS_HD bool jac_on_curve(const Jac<F>& a) {
  if (is_zero(a.z)) return true;
  const F z2 = fsqr(a.z);
  const F z6 = fmul(fsqr(z2), z2);
  return eq(fsqr(a.y), add(fmul(fsqr(a.x), a.x), mul_curve_b(z6)));
}
// [r] a == infinity for a point ON the curve: membership in the order-r subgroup (what arkworks' `G1Affine` / `G2Affine`
// deserialisation checks, `is_in_correct_subgroup_assuming_on_curve` [third party]); plain MSB-first double-and-add with the
// group law above, r is a constant so control flow is uniform.  The point at infinity is a member.
template <class F>
BLS_HD_NOINLINE bool jac_in_subgroup(const Jac<F>& a) {
  if (!jac_on_curve(a)) return false;
  Jac<F> acc = a;                                   // leading one bit of r (bit 254)
  for (int b = 253; b >= 0; --b) {
    acc = jac_double_call(acc);
    if ((BLS_R_ORDER[b >> 6] >> (b & 63)) & 1ull) acc = jac_add(acc, a);
  }
  return is_zero(acc.z);
}
// the representative with Z = 1 (`Affine::from(Projective)` followed by `Projective::from`); infinity -> (1, 1, 0),
// arkworks' `Projective::zero()`
template <class F>
BLS_HD Jac<F> jac_normalize(const Jac<F>& a) {
  Jac<F> r;
  if (is_zero(a.z)) { set_one(r.x); set_one(r.y); set_zero(r.z); return r; }
  const F zi = inv(a.z);
  const F zi2 = fsqr(zi);
  r.x = fmul(a.x, zi2);
  r.y = fmul(a.y, fmul(zi2, zi));
  set_one(r.z);
  return r;
}
// one element of bls381_g1_op_batch / bls381_g2_op_batch; returns the flag byte
template <class F>
BLS_HD uint8_t point_op(int op, const Jac<F>& a, const Jac<F>& b, Jac<F>& r) {
  bool pred = false;
  switch (op) {
    case POINT_ADD: r = jac_add(a, b); break;
    case POINT_DOUBLE: r = jac_double_call(a); break;
    case POINT_NORMALIZE: r = jac_normalize(a); break;
    case POINT_EQ: r = a; pred = jac_eq(a, b); break;
    case POINT_ON_CURVE: r = a; pred = jac_on_curve(a); break;
    case POINT_IN_SUBGROUP: r = a; pred = jac_in_subgroup(a); break;
    default: r = a; break;
  }
  return (uint8_t)((is_zero(r.z) ? FLAG_INFINITY : 0) | (pred ? FLAG_TRUE : 0));
}

// ------------------------------------------------------------------------------------------ Mode A
struct LineND { Fp2R n, d; };

// shared tail of branches A and T (src/miller_loop_native_optimized.rs:20-23,31 / :44-47,55)
template <int Ln, int Vn, int Ld, int Vd>
BLS_HD LineND line_tail(const Fp2<Ln, Vn>& num, const Fp2<Ld, Vd>& den, const G2Jac& q1, const G1Jac& p) {
  const auto a = sub(mul_fp(q1.z, p.x), mul_fp(q1.x, p.z));      // xp z1 - x1 zp
  const auto b = sub(mul_fp(q1.z, p.y), mul_fp(q1.y, p.z));      // yp z1 - y1 zp
  LineND r;
  r.n = fit<1, 2>(sub(mul(num, a), mul(den, b)));
  r.d = mul(mul_fp(fit<1, 2>(den), p.z), q1.z);
  return r;
}

// tangent branch, taken unconditionally for the (R, R) call: num = y z - y z and den = x z - x z are exactly
// zero, so the reference always lands in `else if lambda_numerator == zero` (:38-61)
BLS_FN_COLD LineND line_tangent(const G2Jac& q1, const G1Jac& p) {
  const Fp2R xx = sqr(q1.x);
  const auto num = add(dbl(xx), xx);                               // 3 x1^2        (:39)
  const auto den = dbl(mul(q1.y, q1.z));                           // 2 y1 z1       (:40)
  return line_tail(num, den, q1, p);
}

// general call (R, Q): three-way branch of :14-77
BLS_FN_COLD LineND line_chord(const G2Jac& q1, const G2Jac& q2, const G1Jac& p) {
  const auto num = fit<1, 2>(sub(mul(q2.y, q1.z), mul(q1.y, q2.z)));   // :14
  const auto den = fit<1, 2>(sub(mul(q2.x, q1.z), mul(q1.x, q2.z)));   // :15
  if (!is_zero(den)) return line_tail(num, den, q1, p);                 // :16-37
  if (is_zero(num)) return line_tangent(q1, p);                        // :38-61
  LineND r;                                                            // :62-77 vertical
  r.n = fit<1, 2>(sub(mul_fp(q1.z, p.x), mul_fp(q1.x, p.z)));
  r.d = mul_fp(q1.z, p.z);
  return r;
}

// the reference squares its accumulators with the generic product, `f_num * f_num` (:91-92).  The value of a * a is the value of
// the Fq12 squaring formula (fq12_target_tree.rs:143-155: two Fq6 products instead of three), so that is what runs; the generic
// product stays selectable for A/B (BLS_REF_OPT_SQR_GENERIC=1, tools/kbench/kb_refopt.hip).
#ifndef BLS_REF_OPT_SQR_GENERIC
#define BLS_REF_OPT_SQR_GENERIC 0
#endif
BLS_HD Fp12R sqr_via_mul(const Fp12R& a) {
#if BLS_REF_OPT_SQR_GENERIC
  return mul_call(a, a);
#else
  return sqr_call(a);
#endif
}
// a * Fq12::new(Fq6::new(c, 0, 0), Fq6::zero()) -- the product of the reference's generic `*` with an embedded Fp2 value
// (:91-97), written without the partial products that meet the embedding's structural zeros: every Fp2 coefficient times c.
// Same field element as mul_call(a, embed_fp2(c)), hence the same bits; the accumulators stay full Fq12 values.
#ifndef BLS_REF_OPT_EMBED_GENERIC        // 1: multiply by the embedded value with the generic Fq12 product (A/B, tools/kbench/kb_refopt.hip)
#define BLS_REF_OPT_EMBED_GENERIC 0
#endif
BLS_HD Fp12R mul_by_embedded_fp2(const Fp12R& a, const Fp2R& c) {
#if BLS_REF_OPT_EMBED_GENERIC
  Fp12R e;
  e.c0 = Fp6R(c, Fp2R(fp2_zero()), Fp2R(fp2_zero()));
  e.c1 = Fp6R(fp6_zero());
  return mul_call(a, e);
#else
  return Fp12R(Fp6R(mul(a.c0.c0, c), mul(a.c0.c1, c), mul(a.c0.c2, c)), Fp6R(mul(a.c1.c0, c), mul(a.c1.c1, c), mul(a.c1.c2, c)));
#endif
}
BLS_HD Fp12R embed_fp2(const Fp2R& v) {        // Fq12::new(Fq6::new(v, 0, 0), Fq6::zero())  (:18-36)
  Fp12R r;
  r.c0 = Fp6R(v, Fp2R(fp2_zero()), Fp2R(fp2_zero()));
  r.c1 = Fp6R(fp6_zero());
  return r;
}

// optimized_miller_loop (:81-127).  GENERIC = true multiplies full Fq12 values like the reference does;
// false keeps the (provably Fp2-valued) accumulators in Fp2 -- identical output bits.
template <bool GENERIC>
BLS_HD Fp12R miller_ref_opt(const G1Jac& p, const G2Jac& q, uint8_t& flags) {
  G2Jac r = q;
  flags = 0;
  if constexpr (GENERIC) {
    Fp12R f_num = fp12_one(), f_den = fp12_one();
    for (int i = 0; i < 64; ++i) {                                        // :89, bit i of BLS_X, LSB first
      const LineND t = line_tangent(r, p);                                // :90
      f_num = mul_by_embedded_fp2(sqr_via_mul(f_num), t.n);               // :91
      f_den = mul_by_embedded_fp2(sqr_via_mul(f_den), t.d);               // :92
      r = g2_jac_double(r);                                                  // :93
      if ((BLS_X_PARAM >> i) & 1ull) {
        const LineND c = line_chord(r, q, p);                             // :95
        f_num = mul_by_embedded_fp2(f_num, c.n);                          // :96
        f_den = mul_by_embedded_fp2(f_den, c.d);                          // :97
        r = g2_jac_add(r, q);                                             // :98
      }
    }
    if (is_zero(f_den)) { flags |= FLAG_DEN_ZERO; }                       // :102 panics in the reference
    Fp12R f = mul_call(f_num, inv(f_den));                                // :102
    return mul_call(f, f);                                                // :116-121
  } else {
    Fp2R f_num = fp2_one(), f_den = fp2_one();
    for (int i = 0; i < 64; ++i) {
      const LineND t = line_tangent(r, p);
      f_num = mul(sqr(f_num), t.n);
      f_den = mul(sqr(f_den), t.d);
      r = g2_jac_double(r);   
      if ((BLS_X_PARAM >> i) & 1ull) {
        const LineND c = line_chord(r, q, p);
        f_num = mul(f_num, c.n);
        f_den = mul(f_den, c.d);
        r = g2_jac_add(r, q);
      }
    }
    if (is_zero(f_den)) { flags |= FLAG_DEN_ZERO; }
    const Fp2R f = mul(f_num, inv(f_den));
    return embed_fp2(sqr(f));
  }
}

// ------------------------------------------------------------------------------------------ Mode B
struct LineCoeffs { Fp2R a, b, c; };     // (tmp0, tmp3, tmp6) / (t10, t1, t9)

// src/miller_loop_native.rs:27-55 (Alg. 26 of ePrint 2010/354)
BLS_HD LineCoeffs doubling_step(G2Jac& r) {
  const Fp2R tmp0 = sqr(r.x);
  const Fp2R tmp1 = sqr(r.y);
  const Fp2R tmp2 = sqr(tmp1);
  const auto tmp3 = dbl(sub(sub(sqr(add(tmp1, r.x)), tmp0), tmp2));
  const auto tmp4 = add(dbl(tmp0), tmp0);
  const auto tmp6 = add(r.x, tmp4);
  const Fp2R tmp5 = sqr(tmp4);
  const Fp2R zsq = sqr(r.z);
  const auto X3 = fit<1, 2>(sub(sub(tmp5, tmp3), tmp3));
  const auto Z3 = fit<1, 2>(sub(sub(sqr(add(r.z, r.y)), tmp1), zsq));
  const auto Y3 = fit<1, 2>(sub(mul(sub(tmp3, X3), tmp4), dbl(dbl(dbl(tmp2)))));
  LineCoeffs l;
  l.b = fit<1, 2>(neg(dbl(mul(tmp4, zsq))));                               // :44-46
  l.c = fit<1, 2>(sub(sub(sub(sqr(tmp6), tmp0), tmp5), dbl(dbl(tmp1))));   // :47-50
  l.a = fit<1, 2>(dbl(mul(Z3, zsq)));                                      // :51-52
  r.x = X3; r.y = Y3; r.z = Z3;
  return l;
}

// src/miller_loop_native.rs:58-87 (Alg. 27)
BLS_HD LineCoeffs addition_step(G2Jac& r, const G2Aff& q) {
  const Fp2R zsq = sqr(r.z);
  const Fp2R ysq = sqr(q.y);
  const Fp2R t0 = mul(zsq, q.x);
  const Fp2R t1 = mul(sub(sub(sqr(add(q.y, r.z)), ysq), zsq), zsq);
  const auto t2 = fit<1, 2>(sub(t0, r.x));
  const Fp2R t3 = sqr(t2);
  const auto t4 = dbl(dbl(t3));
  const Fp2R t5 = mul(t4, t2);
  const auto t6 = fit<1, 2>(sub(sub(t1, r.y), r.y));
  const Fp2R t9 = mul(t6, q.x);
  const Fp2R t7 = mul(t4, r.x);
  const auto X3 = fit<1, 2>(sub(sub(sub(sqr(t6), t5), t7), t7));
  const auto Z3 = fit<1, 2>(sub(sub(sqr(add(r.z, t2)), zsq), t3));
  const auto t10 = add(q.y, Z3);
  const Fp2R t8 = mul(sub(t7, X3), t6);
  const auto Y3 = fit<1, 2>(sub(t8, dbl(mul(r.y, t5))));
  const auto t10b = sub(sub(sqr(t10), ysq), sqr(Z3));
  LineCoeffs l;
  l.c = fit<1, 2>(sub(dbl(t9), t10b));          // t9
  l.a = fit<1, 2>(dbl(Z3));                     // t10
  l.b = fit<1, 2>(dbl(neg(t6)));                // t1   (:83-84)
  r.x = X3; r.y = Y3; r.z = Z3;
  return l;
}

// ---- the same two steps in homogeneous projective coordinates (x = X/Z, y = Y/Z), for callers that only need the value
// AFTER the final exponentiation: 3 M + 6 S for the doubling instead of Alg. 26's 3 M + 8 S.  The line is the same line
// times an Fp2 factor (-Z for the tangent), which the easy part of the final exponentiation kills, so pairings and pairing
// products come out bit-identical while Miller-only results (bls381_multi_miller_batch) keep Alg. 26/27.
// Formulas: Costello-Lange-Naehrig / Aranha et al. (ePrint 2010/526) for the M-type twist y^2 = x^3 + 4 xi, the whole
// point scaled by 4 so that no halving is needed.  Coefficient slots as `ell` reads them: a <-> yP, b <-> xP, c <-> 1.
BLS_HD LineCoeffs doubling_step_hom(G2Jac& r) {
  const Fp2R A = mul(r.x, r.y);                                   // 2a
  const Fp2R B = sqr(r.y);
  const Fp2R C = sqr(r.z);
  const auto c3 = add(dbl(C), C);
  const auto E = fit<1, 2>(dbl(dbl(mul_xi(fit<1, 2>(c3)))));       // 3 b' C, b' = 4 xi
  const auto F = add(dbl(E), E);
  const auto G = add(B, F);                                       // 2g
  const auto H = fit<1, 2>(sub(sub(sqr(add(r.y, r.z)), B), C));   // 2 Y Z
  const auto I = fit<1, 2>(sub(E, B));
  const Fp2R J = sqr(r.x);
  LineCoeffs l;
  l.c = I;
  l.b = fit<1, 2>(add(dbl(J), J));
  l.a = fit<1, 2>(neg(H));
  const auto X3 = fit<1, 2>(dbl(mul(A, sub(B, F))));
  const auto m12e = fit<1, 64>(neg(dbl(dbl(add(dbl(E), E)))));   // -12 E
  const Fp2R Y3 = dot2(sq2(fit<1, 64>(G)), pr2(m12e, E));        // G^2 - 12 E^2 with one reduction
  const auto Z3 = fit<1, 2>(dbl(dbl(mul(B, H))));
  r.x = X3; r.y = Y3; r.z = Z3;
  return l;
}
BLS_HD LineCoeffs addition_step_hom(G2Jac& r_io, const G2Aff& q) {
  const G2Jac r = r_io;
  const auto theta = fit<1, 2>(sub(r.y, mul(q.y, r.z)));
  const auto lambda = fit<1, 2>(sub(r.x, mul(q.x, r.z)));
  const Fp2R c = sqr(theta);
  const Fp2R d = sqr(lambda);
  const Fp2R e = mul(lambda, d);
  const Fp2R f = mul(r.z, c);
  const Fp2R g = mul(r.x, d);
  const auto h = fit<1, 2>(sub(add(e, f), dbl(g)));
  const Fp2R j = dot2(pr2(theta, q.x), pr2(fit<1, 64>(neg(lambda)), q.y));          // theta xq - lambda yq, one reduction
  LineCoeffs l;
  l.c = j;
  l.b = fit<1, 2>(neg(theta));
  l.a = lambda;
  G2Jac o;
  o.x = mul(lambda, h);
  o.y = dot2(pr2(theta, fit<1, 64>(sub(g, h))), pr2(fit<1, 64>(neg(e)), r.y));     // theta (g - h) - e Y, one reduction
  o.z = mul(r.z, e);
  r_io = o;
  return l;
}

// commented `ell`, src/miller_loop_native.rs:139-152
BLS_HD Fp12R ell(const Fp12R& f, const LineCoeffs& l, const G1Aff& p) {
  const Fp2R c0 = mul_fp(l.a, p.y);
  const Fp2R c1 = mul_fp(l.b, p.x);
  return mul_by_014(f, l.c, c1, c0);
}
// ell(1, l, p): the line itself as a (sparse) Fq12 value -- what the first step of a Miller loop multiplies into f = 1
BLS_HD Fp12R line_value(const LineCoeffs& l, const G1Aff& p) {
  const Fp2R z = Fp2R(fp2_zero());
  return Fp12R(Fp6R(l.c, mul_fp(l.b, p.x), z), Fp6R(z, mul_fp(l.a, p.y), z));
}
BLS_HD_NOINLINE LineCoeffs doubling_step_call(G2Jac& r) { return doubling_step(r); }
BLS_HD_NOINLINE LineCoeffs addition_step_call(G2Jac& r, const G2Aff& q) { return addition_step(r, q); }
BLS_HD_NOINLINE LineCoeffs doubling_step_hom_call(G2Jac& r) { return doubling_step_hom(r); }
BLS_HD_NOINLINE LineCoeffs addition_step_hom_call(G2Jac& r, const G2Aff& q) { return addition_step_hom(r, q); }
BLS_HD_NOINLINE Fp12R ell_call(const Fp12R& f, const LineCoeffs& l, const G1Aff& p) { return ell(f, l, p); }

// Miller loop over N_PAIRS pairs with shared squarings: src/miller_loop_native.rs:89-116 driven the way
// multi_miller_loop (:154-212) intends.  skip[j] marks a pair with a point at infinity (its lines are
// skipped, like the commented `either_identity` select at :167-170).
//
// Schedule (:94-109): for the 62 bits of BLS_X >> 1 below its leading one, MSB first: doubling step, addition
// step if the bit is set, then one squaring; after the loop one more doubling step; conjugate (x < 0).
// The code below folds the trailing doubling step into the loop (iteration b = -1, no squaring) and walks the
// (doubling, addition) steps of an iteration with a uniform `pass` loop, so the instruction stream holds ONE copy of
// each step and of the sparse multiplication -- the loop body is the whole hot path and stays I-cache friendly.
// HOM = true walks the same schedule with the homogeneous-projective steps above (full pairings only).
// The squaring of an iteration.  A loop whose value goes through the final exponentiation (HOM) may carry factors from Fp, so it
// COULD square with 2 f^2 by SQR3 over Fp4 (tower.hip.h sqr_twice: 6 468 instead of 7 056 multiply-adds per lane) -- measured in round 3
// (BLS_HOM_SQR3=1): 588 fewer multiply-adds, 852 more other instructions, 32.4 vs 31.5 ms per 2^18 loops.  Off; Miller-only results
// need the exact square in any case.
#ifndef BLS_HOM_SQR3
#define BLS_HOM_SQR3 0
#endif
template <bool HOM>
BLS_HD Fp12R miller_sqr(const Fp12R& f) {
  if constexpr (HOM && BLS_HOM_SQR3) return sqr_twice(f);
  else return sqr(f);
}

template <int N_PAIRS, bool HOM = false>
BLS_HD Fp12R miller_ate(const G1Aff* p, const G2Aff* q, const bool* skip) {
  G2Jac r[N_PAIRS];
#pragma unroll
  for (int j = 0; j < N_PAIRS; ++j) { r[j].x = q[j].x; r[j].y = q[j].y; r[j].z = Fp2R(fp2_one()); }
  Fp12R f = fp12_one();
  for (int b = 61; b >= -1; --b) {
    const bool last = b < 0;
    const int passes = (!last && (((BLS_X_PARAM >> 1) >> (b < 0 ? 0 : b)) & 1ull)) ? 2 : 1;
    BLS_MILLER_ITERATION_HOOK();
    for (int pass = 0; pass < passes; ++pass) {
      if constexpr (N_PAIRS <= BLS_INLINE_PAIRS_MAX) {
#pragma unroll
        for (int j = 0; j < N_PAIRS; ++j) {
          const LineCoeffs l = HOM ? ((pass == 0) ? doubling_step_hom(r[j]) : addition_step_hom(r[j], q[j]))
                                   : ((pass == 0) ? doubling_step(r[j]) : addition_step(r[j], q[j]));
          // the very first line meets f = 1: no multiplication (a uniform branch; a skipped first pair leaves f = 1 for the next)
          if (!skip[j]) f = (BLS_MILLER_PEEL && b == 61 && pass == 0 && j == 0) ? line_value(l, p[j]) : ell(f, l, p[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < N_PAIRS; ++j) {
          const LineCoeffs l = HOM ? ((pass == 0) ? doubling_step_hom_call(r[j]) : addition_step_hom_call(r[j], q[j]))
                                   : ((pass == 0) ? doubling_step_call(r[j]) : addition_step_call(r[j], q[j]));
          if (!skip[j]) f = ell_call(f, l, p[j]);
        }
      }
    }
    if (!last) f = (N_PAIRS <= BLS_INLINE_PAIRS_MAX) ? miller_sqr<HOM>(f) : sqr_call(f);   // :106
  }
  return fit<1, 2>(conj(f));                                              // :111-113 (x < 0)
}

// Pairing products with FIXED (prepared) G2 points: NVAR in {0, 1} pairs walk their own curve steps, NFIXED pairs take their line
// coefficients from `fixed_line(j, idx)` -- triple `idx` (0..67, loop order) of fixed point j, e.g. out of LDS (k_groth16.hip) or out of
// a plain array (tests/hostsim).  p[0..NVAR) belong to the variable pair, p[NVAR..) to the fixed ones; same schedule as miller_ate.
// HOM: homogeneous steps for the variable pair (its lines then differ from Alg. 26/27's by Fp2 factors the final exponentiation
// kills; the prepared lines are Alg. 26/27's either way).
template <int NVAR, int NFIXED, bool HOM, class FixedLine>
BLS_HD Fp12R miller_prepared(const G1Aff* p, const G2Aff& q0, const bool* skip, FixedLine fixed_line) {
  G2Jac r;
  if constexpr (NVAR == 1) { r.x = q0.x; r.y = q0.y; r.z = Fp2R(fp2_one()); }
  Fp12R f = fp12_one();
  int idx = 0;
  for (int b = 61; b >= -1; --b) {
    const bool last = b < 0;
    const int passes = (!last && (((BLS_X_PARAM >> 1) >> (b < 0 ? 0 : b)) & 1ull)) ? 2 : 1;
    for (int pass = 0; pass < passes; ++pass) {
      if constexpr (NVAR == 1) {
        const LineCoeffs l0 = HOM ? ((pass == 0) ? doubling_step_hom(r) : addition_step_hom(r, q0))
                                  : ((pass == 0) ? doubling_step(r) : addition_step(r, q0));
        if (!skip[0]) f = ell(f, l0, p[0]);
      }
#pragma unroll
      for (int j = 0; j < NFIXED; ++j) {       // no curve arithmetic for the fixed points
        const LineCoeffs lj = fixed_line(j, idx);
        if (!skip[NVAR + j]) f = ell(f, lj, p[NVAR + j]);
      }
      ++idx;
    }
    if (!last) f = sqr(f);
  }
  return fit<1, 2>(conj(f));
}

// G2Prepared: the 68 line-coefficient triples of one G2 point in loop order (63 doubling + 5 addition steps),
// the precomputation behind `G2PreparedTarget::from` (src/miller_loop_target.rs:23-76; `assert_eq!(coeffs.len(), 68)`
// at :69).  `emit(index, coeffs)` receives each triple (a, b, c) = (tmp0, tmp3, tmp6) / (t10, t1, t9).
template <class Emit>
BLS_HD void g2_prepare(const G2Aff& q, Emit emit) {
  G2Jac r; r.x = q.x; r.y = q.y; r.z = Fp2R(fp2_one());
  int idx = 0;
  for (int b = 61; b >= -1; --b) {
    const int passes = (b >= 0 && (((BLS_X_PARAM >> 1) >> (b < 0 ? 0 : b)) & 1ull)) ? 2 : 1;
    for (int pass = 0; pass < passes; ++pass) {
      const LineCoeffs l = (pass == 0) ? doubling_step(r) : addition_step(r, q);
      emit(idx++, l);
    }
  }
}

// ------------------------------------------------------------------------------------------- I/O
BLS_HD G1Jac g1_jac_load(const uint64_t* w, int form) {
  G1Jac p; p.x = fp_load(w, form); p.y = fp_load(w + 6, form); p.z = fp_load(w + 12, form); return p;
}
BLS_HD G2Jac g2_jac_load(const uint64_t* w, int form) {
  G2Jac q; q.x = fp2_load(w, form); q.y = fp2_load(w + 12, form); q.z = fp2_load(w + 24, form); return q;
}
BLS_HD void jac_load(G1Jac& p, const uint64_t* w, int form) { p = g1_jac_load(w, form); }
BLS_HD void jac_load(G2Jac& q, const uint64_t* w, int form) { q = g2_jac_load(w, form); }
BLS_HD void jac_store(uint64_t* w, const G1Jac& p, int form) { fp_store(w, p.x, form); fp_store(w + 6, p.y, form); fp_store(w + 12, p.z, form); }
BLS_HD void jac_store(uint64_t* w, const G2Jac& q, int form) { fp2_store(w, q.x, form); fp2_store(w + 12, q.y, form); fp2_store(w + 24, q.z, form); }
// affine points; the all-zero encoding (x = y = 0, not on either curve) denotes the point at infinity
BLS_HD G1Aff g1_aff_load(const uint64_t* w, int form, bool& inf) {
  uint64_t o = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) o |= w[i];
  inf = (o == 0);
  G1Aff p; p.x = fp_load(w, form); p.y = fp_load(w + 6, form); return p;
}
BLS_HD G2Aff g2_aff_load(const uint64_t* w, int form, bool& inf) {
  uint64_t o = 0;
#pragma unroll
  for (int i = 0; i < 24; ++i) o |= w[i];
  inf = (o == 0);
  G2Aff q; q.x = fp2_load(w, form); q.y = fp2_load(w + 12, form); return q;
}

// ---- products of K >= 2 pairs on the device: the same loop with the idle pairs' R parked in LDS.
// (Written against the Fp2 backend's slot helpers, so the host build runs the same source over a plain buffer: tests/hostsim.)
// What a lane must keep through an iteration of miller_ate is f (84 registers) and, per pair, R (42), Q (28) and P (28); with the
// working set of the Fq12 squaring (~170) one pair just fits 512 registers, two or three do not.  Left to itself the compiler keeps
// r[], q[], p[] in the private segment (1.0-1.4 KB per lane) and walks them with a run-time pair index -- one copy of the step code,
// but every pair-step starts by waiting for its R, P (and Q) to come back from scratch, alone on its SIMD: 91 ms per 2^18 three-pair
// products where the one-pair kernel's instruction rate would give 71.  Here:
//   * the pair index is a run-time loop by design (one copy of the step code);
//   * R of the current pair is in registers and CHANGES PLACES with the next pair's in LDS after every step (the K - 1 slots hold
//     the other pairs in FIFO order, so the slot to use is the step count mod K - 1): six 128-bit LDS instructions each way;
//   * P and Q of ALL pairs stay in registers (AGPRs in practice) and the current pair's are picked by comparing the loop index --
//     wave-uniform, 28 + 28 register moves.  Measured alternatives (tools/kbench/kb_multi.hip, 2^18 products, K = 3): P from LDS
//     102 ms (an LDS load lands in VGPRs as whole 128-bit quads exactly where the line multiplication is at its register peak, and
//     gets spilled at once), Q re-read from the input array at the five addition steps 73.9 ms, both in registers 72.3 ms.
// A slot is 16 dwords per lane (14 used), lane-sliced in four 16-byte quarters: every LDS instruction of a wave moves 64 consecutive
// 16-byte pieces (measured, K = 3: LDS busy 0.6 % of the wave cycles, 6.5 % of that in bank conflicts; profiles/r03_soak); addresses are one per-lane base register plus immediate offsets (LANES is a compile-time
// constant).  K = 3: 2 x 3 x 64 B = 384 B per lane, 24 KB per one-wave workgroup, four workgroups per CU.
struct alignas(16) LdsQuad { uint32_t w[4]; };
template <int LANES>
struct LaneLds {
  LdsQuad* mine;          // this lane's column:  base + lane
  BLS_HD void put(int slot, const FpR& a) const {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      LdsQuad v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v.w[j] = (4 * q + j < NL) ? a.l[4 * q + j] : 0u;
      mine[(4 * slot + q) * LANES] = v;
    }
  }
  BLS_HD FpR get(int slot) const {
    FpR r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const LdsQuad v = mine[(4 * slot + q) * LANES];
#pragma unroll
      for (int j = 0; j < 4; ++j) if (4 * q + j < NL) r.l[4 * q + j] = v.w[j];
    }
    return r;
  }
};
template <int LANES>
BLS_HD LaneLds<LANES> lane_lds(LdsQuad* base, unsigned lane) { return LaneLds<LANES>{base + lane}; }
// Scheduling fence between the phases of a pair-step: without it the instruction scheduler pulls the next phase's LDS loads and
// register picks up into the current phase and the allocator answers with a private segment of 1.8 KB instead of 0.5 KB (K = 3).
BLS_HD void phase_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
#endif
}
constexpr int miller_lds_slots(int k) { return 3 * (k - 1) * SCRATCH_FP_PER_FP2; }     // per lane, 64 bytes each (one Fp per slot)
constexpr size_t miller_lds_bytes_per_lane(int k) { return (size_t)miller_lds_slots(k) * 64; }

// g1e / g2e: the element's first pair in the caller's arrays (12 / 24 u64 per pair, pairs consecutive); returns the Miller value and
// ORs the flag bits into fl
template <int K, bool HOM, int LANES>
BLS_HD Fp12R miller_ate_lds(const uint64_t* g1e, const uint64_t* g2e, int form, const LaneLds<LANES>& lds, uint8_t& fl) {
  static_assert(K >= 2, "one pair: miller_ate");
  unsigned skip = 0;
  G1Aff p_all[K]; G2Aff q_all[K];          // only ever indexed by constants (unrolled comparisons): registers, not memory
  G2Jac cur;
  for (int j = 0; j < K; ++j) {            // (the compiler keeps this a loop: the conversions are one copy of code)
    bool i1, i2;
    const G1Aff p = g1_aff_load(g1e + 12 * j, form, i1);
    const G2Aff q = g2_aff_load(g2e + 24 * j, form, i2);
    if (i1 || i2) { skip |= 1u << j; fl |= FLAG_INFINITY; }
#pragma unroll
    for (int t = 0; t < K; ++t) if (j == t) { p_all[t] = p; q_all[t] = q; }      // static indices only
    const Fp2R one = Fp2R(fp2_one());
    if (j == 0) { cur.x = q.x; cur.y = q.y; cur.z = one; }
    else { fp2_slot_put(lds, 3 * (j - 1) + 0, q.x); fp2_slot_put(lds, 3 * (j - 1) + 1, q.y); fp2_slot_put(lds, 3 * (j - 1) + 2, one); }
  }
  Fp12R f = fp12_one();
  int slot = 0;                                    // the slot that holds the pair after the current one
  for (int b = 61; b >= -1; --b) {
    const bool last = b < 0;
    const int passes = (!last && (((BLS_X_PARAM >> 1) >> (b < 0 ? 0 : b)) & 1ull)) ? 2 : 1;
    BLS_MILLER_ITERATION_HOOK();
    for (int pass = 0; pass < passes; ++pass) {
#pragma clang loop unroll(disable)
      for (int j = 0; j < K; ++j) {
        LineCoeffs l;
        if (pass == 0) {
          l = HOM ? doubling_step_hom(cur) : doubling_step(cur);
        } else {
          G2Aff q = q_all[0];
#pragma unroll
          for (int t = 1; t < K; ++t) if (j == t) q = q_all[t];
          l = HOM ? addition_step_hom(cur, q) : addition_step(cur, q);
        }
        phase_fence();
        if (!((skip >> j) & 1u)) {
          G1Aff p = p_all[0];
#pragma unroll
          for (int t = 1; t < K; ++t) if (j == t) p = p_all[t];
          // the very first line meets f = 1: no multiplication (a uniform branch; a skipped first pair leaves f = 1 for the next)
          f = (BLS_MILLER_PEEL && b == 61 && pass == 0 && j == 0) ? line_value(l, p) : ell(f, l, p);
        }
        phase_fence();
        {                                           // the current R changes places with the next pair's
          G2Jac nx;
          nx.x = fp2_slot_get(lds, 3 * slot + 0); nx.y = fp2_slot_get(lds, 3 * slot + 1); nx.z = fp2_slot_get(lds, 3 * slot + 2);
          fp2_slot_put(lds, 3 * slot + 0, cur.x); fp2_slot_put(lds, 3 * slot + 1, cur.y); fp2_slot_put(lds, 3 * slot + 2, cur.z);
          cur = nx;
          slot = (slot + 1 == K - 1) ? 0 : slot + 1;
        }
        phase_fence();
      }
    }
    if (!last) f = miller_sqr<HOM>(f);
  }
  return fit<1, 2>(conj(f));
}

// ---- Mode A with generic Fq12 accumulators, one accumulator in registers and the other in LDS.
// optimized_miller_loop (src/miller_loop_native_optimized.rs:81-127) carries TWO Fq12 accumulators, f_num and f_den (168 registers
// per lane), next to R (42), Q (42) and P (42), and multiplies them as full Fq12 values four times per iteration; with an inlined
// Fq12 product (~250 registers of operands and partial results) that does not fit 512 registers and k_miller_ref_opt_wide spilled
// 1.4 KB per lane.  The two accumulators are never needed at the same time: the one that is not being multiplied waits in six LDS
// slots and the two change places after every half-iteration -- which also makes the two halves ONE instruction stream (a run-time
// loop over {numerator, denominator}) instead of two copies.  Q is re-read from the input array at the six addition steps.
template <int LANES>
BLS_HD void fp12_lds_put(const LaneLds<LANES>& lds, const Fp12R& a) {
  fp2_slot_put(lds, 0, a.c0.c0); fp2_slot_put(lds, 1, a.c0.c1); fp2_slot_put(lds, 2, a.c0.c2);
  fp2_slot_put(lds, 3, a.c1.c0); fp2_slot_put(lds, 4, a.c1.c1); fp2_slot_put(lds, 5, a.c1.c2);
}
template <int LANES>
BLS_HD Fp12R fp12_lds_get(const LaneLds<LANES>& lds) {
  return Fp12R(Fp6R(fp2_slot_get(lds, 0), fp2_slot_get(lds, 1), fp2_slot_get(lds, 2)),
               Fp6R(fp2_slot_get(lds, 3), fp2_slot_get(lds, 4), fp2_slot_get(lds, 5)));
}
template <int LANES>
BLS_HD void fp12_lds_exchange(Fp12R& cur, const LaneLds<LANES>& lds) {
  const Fp12R nx = fp12_lds_get(lds);
  fp12_lds_put(lds, cur);
  cur = nx;
}
constexpr size_t ref_opt_lds_bytes_per_lane() { return 6 * SCRATCH_FP_PER_FP2 * 64; }

template <int LANES>
BLS_HD Fp12R miller_ref_opt_generic_lds(const G1Jac& p, const uint64_t* g2w, int form, const LaneLds<LANES>& lds, uint8_t& flags) {
  G2Jac r = g2_jac_load(g2w, form);
  flags = 0;
  Fp12R cur = fp12_one();                                                  // f_num; f_den = 1 waits in LDS
  fp12_lds_put(lds, cur);
  for (int i = 0; i < 64; ++i) {                                           // :89, bit i of BLS_X, LSB first
    const LineND t = line_tangent(r, p);                                   // :90
#pragma clang loop unroll(disable)
    for (int h = 0; h < 2; ++h) {                                          // h = 0: f_num (:91), h = 1: f_den (:92)
      const Fp2R c = select(h == 0, t.n, t.d);
      cur = mul_by_embedded_fp2(sqr_via_mul(cur), c);
      fp12_lds_exchange(cur, lds);
    }
    r = g2_jac_double(r);                                                     // :93
    if ((BLS_X_PARAM >> i) & 1ull) {
      const G2Jac q = g2_jac_load(g2w, form);
      const LineND cl = line_chord(r, q, p);                               // :95
#pragma clang loop unroll(disable)
      for (int h = 0; h < 2; ++h) {                                        // :96, :97
        const Fp2R c = select(h == 0, cl.n, cl.d);
        cur = mul_by_embedded_fp2(cur, c);
        fp12_lds_exchange(cur, lds);
      }
      r = g2_jac_add(r, q);                                                // :98
    }
  }
  const Fp12R f_num = cur;
  const Fp12R f_den = fp12_lds_get(lds);
  if (is_zero(f_den)) { flags |= FLAG_DEN_ZERO; }                          // :102 panics in the reference
  const Fp12R f = mul_call(f_num, inv(f_den));                             // :102
  return mul_call(f, f);                                                   // :116-121
}

}  // namespace bls
