// Batched witness-generator backend: the native arithmetic plonky2 invokes while it
// fills a witness -- the `SimpleGenerator::run_once` hooks of the reference's circuit gadgets -- as device
// routines.  Values cross the boundary as CANONICAL integers in 12 x u32 little-endian digits (== 6 x u64), the
// BigUint limb format of FqTarget (src/fields/fq_target.rs:289-313), with the flat coefficient orders of
// MyFq12 / MyFq6 (src/fields/helpers.rs:22-76, src/fields/my_fq6.rs:12-44).
//
//   Fq12InverseGenerator::run_once   src/fields/fq12_target.rs:351-374   (inverse().unwrap(): zero panics)
//   Fq6InverseGenerator::run_once    src/fields/fq6_target.rs:395-418    (same)
//   Fq2InverseGenerator::run_once    src/fields/fq2_target.rs:331-352    (None -> zero)
//   FqSqrtGenerator::run_once        src/fields/fq_target.rs:330-343     (sqrt().unwrap(), then sgn0 fix-up)
//   Fq2SqrtGenerator::run_once       src/fields/fq2_target.rs:385-410    (same with sgn0_fq2)
//
// The square roots are defined by their value, not by arkworks' algorithm: of the two roots +-s the generators
// keep the one whose sgn0 equals the requested bit (src/fields/helpers.rs:159-174), so any correct root finder
// gives identical bits.
#pragma once
#include "tower.hip.h"

namespace bls {

enum : uint8_t { WFLAG_NO_INVERSE = 4, WFLAG_NOT_SQUARE = 8, WFLAG_SIGN_UNSATISFIABLE = 16 };

// flat index of tree coefficient (half h, Fp2 slot q, component c): MyFq12 coeffs[2q + h + 6c]  (helpers.rs:39-41)
BLS_HD int myfq12_index(int h, int q, int c) { return 2 * q + h + 6 * c; }
// MyFq6 coeffs[q + 3c]  (my_fq6.rs:25)
BLS_HD int myfq6_index(int q, int c) { return q + 3 * c; }

BLS_HD Fp6R fp6_load_my(const uint64_t* w) {
  return Fp6R(fp2_load2(w + 6 * myfq6_index(0, 0), w + 6 * myfq6_index(0, 1), IO_CANONICAL),
              fp2_load2(w + 6 * myfq6_index(1, 0), w + 6 * myfq6_index(1, 1), IO_CANONICAL),
              fp2_load2(w + 6 * myfq6_index(2, 0), w + 6 * myfq6_index(2, 1), IO_CANONICAL));
}
BLS_HD void fp6_store_my(uint64_t* w, const Fp6R& a) {
  fp2_store2(w + 6 * myfq6_index(0, 0), w + 6 * myfq6_index(0, 1), a.c0, IO_CANONICAL);
  fp2_store2(w + 6 * myfq6_index(1, 0), w + 6 * myfq6_index(1, 1), a.c1, IO_CANONICAL);
  fp2_store2(w + 6 * myfq6_index(2, 0), w + 6 * myfq6_index(2, 1), a.c2, IO_CANONICAL);
}
BLS_HD Fp12R fp12_load_my(const uint64_t* w) {
  Fp12R r;
  r.c0 = Fp6R(fp2_load2(w + 6 * myfq12_index(0, 0, 0), w + 6 * myfq12_index(0, 0, 1), IO_CANONICAL),
              fp2_load2(w + 6 * myfq12_index(0, 1, 0), w + 6 * myfq12_index(0, 1, 1), IO_CANONICAL),
              fp2_load2(w + 6 * myfq12_index(0, 2, 0), w + 6 * myfq12_index(0, 2, 1), IO_CANONICAL));
  r.c1 = Fp6R(fp2_load2(w + 6 * myfq12_index(1, 0, 0), w + 6 * myfq12_index(1, 0, 1), IO_CANONICAL),
              fp2_load2(w + 6 * myfq12_index(1, 1, 0), w + 6 * myfq12_index(1, 1, 1), IO_CANONICAL),
              fp2_load2(w + 6 * myfq12_index(1, 2, 0), w + 6 * myfq12_index(1, 2, 1), IO_CANONICAL));
  return r;
}
BLS_HD void fp12_store_my(uint64_t* w, const Fp12R& a) {
  fp2_store2(w + 6 * myfq12_index(0, 0, 0), w + 6 * myfq12_index(0, 0, 1), a.c0.c0, IO_CANONICAL);
  fp2_store2(w + 6 * myfq12_index(0, 1, 0), w + 6 * myfq12_index(0, 1, 1), a.c0.c1, IO_CANONICAL);
  fp2_store2(w + 6 * myfq12_index(0, 2, 0), w + 6 * myfq12_index(0, 2, 1), a.c0.c2, IO_CANONICAL);
  fp2_store2(w + 6 * myfq12_index(1, 0, 0), w + 6 * myfq12_index(1, 0, 1), a.c1.c0, IO_CANONICAL);
  fp2_store2(w + 6 * myfq12_index(1, 1, 0), w + 6 * myfq12_index(1, 1, 1), a.c1.c1, IO_CANONICAL);
  fp2_store2(w + 6 * myfq12_index(1, 2, 0), w + 6 * myfq12_index(1, 2, 1), a.c1.c2, IO_CANONICAL);
}

// FqSqrtGenerator (fq_target.rs:330-343): the root of x whose sgn0 equals `sgn`.
BLS_HD FpR witness_fq_sqrt(const FpR& x, bool sgn, uint8_t& flags) {
  const FpR s = fp_sqrt_candidate(x);
  if (!fp_eq(sqr(s), x)) { flags |= WFLAG_NOT_SQUARE; return FpR(fp_zero()); }       // sqrt().unwrap() panics
  const bool s0 = fp_sgn0(s);
  const FpR r = select(s0 != sgn, FpR(fit<1, 2>(neg(s))), s);
  if (is_zero(r) && sgn) { flags |= WFLAG_SIGN_UNSATISFIABLE; }                      // assert_eq!(sgn0(0), true) fails
  return r;
}

// Fq2SqrtGenerator (fq2_target.rs:385-410).  p = 3 mod 4, so -1 is a non-residue and for x = a + b u:
//   b == 0: sqrt(a) if a is a square, else sqrt(-a) u;
//   b != 0: n = a^2 + b^2 must be a square with root sn; c0^2 = (a +- sn)/2 (one of the two is a square),
//           c1 = b / (2 c0).
// All Fp work is replicated on both lanes of a pair (coeff0/coeff1), so the branches are pair-uniform.
BLS_HD Fp2R witness_fq2_sqrt(const Fp2R& x, bool sgn, uint8_t& flags) {
  const FpR a = coeff0(x), b = coeff1(x);
  FpR c0 = FpR(fp_zero()), c1 = FpR(fp_zero());
  bool ok = true;
  if (is_zero(b)) {
    const FpR s = fp_sqrt_candidate(a);
    if (fp_eq(sqr(s), a)) { c0 = s; }
    else {
      const FpR na = fit<1, 2>(neg(a));
      const FpR t = fp_sqrt_candidate(na);
      ok = fp_eq(sqr(t), na);
      c1 = t;
    }
  } else {
    const FpR n = mul2(a, a, b, b);
    const FpR sn = fp_sqrt_candidate(n);
    if (!fp_eq(sqr(sn), n)) ok = false;
    else {
      const Fp<1, 1> half = fp_const(BLS_INV2_INT);
      FpR d = mul(add(a, sn), half);
      FpR r0 = fp_sqrt_candidate(d);
      if (!fp_eq(sqr(r0), d)) {
        d = mul(sub(a, sn), half);
        r0 = fp_sqrt_candidate(d);
        ok = fp_eq(sqr(r0), d);
      }
      c0 = r0;
      c1 = mul(b, fp_inv(dbl(r0)));
    }
  }
  if (!ok) { flags |= WFLAG_NOT_SQUARE; return Fp2R(fp2_zero()); }
  // sgn0_fq2 (helpers.rs:169-174): parity of c0, or of c1 when c0 == 0
  const bool z0 = is_zero(c0);
  const bool s0 = fp_sgn0(c0) || (z0 && fp_sgn0(c1));
  const bool flip = s0 != sgn;
  const FpR o0 = select(flip, FpR(fit<1, 2>(neg(c0))), c0);
  const FpR o1 = select(flip, FpR(fit<1, 2>(neg(c1))), c1);
  if (z0 && is_zero(c1) && sgn) flags |= WFLAG_SIGN_UNSATISFIABLE;
  return Fp2R(fp2_from_coeffs(o0, o1));
}

// Point decompression: the affine point with abscissa x whose ordinate has the requested sign, built from the
// reference's own pieces -- the curve equations y^2 = x^3 + 4 / y^2 = x^3 + 4 (1 + u) (src/curves/g1_curve_target.rs,
// g2_curve_target.rs), the square roots above and sgn0_fq / sgn0_fq2 (src/fields/helpers.rs:159-174).  y = 0 does not occur:
// both group orders are odd, so neither curve has a point of order two.
BLS_HD FpR curve_rhs(const FpR& x) { return fit<1, 2>(add(mul(sqr(x), x), dbl(dbl(fp_one())))); }
BLS_HD Fp2R curve_rhs(const Fp2R& x) { return fit<1, 2>(add(mul(sqr(x), x), dbl(dbl(mul_xi(fp2_one()))))); }
BLS_HD FpR decompress_y(const FpR& x, bool sgn, uint8_t& flags) { return witness_fq_sqrt(curve_rhs(x), sgn, flags); }
BLS_HD Fp2R decompress_y(const Fp2R& x, bool sgn, uint8_t& flags) { return witness_fq2_sqrt(curve_rhs(x), sgn, flags); }

}  // namespace bls
