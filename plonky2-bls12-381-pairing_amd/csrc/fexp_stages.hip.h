// Staged final exponentiation with Karabina-compressed cyclotomic squarings (MI355X-side algorithm; same field values
// as src/fields_as_trees/miller_loop.rs:128-178, hence the same output bits).
//
// After the easy part every value lives in the cyclotomic subgroup, where a squaring needs only four of the six Fp2
// coordinates (z2..z5 in the naming of `cyclotomic_square`, miller_loop.rs:46-104): the Granger-Scott formulas for
// z2', z3', z4', z5' never read z0, z1, so dropping that pair saves one of the three Fp4 squarings -- a third of the
// work of the 5 x 63 squarings that are ~78 % of the final exponentiation.  Getting z0, z1 back (Karabina 2013),
//     z1 = (xi z5^2 + 3 z4^2 - 2 z3) / (4 z2)        [ z1 = 2 z4 z5 / z3  when z2 = 0 ]
//     z0 = (2 z1^2 + z2 z5 - 3 z3 z4) xi + 1,
// costs a division, i.e. one Fp inversion per exponentiation by |x| -- more than the squarings save when every lane pays
// its own Fermat inversion.  A GPU batch does not have to: the exponentiation is cut at the inversion, the Fp norms of
// all elements go through ONE simultaneous inversion (k_fp_batch_inv, Montgomery's trick), and the next kernel resumes.
// f^|x| = f^(2^63) f^(2^62) f^(2^60) f^(2^57) f^(2^48) f^(2^16): the compressed chain is walked once, the six marked
// powers are parked in scratch, their six denominators share one inversion, then five multiplications.
//
// Six stage kernels (k_fexp.hip) with a batch inversion between consecutive ones.  The hard part is walked along
//     3 (p^4 - p^2 + 1) / r  =  (x - 1)^2 (x + p) (x^2 + p^2 - 1) + 3          (Hayashida-Hayasaka-Teruya; integer identity,
// checked in tests/test_oracle_kats.py), the SAME exponent f_conversion (:137-163) reaches with its own addition chain -- hence
// the same field element, bit for bit -- but with 9 Fq12 products, 3 Frobenius maps and 1 cyclotomic squaring around the five
// exponentiations by x instead of 12, 8 and 2, and three parked Fq12 values instead of six.  With m the easy-part value and
// E(g) = g^x = conj(g^|x|):
//   1  f^-1 from the supplied norm inverse, easy part -> m (:166-178, :133-136);        start E(m)
//   2  a = E(m) conj(m)               = m^(x-1);                                        start E(a)
//   3  b = E(a) conj(a)               = a^(x-1);                                        start E(b)
//   4  c = E(b) frob(b)               = b^(x+p);                                        start E(c)
//   5  d1 = E(c);  W = frob^2(c) conj(c) (m^2 m);                                       start E(d1)
//   6  result = E(d1) W               = c^(x^2+p^2-1) m^3
// The one-kernel forms (tower.hip.h final_exponentiation*) keep the reference's own chain; tests compare all of them.
#pragma once
#include "tower.hip.h"

namespace bls {

struct Fp12C { Fp2R z2, z3, z4, z5; };       // compressed cyclotomic element

BLS_HD Fp12C compress(const Fp12R& f) { return Fp12C{f.c1.c0, f.c0.c2, f.c0.c1, f.c1.c2}; }

// The z2..z5 half of cyclotomic_square (tower.hip.h).  Two forms with the same values:
//   BLS_CSQR_PRODUCTS = 0   verbatim: two Fp4 squarings = six Fp2 squarings
//   BLS_CSQR_PRODUCTS = 1   z2^2 + xi z3^2 = (z2 + z3)(z2 + xi z3) - (1 + xi) z2 z3  and  2 z2 z3 from the same product:
//                           four Fp2 multiplications.  Per lane the multiply-add count is the same (4 x 588 = 6 x 392), but
//                           four operand exchanges and four reductions replace six.
#ifndef BLS_CSQR_PRODUCTS
#define BLS_CSQR_PRODUCTS 1
#endif
BLS_HD Fp12C compressed_square(const Fp12C& g) {
#if BLS_CSQR_PRODUCTS
  const Fp2R b23 = mul(g.z2, g.z3), b45 = mul(g.z4, g.z5);
  const Fp2R a23 = mul(add(g.z2, g.z3), add(g.z2, mul_xi(g.z3)));
  const Fp2R a45 = mul(add(g.z4, g.z5), add(g.z4, mul_xi(g.z5)));
  const auto xb23 = mul_xi(b23), xb45 = mul_xi(b45);
  const auto ba = sub(a23, add(b23, xb23));                 // z2^2 + xi z3^2          (fp4_square(z2, z3).a)
  const auto ca = sub(a45, add(b45, xb45));                 // z4^2 + xi z5^2          (fp4_square(z4, z5).a)
  const auto n4 = sub(add(dbl(ba), ba), dbl(g.z4));         // 3 B.a - 2 z4
  const auto n3 = sub(add(dbl(ca), ca), dbl(g.z3));         // 3 C.a - 2 z3
  const auto b23x2 = dbl(b23);
  const auto n5 = dbl(add(add(b23x2, b23), g.z5));          // 3 (2 z2 z3) + 2 z5
  const auto xb45x2 = dbl(fit<1, 8>(xb45));
  const auto n2 = dbl(add(add(xb45x2, fit<1, 8>(xb45)), g.z2));   // 3 xi (2 z4 z5) + 2 z2
  return Fp12C{fit<1, 2>(n2), fit<1, 2>(n3), fit<1, 2>(n4), fit<1, 2>(n5)};
#else
  const auto B = fp4_square(g.z2, g.z3);
  const auto C = fp4_square(g.z4, g.z5);
  const auto n4 = add(dbl(sub(B.a, g.z4)), B.a);
  const auto n5 = add(dbl(add(B.b, g.z5)), B.b);
  const auto t0 = mul_xi(fit<1, 2>(C.b));
  const auto n2 = add(dbl(add(t0, g.z2)), t0);
  const auto n3 = add(dbl(sub(C.a, g.z3)), C.a);
  return Fp12C{fit<1, 2>(n2), fit<1, 2>(n3), fit<1, 2>(n4), fit<1, 2>(n5)};
#endif
}

// denominator of z1 (4 z2, or z3 when z2 = 0) with zero replaced by one, and whether it was zero: then z1 = 0, which
// is right for the only such element that occurs, the identity (z2 = z3 = z4 = z5 = 0 -> z1 = 0, z0 = 1)
BLS_HD Fp2R decompress_den(const Fp12C& g, bool& was_zero) {
  const Fp2R d = select(is_zero(g.z2), g.z3, Fp2R(fit<1, 2>(dbl(dbl(g.z2)))));
  was_zero = is_zero(d);
  return select(was_zero, Fp2R(fp2_one()), d);
}

BLS_HD Fp12R decompress(const Fp12C& g, const Fp2R& den_inv) {
  Fp2R num;
  if (is_zero(g.z2)) {
    num = fit<1, 2>(dbl(mul(g.z4, g.z5)));
  } else {
    const Fp2R s4 = sqr(g.z4);
    num = fit<1, 2>(sub(add(mul_xi(sqr(g.z5)), add(dbl(s4), s4)), dbl(g.z3)));
  }
  const Fp2R z1 = mul(num, den_inv);
  // 2 z1^2 + z2 z5 - 3 z3 z4 as one lazy sum (fp.hip.h dot): five Fp products and one reduction per lane instead of 1 + 2 + 2 and three
  const auto m3z3 = neg(add(dbl(g.z3), g.z3));                 // limb bounds of the three products add up to exactly 17 (fp.hip.h)
  const Fp2R inner = dot2(pr2(dbl(z1), z1), pr2(g.z2, g.z5), pr2(m3z3, g.z4));
  const Fp2R z0 = fit<1, 2>(add(mul_xi(inner), fp2_one()));
  return Fp12R(Fp6R(z0, g.z4, g.z3), Fp6R(g.z2, z1, g.z5));
}

// ------------------------------------------------------------------------------------------ scratch layout
// Per element, in units of Fp2 values inside a LaneScratch (fp.hip.h): three Fq12 slots, the six compressed powers of the
// running exponentiation, and the prefix products of their denominators.
enum : int { FX_M = 0, FX_A = 1, FX_W = 2, FX_SLOTS = 3 };     // m; a, then b, then c; W
constexpr int FX_COMP_BASE = FX_SLOTS * 6;
constexpr int FX_PRE_BASE = FX_COMP_BASE + 6 * 4;
constexpr int FX_X_BASE = FX_PRE_BASE + 6;                                       // input of the next squaring chain (split launch)
constexpr int FX_FP2_TOTAL = FX_X_BASE + 4;                                      // 52 Fp2 per element
constexpr int FX_FP_PER_LANE = FX_FP2_TOTAL * SCRATCH_FP_PER_FP2;
constexpr int FX_DWORDS_PER_LANE = FX_FP_PER_LANE * LANE_SCRATCH_DWORDS;         // 14 dwords per parked Fp
BLS_HD LaneScratch fx_lane_scratch(uint32_t* buf, size_t lanes, size_t lane) { return lane_scratch(buf, lanes, lane, FX_FP_PER_LANE); }

struct FexpScratch {
  LaneScratch ls;          // stage values, lane-sliced
  uint32_t* norm_raw;      // RAW_FP_STRIDE dwords per element: norm out / norm inverse in (k_fp_batch_inv in between)
  size_t elem;
};

BLS_HD void fx_put12(const LaneScratch& s, int slot, const Fp12R& a) {
  fp2_scratch_put(s, slot * 6 + 0, a.c0.c0); fp2_scratch_put(s, slot * 6 + 1, a.c0.c1); fp2_scratch_put(s, slot * 6 + 2, a.c0.c2);
  fp2_scratch_put(s, slot * 6 + 3, a.c1.c0); fp2_scratch_put(s, slot * 6 + 4, a.c1.c1); fp2_scratch_put(s, slot * 6 + 5, a.c1.c2);
}
BLS_HD Fp12R fx_get12(const LaneScratch& s, int slot) {
  return Fp12R(Fp6R(fp2_scratch_get(s, slot * 6 + 0), fp2_scratch_get(s, slot * 6 + 1), fp2_scratch_get(s, slot * 6 + 2)),
               Fp6R(fp2_scratch_get(s, slot * 6 + 3), fp2_scratch_get(s, slot * 6 + 4), fp2_scratch_get(s, slot * 6 + 5)));
}
BLS_HD void fx_putc(const LaneScratch& s, int m, const Fp12C& g) {
  fp2_scratch_put(s, FX_COMP_BASE + 4 * m + 0, g.z2); fp2_scratch_put(s, FX_COMP_BASE + 4 * m + 1, g.z3);
  fp2_scratch_put(s, FX_COMP_BASE + 4 * m + 2, g.z4); fp2_scratch_put(s, FX_COMP_BASE + 4 * m + 3, g.z5);
}
BLS_HD void fx_putx(const LaneScratch& s, const Fp12C& g) {
  fp2_scratch_put(s, FX_X_BASE + 0, g.z2); fp2_scratch_put(s, FX_X_BASE + 1, g.z3);
  fp2_scratch_put(s, FX_X_BASE + 2, g.z4); fp2_scratch_put(s, FX_X_BASE + 3, g.z5);
}
BLS_HD Fp12C fx_getx(const LaneScratch& s) {
  return Fp12C{fp2_scratch_get(s, FX_X_BASE + 0), fp2_scratch_get(s, FX_X_BASE + 1), fp2_scratch_get(s, FX_X_BASE + 2), fp2_scratch_get(s, FX_X_BASE + 3)};
}
BLS_HD Fp12C fx_getc(const LaneScratch& s, int m) {
  return Fp12C{fp2_scratch_get(s, FX_COMP_BASE + 4 * m + 0), fp2_scratch_get(s, FX_COMP_BASE + 4 * m + 1),
               fp2_scratch_get(s, FX_COMP_BASE + 4 * m + 2), fp2_scratch_get(s, FX_COMP_BASE + 4 * m + 3)};
}

// ------------------------------------------------------------------------------------------ one exponentiation, in two halves
// bits of |x| = 0xd201000000010000 above bit 0: the powers f^(2^k) that make up f^|x|
constexpr uint64_t FX_MARKS = BLS_X_PARAM;
static_assert(FX_MARKS == ((1ull << 63) | (1ull << 62) | (1ull << 60) | (1ull << 57) | (1ull << 48) | (1ull << 16)), "six marked powers");

// first half: walk the compressed squaring chain, park the six marked powers and the prefix products of their
// denominators, leave the Fp norm of the full denominator product for the batch inversion
BLS_HD void fexp_chain(const Fp12C& x, const FexpScratch& sc) {
  Fp12C g = x;
  Fp2R prod = Fp2R(fp2_one());
  int m = 0;
  for (int k = 1; k <= 63; ++k) {
    g = compressed_square(g);
    if ((FX_MARKS >> k) & 1ull) {
      fx_putc(sc.ls, m, g);
      fp2_scratch_put(sc.ls, FX_PRE_BASE + m, prod);
      bool z;
      const Fp2R d = decompress_den(g, z);
      prod = mul(prod, d);
      ++m;
    }
  }
  const FpR nrm = fp2_norm(prod);
  if (lane_part() == 0) raw_fp_store(sc.norm_raw + sc.elem * RAW_FP_STRIDE, nrm);
}
BLS_HD_NOINLINE void fexp_start_exp(const Fp12R& x, const FexpScratch& sc) { fexp_chain(compress(x), sc); }

// second half: conj(x^|x|) from the parked powers and the inverse of the norm (== cyclotomic_exp(x), tower.hip.h)
BLS_FN_COLD Fp12R fexp_finish_exp(const FexpScratch& sc) {
  const FpR norm_inv = raw_fp_load(sc.norm_raw + sc.elem * RAW_FP_STRIDE);
  // inverse of the product of all six denominators; peel one denominator per step, last power first
  Fp2R prod = fp2_scratch_get(sc.ls, FX_PRE_BASE + 5);
  {
    bool z;
    prod = mul(prod, decompress_den(fx_getc(sc.ls, 5), z));
  }
  Fp2R run = fp2_inv_given(prod, norm_inv);
  Fp12R acc;
  for (int m = 5; m >= 0; --m) {
    const Fp12C g = fx_getc(sc.ls, m);
    bool z;
    const Fp2R d = decompress_den(g, z);
    const Fp2R pre = fp2_scratch_get(sc.ls, FX_PRE_BASE + m);
    const Fp2R dinv = select(z, Fp2R(fp2_zero()), mul(run, pre));
    run = mul(run, d);
    const Fp12R v = decompress(g, dinv);
    acc = (m == 5) ? v : mul_call(acc, v);
  }
  return fit<1, 2>(conj(acc));
}

// ------------------------------------------------------------------------------------------ the six stages
BLS_HD Fp12R frobenius_n(const Fp12R& a, int n) {
  Fp12R r = a;
  for (int i = 0; i < n; ++i) r = frobenius_call(r);
  return r;
}

// Each stage = its glue (Fq12 products around the exponentiation just finished; returns the value whose exponentiation starts
// next) followed by that value's squaring chain.  The two halves can run fused in one kernel (k_fexp_stage<i>, two waves per
// SIMD) or as two launches with the glue compiled for one wave per SIMD (k_fexp_glue + k_fexp_chain, k_fexp_wide.hip / k_fexp.hip).
// The pass in front of stage 1: the Fp norm under f^-1 (fq12_target_tree.rs:77-90 bottoms out in one Fp inversion) for the batch
// inversion, and the four Fp2 values of the inversion that stage 1 needs again, parked in the (not yet used) chain-input slots
// instead of being recomputed (~9 400 multiply-adds per lane).
BLS_HD FpR fexp_norm_pass(const Fp12R& f, const LaneScratch& ls) {
  Fp6InvState st;
  const FpR nrm = fp12_inv_begin(f, st);
  fp2_scratch_put(ls, FX_X_BASE + 0, st.c0); fp2_scratch_put(ls, FX_X_BASE + 1, st.c1);
  fp2_scratch_put(ls, FX_X_BASE + 2, st.c2); fp2_scratch_put(ls, FX_X_BASE + 3, st.t);
  return nrm;
}
BLS_HD Fp12R fexp_glue1(const Fp12R& f, const FexpScratch& sc) {
  const FpR f_norm_inv = raw_fp_load(sc.norm_raw + sc.elem * RAW_FP_STRIDE);
  Fp6InvState st;
  st.c0 = fp2_scratch_get(sc.ls, FX_X_BASE + 0); st.c1 = fp2_scratch_get(sc.ls, FX_X_BASE + 1);
  st.c2 = fp2_scratch_get(sc.ls, FX_X_BASE + 2); st.t = fp2_scratch_get(sc.ls, FX_X_BASE + 3);
  const Fp12R f_inv = fp12_inv_end(f, st, f_norm_inv);
  Fp12R m = mul_call(fit<1, 2>(conj(f)), f_inv);                        // :168-175, :133
  m = mul_call(frobenius_n(m, 2), m);                                   // :135-136
  fx_put12(sc.ls, FX_M, m);
  return m;
}
BLS_HD Fp12R fexp_glue2(const FexpScratch& sc) {
  const Fp12R a = mul_call(fexp_finish_exp(sc), fit<1, 2>(conj(fx_get12(sc.ls, FX_M))));
  fx_put12(sc.ls, FX_A, a);
  return a;
}
BLS_HD Fp12R fexp_glue3(const FexpScratch& sc) {
  const Fp12R b = mul_call(fexp_finish_exp(sc), fit<1, 2>(conj(fx_get12(sc.ls, FX_A))));
  fx_put12(sc.ls, FX_A, b);
  return b;
}
BLS_HD Fp12R fexp_glue4(const FexpScratch& sc) {
  const Fp12R c = mul_call(fexp_finish_exp(sc), frobenius_n(fx_get12(sc.ls, FX_A), 1));
  fx_put12(sc.ls, FX_A, c);
  return c;
}
BLS_HD Fp12R fexp_glue5(const FexpScratch& sc) {
  const Fp12R d1 = fexp_finish_exp(sc);
  const Fp12R m = fx_get12(sc.ls, FX_M);
  const Fp12R m3 = mul_call(cyclotomic_square_call(m), m);
  const Fp12R c = fx_get12(sc.ls, FX_A);
  const Fp12R w = mul_call(frobenius_n(c, 2), fit<1, 2>(conj(c)));
  fx_put12(sc.ls, FX_W, mul_call(w, m3));
  return d1;
}
BLS_HD void fexp_stage1(const Fp12R& f, const FexpScratch& sc) { fexp_start_exp(fexp_glue1(f, sc), sc); }
BLS_HD void fexp_stage2(const FexpScratch& sc) { fexp_start_exp(fexp_glue2(sc), sc); }
BLS_HD void fexp_stage3(const FexpScratch& sc) { fexp_start_exp(fexp_glue3(sc), sc); }
BLS_HD void fexp_stage4(const FexpScratch& sc) { fexp_start_exp(fexp_glue4(sc), sc); }
BLS_HD void fexp_stage5(const FexpScratch& sc) { fexp_start_exp(fexp_glue5(sc), sc); }
BLS_HD Fp12R fexp_stage6(const FexpScratch& sc) {
  return mul_call(fexp_finish_exp(sc), fx_get12(sc.ls, FX_W));
}

}  // namespace bls
