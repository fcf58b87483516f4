// TEST INFRASTRUCTURE -- not product code, never linked into libbls381_hip.so.
//
// Compiles the device headers (plonky2-bls12-381-pairing_amd/csrc/*.hip.h) for the HOST with g++ and
// -DBLS_BOUND_CHECK so the `-m "not gpu"` suite can (1) check the kernels' algorithms against the oracle
// without a GPU and (2) verify at run time every static limb/value bound the lazy-reduction types claim.
// The product library has no CPU path: it fails loudly without a GPU.
#define BLS_BOUND_CHECK 1
#define BLS_LAZY_014 1      // the form the Miller kernels compile (k_multi.hip, k_groth16.hip); Karatsuba stays reachable as op 11
#include "pairing.hip.h"
#include "witness.hip.h"
#include "fexp_stages.hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace bls {

static long g_checks = 0;
long g_op_count[8] = {0};

void bound_check_fail(const char* what) {
  fprintf(stderr, "hostsim: BOUND VIOLATION: %s\n", what);
  abort();
}

// value < V * 1.01 * p  and  limb_i < L * 2^28 + 64 (top limb: value bound only)
void bound_check(const uint32_t* l, int L, int V) {
  ++g_checks;
  for (int i = 0; i < NL - 1; ++i) {
    if ((uint64_t)l[i] >= (uint64_t)L * (1ull << LW) + 64) {
      char buf[128]; snprintf(buf, sizeof buf, "limb %d = 0x%x exceeds L = %d", i, l[i], L);
      bound_check_fail(buf);
    }
  }
  // exact value as 15 x 28-bit normalised limbs
  uint64_t v[NL + 1];
  uint64_t carry = 0;
  for (int i = 0; i < NL; ++i) { uint64_t s = (uint64_t)l[i] + carry; v[i] = s & LMASK; carry = s >> LW; }
  v[NL] = carry;
  // bound = floor(V * 1.01 * p) ~ V*p + V*p/100 : compute V*101*p and compare 100*value
  uint64_t b[NL + 2] = {0}, x[NL + 2] = {0};
  carry = 0;
  for (int i = 0; i < NL; ++i) { uint64_t s = (uint64_t)BLS_P28[i] * (uint64_t)(V * 101) + carry; b[i] = s & LMASK; carry = s >> LW; }
  b[NL] = carry & LMASK; b[NL + 1] = carry >> LW;
  carry = 0;
  for (int i = 0; i <= NL; ++i) { uint64_t s = v[i] * 100ull + carry; x[i] = s & LMASK; carry = s >> LW; }
  x[NL + 1] = carry;
  for (int i = NL + 1; i >= 0; --i) {
    if (x[i] < b[i]) return;
    if (x[i] > b[i]) { char buf[128]; snprintf(buf, sizeof buf, "value exceeds V = %d (x 1.01 p)", V); bound_check_fail(buf); }
  }
  bound_check_fail("value equals the bound");
}

}  // namespace bls

using namespace bls;

extern "C" {

long hostsim_bound_checks() { return bls::g_checks; }
void hostsim_op_counts(long* out, int reset) { for (int i = 0; i < 8; ++i) { out[i] = bls::g_op_count[i]; if (reset) bls::g_op_count[i] = 0; } }

// op codes shared with include/bls381_hip.h
enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_SQR = 3, OP_INV = 4, OP_NEG = 5, OP_TO_MONT = 6, OP_FROM_MONT = 7 };

int hostsim_fp_op(int op, int form, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    if (op == OP_TO_MONT) { FpR x = fp_load(a + 6 * i, IO_CANONICAL); fp_store(out + 6 * i, x, IO_ARK_MONT); continue; }
    if (op == OP_FROM_MONT) { FpR x = fp_load(a + 6 * i, IO_ARK_MONT); fp_store(out + 6 * i, x, IO_CANONICAL); continue; }
    FpR x = fp_load(a + 6 * i, form);
    FpR y = b ? fp_load(b + 6 * i, form) : x;
    switch (op) {
      case OP_ADD: fp_store(out + 6 * i, add(x, y), form); break;
      case OP_SUB: fp_store(out + 6 * i, sub(x, y), form); break;
      case OP_MUL: fp_store(out + 6 * i, mul(x, y), form); break;
      case OP_SQR: fp_store(out + 6 * i, sqr(x), form); break;
      case OP_INV: fp_store(out + 6 * i, fp_inv(x), form); break;
      case 12: fp_store(out + 6 * i, fp_inv_fermat(x), form); break;      // a^(p-2): cross-check of the binary-GCD inversion
      case OP_NEG: fp_store(out + 6 * i, neg(x), form); break;
      case 8: fp_store(out + 6 * i, mul(x, fp_inv(y)), form); break;
      case 9: fp_store(out + 6 * i, fp_pow_const(x, BLS_P_MINUS_1_DIV_2), form); break;
      case 10: fp_store(out + 6 * i, select(fp_eq(fp_pow_const(x, BLS_P_MINUS_1_DIV_2), fp_one()), FpR(fp_one()), FpR(fp_zero())), form); break;
      case 11: fp_store(out + 6 * i, select(fp_sgn0(x), FpR(fp_one()), FpR(fp_zero())), form); break;
      default: return 1;
    }
  }
  return 0;
}

enum { OP2_MUL = 0, OP2_SQR = 1, OP2_INV = 2, OP2_MUL_XI = 3, OP2_CONJ = 4 };
int hostsim_fp2_op(int op, int form, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    Fp2R x = fp2_load(a + 12 * i, form);
    Fp2R y = b ? fp2_load(b + 12 * i, form) : x;
    switch (op) {
      case OP2_MUL: fp2_store(out + 12 * i, mul(x, y), form); break;
      case OP2_SQR: fp2_store(out + 12 * i, sqr(x), form); break;
      case OP2_INV: fp2_store(out + 12 * i, inv(x), form); break;
      case OP2_MUL_XI: fp2_store(out + 12 * i, mul_xi(x), form); break;
      case OP2_CONJ: fp2_store(out + 12 * i, conj(x), form); break;
      default: return 1;
    }
  }
  return 0;
}

enum { OP12_MUL = 0, OP12_SQR = 1, OP12_INV = 2, OP12_FROBENIUS = 3, OP12_CYC_SQR = 4, OP12_CONJ = 5,
       OP12_CYC_EXP = 6, OP12_FINAL_EXP = 7, OP12_FINAL_EXP_SPLIT = 8 };

// The split final exponentiation exactly as the three kernels run it (k_fexp_norm, k_fp_batch_inv in chunks of FP_BATCH_INV_CHUNK
// with the tail padded by ones, k_final_exp_given), raw limbs parked 16 dwords apart in between.
static int final_exp_split(int form, const uint64_t* a, uint64_t* out, size_t n) {
  constexpr int CHUNK = FP_BATCH_INV_CHUNK;
  uint32_t* raw = (uint32_t*)calloc(n ? n : 1, RAW_FP_STRIDE * sizeof(uint32_t));
  if (!raw) return 2;
  for (size_t i = 0; i < n; ++i) raw_fp_store(raw + i * RAW_FP_STRIDE, fp12_inv_norm_call(fp12_load(a + 72 * i, form)));
  for (size_t first = 0; first < n; first += CHUNK) {
    FpR x[CHUNK];
    for (int j = 0; j < CHUNK; ++j) x[j] = (first + j < n) ? raw_fp_load(raw + (first + j) * RAW_FP_STRIDE) : FpR(fp_one());
    fp_batch_inv(x);
    for (int j = 0; j < CHUNK; ++j) if (first + j < n) raw_fp_store(raw + (first + j) * RAW_FP_STRIDE, x[j]);
  }
  for (size_t i = 0; i < n; ++i) {
    const Fp12R f = fp12_load(a + 72 * i, form);
    const Fp12R f_inv = fp12_inv_given_call(f, raw_fp_load(raw + i * RAW_FP_STRIDE));
    fp12_store(out + 72 * i, final_exponentiation_given_inverse(f, f_inv), form);
  }
  free(raw);
  return 0;
}

static void batch_inv_raw(uint32_t* raw, size_t n) {
  constexpr int CHUNK = FP_BATCH_INV_CHUNK;
  for (size_t first = 0; first < n; first += CHUNK) {
    FpR x[CHUNK];
    for (int j = 0; j < CHUNK; ++j) x[j] = (first + j < n) ? raw_fp_load(raw + (first + j) * RAW_FP_STRIDE) : FpR(fp_one());
    fp_batch_inv(x);
    for (int j = 0; j < CHUNK; ++j) if (first + j < n) raw_fp_store(raw + (first + j) * RAW_FP_STRIDE, x[j]);
  }
}

// The staged final exponentiation (fexp_stages.hip.h) as the GPU runs it: norm pass, then six stages with a simultaneous
// inversion before each; every stage runs over the whole batch before the next one starts, all state in scratch.
// split = true: every stage as the glue pass (parks the chain input) followed by the chain pass, like the library's default launch
static int final_exp_staged(int form, const uint64_t* a, uint64_t* out, size_t n, bool split = false) {
  uint32_t* raw = (uint32_t*)calloc(n ? n : 1, RAW_FP_STRIDE * sizeof(uint32_t));
  uint32_t* scratch = (uint32_t*)calloc((n ? n : 1) * (size_t)FX_DWORDS_PER_LANE, sizeof(uint32_t));
  if (!raw || !scratch) return 2;
  auto sc = [&](size_t i) { return FexpScratch{fx_lane_scratch(scratch, n, i), raw, i}; };
  for (size_t i = 0; i < n; ++i) raw_fp_store(raw + i * RAW_FP_STRIDE, fexp_norm_pass(fp12_load(a + 72 * i, form), sc(i).ls));
  batch_inv_raw(raw, n);
  if (split) {
    for (int st = 1; st <= 5; ++st) {
      for (size_t i = 0; i < n; ++i) {
        const FexpScratch c = sc(i);
        const Fp12R x = st == 1 ? fexp_glue1(fp12_load(a + 72 * i, form), c) : st == 2 ? fexp_glue2(c) : st == 3 ? fexp_glue3(c)
                      : st == 4 ? fexp_glue4(c) : fexp_glue5(c);
        fx_putx(c.ls, compress(x));
      }
      for (size_t i = 0; i < n; ++i) { const FexpScratch c = sc(i); fexp_chain(fx_getx(c.ls), c); }
      batch_inv_raw(raw, n);
    }
  } else {
  for (size_t i = 0; i < n; ++i) fexp_stage1(fp12_load(a + 72 * i, form), sc(i));
  batch_inv_raw(raw, n);
  for (size_t i = 0; i < n; ++i) fexp_stage2(sc(i));
  batch_inv_raw(raw, n);
  for (size_t i = 0; i < n; ++i) fexp_stage3(sc(i));
  batch_inv_raw(raw, n);
  for (size_t i = 0; i < n; ++i) fexp_stage4(sc(i));
  batch_inv_raw(raw, n);
  for (size_t i = 0; i < n; ++i) fexp_stage5(sc(i));
  batch_inv_raw(raw, n);
  }
  for (size_t i = 0; i < n; ++i) fp12_store(out + 72 * i, fexp_stage6(sc(i)), form);
  free(raw); free(scratch);
  return 0;
}

int hostsim_fp12_op(int op, int form, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  if (op == OP12_FINAL_EXP_SPLIT) return final_exp_split(form, a, out, n);
  if (op == 9) return final_exp_staged(form, a, out, n);
  if (op == 12) return final_exp_staged(form, a, out, n, true);
  for (size_t i = 0; i < n; ++i) {
    Fp12R x = fp12_load(a + 72 * i, form);
    Fp12R y = b ? fp12_load(b + 72 * i, form) : x;
    Fp12R r;
    switch (op) {
      case OP12_MUL: r = mul(x, y); break;
      case OP12_SQR: r = sqr(x); break;
      case OP12_INV: r = inv(x); break;
      case OP12_FROBENIUS: r = frobenius(x); break;
      case OP12_CYC_SQR: r = cyclotomic_square(x); break;
      case OP12_CONJ: r = fit<1, 2>(conj(x)); break;
      case OP12_CYC_EXP: r = cyclotomic_exp(x); break;
      case OP12_FINAL_EXP: r = final_exponentiation(x); break;
      // sparse line multiplication, factor (c0, c1, c4) = (y.c0.c0, y.c0.c1, y.c1.c1): 10 = the form the library's Miller
      // kernels compile (lazy sums, BLS_LAZY_014), 11 = Karatsuba as the reference writes it (fq12_target_tree.rs:157-176)
      case 10: r = mul_by_014_lazy(x, y.c0.c0, y.c0.c1, y.c1.c1); break;
      case 11: r = mul_by_014(x, y.c0.c0, y.c0.c1, y.c1.c1); break;
      case 13: r = sqr_twice(x); break;                                   // 2 x^2 (Chung-Hasan SQR3 over Fp4)
      default: return 1;
    }
    fp12_store(out + 72 * i, r, form);
  }
  return 0;
}

int hostsim_miller_ref_opt(int form, int generic, const uint64_t* g1, const uint64_t* g2, uint64_t* out,
                           uint8_t* flags, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    G1Jac p = g1_jac_load(g1 + 18 * i, form);
    G2Jac q = g2_jac_load(g2 + 36 * i, form);
    uint8_t fl = 0;
    Fp12R f = generic ? miller_ref_opt<true>(p, q, fl) : miller_ref_opt<false>(p, q, fl);
    fp12_store(out + 72 * i, f, form);
    if (flags) flags[i] = fl;
  }
  return 0;
}

int hostsim_line_ref_opt(int form, const uint64_t* q1, const uint64_t* q2, const uint64_t* p, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const G2Jac a = g2_jac_load(q1 + 36 * i, form), b = g2_jac_load(q2 + 36 * i, form);
    const G1Jac pt = g1_jac_load(p + 18 * i, form);
    const LineND l = line_chord(a, b, pt);
    fp2_store(out + 24 * i, l.n, form);
    fp2_store(out + 24 * i + 12, l.d, form);
  }
  return 0;
}

// bls381_g1_op_batch (group 1) / bls381_g2_op_batch (group 2), same element routine as k_point_op
int hostsim_point_op(int group, int op, int form, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* flags, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    if (group == 1) {
      G1Jac x, y, r; jac_load(x, a + 18 * i, form); if (b) jac_load(y, b + 18 * i, form); else y = x;
      flags[i] = point_op(op, x, y, r); jac_store(out + 18 * i, r, form);
    } else {
      G2Jac x, y, r; jac_load(x, a + 36 * i, form); if (b) jac_load(y, b + 36 * i, form); else y = x;
      flags[i] = point_op(op, x, y, r); jac_store(out + 36 * i, r, form);
    }
  }
  return 0;
}

int hostsim_fp_pow(int form, const uint64_t* a, const uint64_t* e, int nwords, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; ++i) fp_store(out + 6 * i, fp_pow_words(fp_load(a + 6 * i, form), e, nwords), form);
  return 0;
}
int hostsim_g2_prepare(int form, const uint64_t* g2, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    bool inf; G2Aff q = g2_aff_load(g2 + 24 * i, form, inf);
    uint64_t* o = out + (size_t)68 * 36 * i;
    g2_prepare(q, [&](int idx, const LineCoeffs& l) {
      fp2_store(o + 36 * idx, l.a, form); fp2_store(o + 36 * idx + 12, l.b, form); fp2_store(o + 36 * idx + 24, l.c, form);
    });
  }
  return 0;
}
int hostsim_witness_inverse(int op, const uint64_t* x, uint64_t* out, uint8_t* flags, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    uint8_t fl = 0;
    if (op == 0) { const Fp12R a = fp12_load_my(x + 72 * i); if (is_zero(a)) fl |= WFLAG_NO_INVERSE; fp12_store_my(out + 72 * i, inv(a)); }
    else if (op == 1) { const Fp6R a = fp6_load_my(x + 36 * i); if (is_zero(a)) fl |= WFLAG_NO_INVERSE; fp6_store_my(out + 36 * i, inv(a)); }
    else { const Fp2R a = fp2_load(x + 12 * i, IO_CANONICAL); fp2_store(out + 12 * i, inv(a), IO_CANONICAL); }
    if (flags) flags[i] = fl;
  }
  return 0;
}
int hostsim_witness_sqrt(int op, const uint64_t* x, const uint8_t* sgn, uint64_t* out, uint8_t* flags, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    uint8_t fl = 0;
    const bool want = (sgn[i] & 1) != 0;
    if (op == 0) fp_store(out + 6 * i, witness_fq_sqrt(fp_load(x + 6 * i, IO_CANONICAL), want, fl), IO_CANONICAL);
    else fp2_store(out + 12 * i, witness_fq2_sqrt(fp2_load(x + 12 * i, IO_CANONICAL), want, fl), IO_CANONICAL);
    if (flags) flags[i] = fl;
  }
  return 0;
}

}  // extern "C"

template <int K>
static void multi_one(int form, int with_final_exp, const uint64_t* g1, const uint64_t* g2, uint64_t* out, uint8_t* flag) {
  G1Aff p[K]; G2Aff q[K]; bool skip[K];
  uint8_t fl = 0;
  for (int j = 0; j < K; ++j) {
    bool i1, i2;
    p[j] = g1_aff_load(g1 + 12 * j, form, i1);
    q[j] = g2_aff_load(g2 + 24 * j, form, i2);
    skip[j] = i1 || i2;
    if (skip[j]) fl |= FLAG_INFINITY;
  }
  // like the GPU: full pairings walk the Miller loop in homogeneous coordinates, Miller-only results keep Alg. 26/27
  Fp12R f = with_final_exp ? miller_ate<K, true>(p, q, skip) : miller_ate<K, false>(p, q, skip);
  if (with_final_exp) f = final_exponentiation(f);
  fp12_store(out, f, form);
  if (flag) *flag = fl;
}

// k_prepared_miller's loop (pairing.hip.h miller_prepared) with the prepared coefficients in a plain array; always followed by
// the final exponentiation.  prepared: nfixed x 68 x 3 x 2 Fp in ABI form (bls381_g2_prepare_batch layout).
template <int NVAR, int NFIXED>
static void prepared_one(int form, int hom, const uint64_t* g1, const uint64_t* g2v, const FpR* coef, uint64_t* out, uint8_t* flag) {
  constexpr int NP = NVAR + NFIXED;
  G1Aff p[NP]; bool skip[NP]; uint8_t fl = 0;
  for (int j = 0; j < NP; ++j) { bool inf; p[j] = g1_aff_load(g1 + 12 * j, form, inf); skip[j] = inf; }
  G2Aff q0{};
  if (NVAR == 1) { bool qinf; q0 = g2_aff_load(g2v, form, qinf); skip[0] = skip[0] || qinf; }
  for (int j = 0; j < NP; ++j) if (skip[j]) fl |= FLAG_INFINITY;
  auto fixed_line = [&](int j, int idx) {
    const FpR* b = coef + (size_t)((j * 68 + idx) * 3) * 2;
    LineCoeffs l;
    l.a = Fp2R(b[0], b[1]); l.b = Fp2R(b[2], b[3]); l.c = Fp2R(b[4], b[5]);
    return l;
  };
  Fp12R f = hom ? miller_prepared<NVAR, NFIXED, true>(p, q0, skip, fixed_line) : miller_prepared<NVAR, NFIXED, false>(p, q0, skip, fixed_line);
  fp12_store(out, final_exponentiation(f), form);
  if (flag) *flag = fl;
}

// the LDS forms of the device kernels (pairing.hip.h miller_ate_lds, miller_ref_opt_generic_lds) over a plain buffer: one "lane"
template <int K>
static void multi_one_lds(int form, int with_final_exp, const uint64_t* g1, const uint64_t* g2, uint64_t* out, uint8_t* flag) {
  LdsQuad mem[4 * miller_lds_slots(K)];
  const LaneLds<1> lds = lane_lds<1>(mem, 0);
  uint8_t fl = 0;
  Fp12R f = with_final_exp ? miller_ate_lds<K, true>(g1, g2, form, lds, fl) : miller_ate_lds<K, false>(g1, g2, form, lds, fl);
  if (with_final_exp) f = final_exponentiation(f);
  fp12_store(out, f, form);
  if (flag) *flag = fl;
}

extern "C" {

int hostsim_prepared_pairing(int form, int nvar, int nfixed, int hom, const uint64_t* g1, const uint64_t* g2v, const uint64_t* prepared,
                             uint64_t* out, uint8_t* flags, size_t n) {
  const size_t ncoef = (size_t)nfixed * 68 * 3 * 2;
  FpR* coef = new FpR[ncoef];
  for (size_t t = 0; t < ncoef; ++t) coef[t] = fp_load(prepared + 6 * t, form);
  int rc = 0;
  for (size_t i = 0; i < n && rc == 0; ++i) {
    const uint64_t* a = g1 + 12 * (size_t)(nvar + nfixed) * i; const uint64_t* b = nvar ? g2v + 24 * i : nullptr;
    uint8_t* fl = flags ? flags + i : nullptr;
    if (nvar == 1 && nfixed == 2) prepared_one<1, 2>(form, hom, a, b, coef, out + 72 * i, fl);
    else if (nvar == 1 && nfixed == 1) prepared_one<1, 1>(form, hom, a, b, coef, out + 72 * i, fl);
    else if (nvar == 0 && nfixed == 1) prepared_one<0, 1>(form, hom, a, b, coef, out + 72 * i, fl);
    else if (nvar == 0 && nfixed == 2) prepared_one<0, 2>(form, hom, a, b, coef, out + 72 * i, fl);
    else rc = 1;
  }
  delete[] coef;
  return rc;
}

int hostsim_multi_miller_lds(int form, int with_final_exp, const uint64_t* g1, const uint64_t* g2, size_t k,
                             uint64_t* out, uint8_t* flags, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const uint64_t* a = g1 + 12 * k * i; const uint64_t* b = g2 + 24 * k * i;
    uint8_t* fl = flags ? flags + i : nullptr;
    switch (k) {
      case 2: multi_one_lds<2>(form, with_final_exp, a, b, out + 72 * i, fl); break;
      case 3: multi_one_lds<3>(form, with_final_exp, a, b, out + 72 * i, fl); break;
      default: return 1;
    }
  }
  return 0;
}

int hostsim_miller_ref_opt_lds(int form, const uint64_t* g1, const uint64_t* g2, uint64_t* out, uint8_t* flags, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    LdsQuad mem[ref_opt_lds_bytes_per_lane() / sizeof(LdsQuad)];
    const LaneLds<1> lds = lane_lds<1>(mem, 0);
    const G1Jac p = g1_jac_load(g1 + 18 * i, form);
    uint8_t fl = 0;
    const Fp12R f = miller_ref_opt_generic_lds(p, g2 + 36 * i, form, lds, fl);
    fp12_store(out + 72 * i, f, form);
    if (flags) flags[i] = fl;
  }
  return 0;
}

int hostsim_multi_miller(int form, int with_final_exp, const uint64_t* g1, const uint64_t* g2, size_t k,
                         uint64_t* out, uint8_t* flags, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const uint64_t* a = g1 + 12 * k * i; const uint64_t* b = g2 + 24 * k * i;
    uint8_t* fl = flags ? flags + i : nullptr;
    switch (k) {
      case 1: multi_one<1>(form, with_final_exp, a, b, out + 72 * i, fl); break;
      case 2: multi_one<2>(form, with_final_exp, a, b, out + 72 * i, fl); break;
      case 3: multi_one<3>(form, with_final_exp, a, b, out + 72 * i, fl); break;
      default: return 1;
    }
  }
  return 0;
}

}  // extern "C"
