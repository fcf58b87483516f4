// Kernel translation unit of libbls381_hip.so (see bls381_hip.hip for the C ABI).  The library is split into
// several TUs only to compile them in parallel; each exposes plain host launch functions (launch.h).
#include <hip/hip_runtime.h>
#include "../../include/bls381_hip.h"
#include "pairing.hip.h"
#include "launch.h"
using namespace bls;
#ifndef BLS_MAX_BLOCK
#define BLS_MAX_BLOCK 256
#endif
static inline unsigned grid_for(size_t n, int block) { return (unsigned)((n * bls::LANES_PER_ELEM + block - 1) / block); }

__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_fp_op(int op, const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint64_t* __restrict__ out, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  if (op == BLS381_FP_TO_MONT) { fp_store(out + 6 * i, fp_load(a + 6 * i, IO_CANONICAL), IO_ARK_MONT); return; }
  if (op == BLS381_FP_FROM_MONT) { fp_store(out + 6 * i, fp_load(a + 6 * i, IO_ARK_MONT), IO_CANONICAL); return; }
  const FpR x = fp_load(a + 6 * i, form);
  const FpR y = b ? fp_load(b + 6 * i, form) : x;
  switch (op) {
    case BLS381_FP_ADD: fp_store(out + 6 * i, add(x, y), form); break;
    case BLS381_FP_SUB: fp_store(out + 6 * i, sub(x, y), form); break;
    case BLS381_FP_MUL: fp_store(out + 6 * i, mul(x, y), form); break;
    case BLS381_FP_SQR: fp_store(out + 6 * i, sqr(x), form); break;
    case BLS381_FP_INV: fp_store(out + 6 * i, fp_inv(x), form); break;
    case BLS381_FP_NEG: fp_store(out + 6 * i, neg(x), form); break;
    case BLS381_FP_DIV: fp_store(out + 6 * i, mul(x, fp_inv(y)), form); break;
    case BLS381_FP_LEGENDRE: fp_store(out + 6 * i, fp_pow_const(x, BLS_P_MINUS_1_DIV_2), form); break;
    case BLS381_FP_IS_SQUARE: {   // legendre == one (fq_target.rs:276-280): zero is not a square by this test
      const bool sq = fp_eq(fp_pow_const(x, BLS_P_MINUS_1_DIV_2), fp_one());
      fp_store(out + 6 * i, select(sq, FpR(fp_one()), FpR(fp_zero())), form); break; }
    case BLS381_FP_SGN0: fp_store(out + 6 * i, select(fp_sgn0(x), FpR(fp_one()), FpR(fp_zero())), form); break;
    default: break;
  }
}

__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_fp12_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int form) {   // out may be a or b (in place)
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  const Fp12R x = fp12_load(a + 72 * i, form);
  Fp12R r = x;
  switch (op) {
    case BLS381_FP12_MUL: { const Fp12R y = b ? fp12_load(b + 72 * i, form) : x; r = mul(x, y); break; }
    case BLS381_FP12_SQR: r = sqr(x); break;
    case BLS381_FP12_INV: r = inv(x); break;
    case BLS381_FP12_FROBENIUS: r = frobenius(x); break;
    case BLS381_FP12_CYCLOTOMIC_SQR: r = cyclotomic_square(x); break;
    case BLS381_FP12_CONJ: r = fit<1, 2>(conj(x)); break;
    case BLS381_FP12_CYCLOTOMIC_EXP: r = cyclotomic_exp(x); break;
    case BLS381_FP12_MUL_BY_014: case BLS381_FP12_MUL_BY_014_KARATSUBA: {      // sparse factor (c0, c1, c4) = (b.c0.c0, b.c0.c1, b.c1.c1)
      const Fp12R y = b ? fp12_load(b + 72 * i, form) : x;
      r = (op == BLS381_FP12_MUL_BY_014) ? mul_by_014_lazy(x, y.c0.c0, y.c0.c1, y.c1.c1) : mul_by_014(x, y.c0.c0, y.c0.c1, y.c1.c1);
      break;
    }
    default: break;
  }
  fp12_store(out + 72 * i, r, form);
}

__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_line_ref_opt(const uint64_t* __restrict__ q1, const uint64_t* __restrict__ q2, const uint64_t* __restrict__ p,
               uint64_t* __restrict__ out, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  const G2Jac a = g2_jac_load(q1 + 36 * i, form), b = g2_jac_load(q2 + 36 * i, form);
  const G1Jac pt = g1_jac_load(p + 18 * i, form);
  const LineND l = line_chord(a, b, pt);
  fp2_store(out + 24 * i, l.n, form);
  fp2_store(out + 24 * i + 12, l.d, form);
}

// One G2 point per lane (pair): 68 x (a, b, c) Fp2 triples, 19 584 bytes per point.
__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_g2_prepare(const uint64_t* __restrict__ g2, uint64_t* __restrict__ out, uint8_t* __restrict__ flags, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  bool inf;
  G2Aff q = g2_aff_load(g2 + 24 * i, form, inf);
  if (inf) {   // the reference substitutes the generator and records the flag (src/miller_loop_target.rs:58-60)
    q.x = Fp2R(fp2_const(BLS_G2_GEN_X_C0, BLS_G2_GEN_X_C1));
    q.y = Fp2R(fp2_const(BLS_G2_GEN_Y_C0, BLS_G2_GEN_Y_C1));
  }
  uint64_t* o = out + (size_t)68 * 36 * i;
  g2_prepare(q, [&](int idx, const LineCoeffs& l) {
    fp2_store(o + 36 * idx, l.a, form);
    fp2_store(o + 36 * idx + 12, l.b, form);
    fp2_store(o + 36 * idx + 24, l.c, form);
  });
  if (flags && lane_part() == 0) flags[i] = inf ? FLAG_INFINITY : 0;
}

// Fp2 surface (T2: src/fields_as_trees/fq2_target_tree.rs:54-142).  a, b, out: n x 2 x 6 u64
__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_fp2_op(int op, const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint64_t* __restrict__ out, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  const Fp2R x = fp2_load(a + 12 * i, form);
  const Fp2R y = b ? fp2_load(b + 12 * i, form) : x;
  switch (op) {
    case BLS381_FP2_ADD: fp2_store(out + 12 * i, add(x, y), form); break;
    case BLS381_FP2_SUB: fp2_store(out + 12 * i, sub(x, y), form); break;
    case BLS381_FP2_MUL: fp2_store(out + 12 * i, mul(x, y), form); break;
    case BLS381_FP2_SQR: fp2_store(out + 12 * i, sqr(x), form); break;
    case BLS381_FP2_INV: fp2_store(out + 12 * i, inv(x), form); break;
    case BLS381_FP2_NEG: fp2_store(out + 12 * i, neg(x), form); break;
    case BLS381_FP2_CONJ: fp2_store(out + 12 * i, conj(x), form); break;
    case BLS381_FP2_MUL_BY_NONRESIDUE: fp2_store(out + 12 * i, mul_xi(x), form); break;
    default: break;
  }
}

BLS_HD Fp6R fp6_load(const uint64_t* w, int form) { return Fp6R(fp2_load(w, form), fp2_load(w + 12, form), fp2_load(w + 24, form)); }
template <int L, int V>
BLS_HD void fp6_store(uint64_t* w, const Fp6<L, V>& a, int form) { fp2_store(w, a.c0, form); fp2_store(w + 12, a.c1, form); fp2_store(w + 24, a.c2, form); }

// Fp6 surface (T3: src/fields_as_trees/fq6_target_tree.rs:59-293).  a, out: n x 3 x 2 x 6 u64; b: another Fp6, or for the
// sparse products the Fp2 coefficients (c0, c1) [MUL_BY_01] / (c1) [MUL_BY_1] packed at the front of an Fp6-sized slot
__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_fp6_op(int op, const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint64_t* __restrict__ out, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  const Fp6R x = fp6_load(a + 36 * i, form);
  const Fp6R y = b ? fp6_load(b + 36 * i, form) : x;
  switch (op) {
    case BLS381_FP6_ADD: fp6_store(out + 36 * i, add(x, y), form); break;
    case BLS381_FP6_SUB: fp6_store(out + 36 * i, sub(x, y), form); break;
    case BLS381_FP6_MUL: fp6_store(out + 36 * i, mul(x, y), form); break;
    case BLS381_FP6_SQR: fp6_store(out + 36 * i, sqr(x), form); break;
    case BLS381_FP6_INV: fp6_store(out + 36 * i, inv(x), form); break;
    case BLS381_FP6_NEG: fp6_store(out + 36 * i, neg(x), form); break;
    case BLS381_FP6_MUL_BY_NONRESIDUE: fp6_store(out + 36 * i, mul_v(x), form); break;
    case BLS381_FP6_FROBENIUS: fp6_store(out + 36 * i, frobenius(x), form); break;
    case BLS381_FP6_MUL_BY_01: fp6_store(out + 36 * i, mul_by_01(x, y.c0, y.c1), form); break;
    case BLS381_FP6_MUL_BY_1: fp6_store(out + 36 * i, mul_by_1(x, y.c0), form); break;
    default: break;
  }
}

struct PowExp { uint64_t w[8]; int nwords; };
// pow_fq (src/fields/helpers.rs:176-195): a^exp; `has_neg_digit` says the NAF of exp contains a -1, where the
// reference divides by a and panics for a == 0.
__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_fp_pow(const uint64_t* __restrict__ a, PowExp e, int has_neg_digit, uint64_t* __restrict__ out, uint8_t* __restrict__ flags, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  const FpR x = fp_load(a + 6 * i, form);
  uint8_t fl = 0;
  if (has_neg_digit && is_zero(x)) fl = 4;       // BLS381_FLAG_NO_INVERSE
  fp_store(out + 6 * i, fp_pow_words(x, e.w, e.nwords), form);
  if (flags && lane_part() == 0) flags[i] = fl;
}

__global__ void k_fill_one(uint64_t* __restrict__ out, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  fp12_store(out + 72 * i, Fp12R(fp12_one()), form);
}

// Point surface: one Jacobian point (pair) per lane pair; W = u64 words per point (18 on G1, 36 on G2).  On G1 both
// lanes of a pair compute the same Fp values (like k_fp_op).
template <class F, int W>
__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_point_op(int op, const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint64_t* __restrict__ out,
           uint8_t* __restrict__ flags, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  Jac<F> x, y, r;
  jac_load(x, a + (size_t)W * i, form);
  if (b) jac_load(y, b + (size_t)W * i, form); else y = x;
  const uint8_t fl = point_op(op, x, y, r);
  if (out) jac_store(out + (size_t)W * i, r, form);
  if (flags && lane_part() == 0) flags[i] = fl;
}

// Simultaneous inversion of n internal-form Fp values in place (raw limbs, RAW_FP_STRIDE dwords apart): one lane per
// chunk of FP_BATCH_INV_CHUNK values (Montgomery's trick in registers), so the batch pays one Fermat inversion per
// chunk instead of one per value.  Serves the split final exponentiation (k_fexp.hip).
__global__ void __launch_bounds__(64)
k_fp_batch_inv(uint32_t* __restrict__ raw, size_t n) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t first = t * FP_BATCH_INV_CHUNK;
  if (first >= n) return;
  FpR x[FP_BATCH_INV_CHUNK];
#pragma unroll
  for (int j = 0; j < FP_BATCH_INV_CHUNK; ++j)
    x[j] = (first + j < n) ? raw_fp_load(raw + (first + j) * RAW_FP_STRIDE) : FpR(fp_one());
  fp_batch_inv(x);
#pragma unroll
  for (int j = 0; j < FP_BATCH_INV_CHUNK; ++j)
    if (first + j < n) raw_fp_store(raw + (first + j) * RAW_FP_STRIDE, x[j]);
}

namespace bls_launch {
void point_op(int group, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s) {
  if (group == 1) hipLaunchKernelGGL((k_point_op<FpR, 18>), dim3(grid_for(n, block)), dim3(block), 0, s, op, a, b, out, flags, n, form);
  else hipLaunchKernelGGL((k_point_op<Fp2R, 36>), dim3(grid_for(n, block)), dim3(block), 0, s, op, a, b, out, flags, n, form);
}
void fp_batch_inv_raw(uint32_t* raw, size_t n, hipStream_t s) {
  const size_t chunks = (n + FP_BATCH_INV_CHUNK - 1) / FP_BATCH_INV_CHUNK;
  hipLaunchKernelGGL(k_fp_batch_inv, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, s, raw, n);
}
bool multi(size_t k, bool hom, const uint64_t* g1, const uint64_t* g2, size_t stride, uint64_t* out, uint8_t* flags, bool flags_or,
           size_t n, int form, int block, hipStream_t s) {
  switch (k) {
    case 1: multi_k1(hom, g1, g2, stride, out, flags, flags_or, n, form, block, s); return true;
    case 2: multi_k2(hom, g1, g2, stride, out, flags, flags_or, n, form, block, s); return true;
    case 3: multi_k3(hom, g1, g2, stride, out, flags, flags_or, n, form, block, s); return true;
    default: return false;
  }
}
void fp_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int form, int block, hipStream_t s) {
  hipLaunchKernelGGL(k_fp_op, dim3(grid_for(n, block)), dim3(block), 0, s, op, a, b, out, n, form);
}
void fp2_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int form, int block, hipStream_t s) {
  hipLaunchKernelGGL(k_fp2_op, dim3(grid_for(n, block)), dim3(block), 0, s, op, a, b, out, n, form);
}
void fp6_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int form, int block, hipStream_t s) {
  hipLaunchKernelGGL(k_fp6_op, dim3(grid_for(n, block)), dim3(block), 0, s, op, a, b, out, n, form);
}
void fp12_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int form, int block, hipStream_t s) {
  hipLaunchKernelGGL(k_fp12_op, dim3(grid_for(n, block)), dim3(block), 0, s, op, a, b, out, n, form);
}
void line_ref_opt(const uint64_t* q1, const uint64_t* q2, const uint64_t* p, uint64_t* out, size_t n, int form, int block, hipStream_t s) {
  hipLaunchKernelGGL(k_line_ref_opt, dim3(grid_for(n, block)), dim3(block), 0, s, q1, q2, p, out, n, form);
}
void g2_prepare(const uint64_t* g2, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s) {
  hipLaunchKernelGGL(k_g2_prepare, dim3(grid_for(n, block)), dim3(block), 0, s, g2, out, flags, n, form);
}
void fp_pow(const uint64_t* a, const uint64_t* exp, int nwords, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s) {
  PowExp e; for (int i = 0; i < 8; ++i) e.w[i] = i < nwords ? exp[i] : 0; e.nwords = nwords;
  int neg = 0; for (int i = 0; i < nwords; ++i) { if (e.w[i] & (e.w[i] >> 1)) neg = 1; if (i + 1 < nwords && (e.w[i] >> 63) && (e.w[i + 1] & 1)) neg = 1; }
  hipLaunchKernelGGL(k_fp_pow, dim3(grid_for(n, block)), dim3(block), 0, s, a, e, neg, out, flags, n, form);
}
void fill_one(uint64_t* out, size_t n, int form, hipStream_t s) {
  hipLaunchKernelGGL(k_fill_one, dim3(grid_for(n, 256)), dim3(256), 0, s, out, n, form);
}
}  // namespace bls_launch
