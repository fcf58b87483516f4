// Kernel translation unit: pairing products with FIXED (prepared) G2 points, e.g. the Groth16-verify shape (BASELINE config 4;
// SURVEY.md 7.1 step 8): per proof  e(P0, Q0) * e(P1, Q1) * e(P2, Q2)  where Q1, Q2 (the verifying key's gamma, delta)
// are the same for the whole batch.  Their 2 x 68 line-coefficient triples (bls381_g2_prepare_batch;
// `G2PreparedTarget`, src/miller_loop_target.rs:23-76) are converted once per workgroup into the internal limb
// form and staged in LDS (52 KB), so the Miller loop runs ONE doubling/addition step per iteration (for Q0) instead
// of three; every lane reads the shared coefficients as LDS broadcasts.
#include <hip/hip_runtime.h>
#include "../../include/bls381_hip.h"
// one wave per SIMD: every instruction costs one issue slot, so the sparse line multiplication runs as lazy sums (tower.hip.h)
#ifndef BLS_LAZY_014
#define BLS_LAZY_014 1
#endif
#include "pairing.hip.h"
#include "launch.h"
using namespace bls;

constexpr int G16_BLOCK = 256;                 // 4 waves share one LDS copy of the coefficients
constexpr int G16_FP_STRIDE = 16;              // dwords per Fp in LDS (14 used; 16 keeps ds_read_b128 aligned)
constexpr int G16_FP_PER_POINT = 68 * 3 * 2;   // 68 steps x (a, b, c) x 2 coefficients

BLS_HD FpR lds_fp(const uint32_t* p) {
  FpR r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = p[i];
  return r;
}

// NVAR in {0, 1} pairs with a per-element G2 point (full doubling/addition steps), NFIXED in {1, 2} pairs whose G2
// point is fixed for the batch (prepared coefficients in LDS).  Per element the G1 points come first for the
// variable pair, then for the fixed pairs.  <1, 2> is the Groth16 shape; <0, 1> is "many P against one Q".
// HOM: the variable pair walks its curve steps in homogeneous projective coordinates (pairing.hip.h); its lines then differ from
// Alg. 26/27's by Fp2 factors, which the final exponentiation that always follows this kernel kills (same rule and same option,
// BLS381_OPT_PAIRING_HOMOGENEOUS_MILLER, as for bls381_multi_pairing_batch).  The fixed pairs' lines are the prepared ones either way.
template <int NVAR, int NFIXED, bool HOM>
__global__ void __launch_bounds__(G16_BLOCK)
k_prepared_miller(const uint64_t* __restrict__ g1, const uint64_t* __restrict__ g2_var, const uint64_t* __restrict__ prepared,
                  uint64_t* __restrict__ out, uint8_t* __restrict__ flags, size_t n, int form) {
  constexpr int G16_N_FP = NFIXED * G16_FP_PER_POINT;
  constexpr int NP = NVAR + NFIXED;
  __shared__ __attribute__((aligned(16))) uint32_t coef[G16_N_FP * G16_FP_STRIDE];
  // stage: ABI form (6 x u64 per Fp) -> internal Montgomery limbs, once per workgroup
  for (int t = threadIdx.x; t < G16_N_FP; t += G16_BLOCK) {
    const FpR v = fp_load(prepared + 6 * t, form);
#pragma unroll
    for (int i = 0; i < NL; ++i) coef[t * G16_FP_STRIDE + i] = v.l[i];
  }
  __syncthreads();
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;                          // after the barrier: every thread took part in the staging

  G1Aff p[NP]; bool skip[NP]; uint8_t fl = 0;
#pragma unroll
  for (int j = 0; j < NP; ++j) { bool inf; p[j] = g1_aff_load(g1 + 12 * (NP * i + j), form, inf); skip[j] = inf; }
  G2Aff q0;
  if constexpr (NVAR == 1) {
    bool qinf;
    q0 = g2_aff_load(g2_var + 24 * i, form, qinf);
    skip[0] = skip[0] || qinf;
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) if (skip[j]) fl |= FLAG_INFINITY;

  // coefficient (pair j, step idx, member m in {a,b,c}) component `part` lives at Fp index ((j*68 + idx)*3 + m)*2 + part
  const int part = (LANES_PER_ELEM == 2) ? lane_part() : 0;
  auto fixed_line = [&](int j, int idx) {
    LineCoeffs l;
    const uint32_t* base = coef + (size_t)(((j * 68 + idx) * 3) * 2) * G16_FP_STRIDE;
#if BLS_LANES == 2
    l.a = Fp2R(lds_fp(base + (0 * 2 + part) * G16_FP_STRIDE));
    l.b = Fp2R(lds_fp(base + (1 * 2 + part) * G16_FP_STRIDE));
    l.c = Fp2R(lds_fp(base + (2 * 2 + part) * G16_FP_STRIDE));
#else
    l.a = Fp2R(lds_fp(base + 0 * G16_FP_STRIDE), lds_fp(base + 1 * G16_FP_STRIDE));
    l.b = Fp2R(lds_fp(base + 2 * G16_FP_STRIDE), lds_fp(base + 3 * G16_FP_STRIDE));
    l.c = Fp2R(lds_fp(base + 4 * G16_FP_STRIDE), lds_fp(base + 5 * G16_FP_STRIDE));
    (void)part;
#endif
    return l;
  };

  const Fp12R f = miller_prepared<NVAR, NFIXED, HOM>(p, q0, skip, fixed_line);      // pairing.hip.h: the loop itself
  fp12_store(out + 72 * i, f, form);
  if (flags && lane_part() == 0) flags[i] = fl;
}

namespace bls_launch {
bool prepared(int nvar, int nfixed, bool hom, const uint64_t* g1, const uint64_t* g2_var, const uint64_t* prep, uint64_t* out, uint8_t* flags,
              size_t n, int form, int block, hipStream_t s) {
  const dim3 grid((unsigned)((n * bls::LANES_PER_ELEM + G16_BLOCK - 1) / G16_BLOCK)), blk(G16_BLOCK);
  if (nvar == 1 && nfixed == 2 && hom) hipLaunchKernelGGL((k_prepared_miller<1, 2, true>), grid, blk, 0, s, g1, g2_var, prep, out, flags, n, form);
  else if (nvar == 1 && nfixed == 2) hipLaunchKernelGGL((k_prepared_miller<1, 2, false>), grid, blk, 0, s, g1, g2_var, prep, out, flags, n, form);
  else if (nvar == 0 && nfixed == 1) hipLaunchKernelGGL((k_prepared_miller<0, 1, false>), grid, blk, 0, s, g1, g2_var, prep, out, flags, n, form);
  else if (nvar == 1 && nfixed == 1 && hom) hipLaunchKernelGGL((k_prepared_miller<1, 1, true>), grid, blk, 0, s, g1, g2_var, prep, out, flags, n, form);
  else if (nvar == 1 && nfixed == 1) hipLaunchKernelGGL((k_prepared_miller<1, 1, false>), grid, blk, 0, s, g1, g2_var, prep, out, flags, n, form);
  else if (nvar == 0 && nfixed == 2) hipLaunchKernelGGL((k_prepared_miller<0, 2, false>), grid, blk, 0, s, g1, g2_var, prep, out, flags, n, form);
  else return false;
  return true;
}
}  // namespace bls_launch
