// Kernel translation unit of libbls381_hip.so: the final-exponentiation stage that has NO squaring chain (stage 6: decompress six
// powers, seven Fq12 products), compiled for ONE wave per SIMD.  In k_fexp.hip every routine gets 256 registers (two waves per
// SIMD pay off in the squaring chains); a stage that is only Fq12 products then passes every operand and result through scratch.
// Here the same source gets 512 registers: no private segment, the products stay in registers.
#include <hip/hip_runtime.h>
#define BLS_COLD_INLINE 1
#include "../../include/bls381_hip.h"
#include "pairing.hip.h"
#include "fexp_stages.hip.h"
#include "launch.h"
using namespace bls;
#ifndef BLS_MAX_BLOCK
#define BLS_MAX_BLOCK 256
#endif

__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_fexp_stage6_wide(uint64_t* out, uint8_t* flags, uint32_t* __restrict__ scratch, uint32_t* __restrict__ norm_raw, size_t n, size_t lanes, int form) {
  const size_t gl = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = gl / LANES_PER_ELEM;
  if (i >= n) return;
  const FexpScratch sc{fx_lane_scratch(scratch, lanes, gl), norm_raw, i};
  fp12_store(out + 72 * i, fexp_stage6(sc), form);
  if (flags && lane_part() == 0) flags[i] = 0;
}

// The glue of stages 1..5 on its own (split launch): ends by parking the compressed input of the next squaring chain, which
// k_fexp_chain (k_fexp.hip, two waves per SIMD) then walks.  One kernel for stages 2..5: their common part (decompress six
// powers, five Fq12 products) is most of the code, the stage is a uniform switch.
__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_fexp_glue1_wide(const uint64_t* __restrict__ in, uint32_t* __restrict__ scratch, uint32_t* __restrict__ norm_raw, size_t n, size_t lanes, int form) {
  const size_t gl = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = gl / LANES_PER_ELEM;
  if (i >= n) return;
  const FexpScratch sc{fx_lane_scratch(scratch, lanes, gl), norm_raw, i};
  fx_putx(sc.ls, compress(fexp_glue1(fp12_load(in + 72 * i, form), sc)));
}
__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_fexp_glue_wide(int stage, uint32_t* __restrict__ scratch, uint32_t* __restrict__ norm_raw, size_t n, size_t lanes) {
  const size_t gl = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = gl / LANES_PER_ELEM;
  if (i >= n) return;
  const FexpScratch sc{fx_lane_scratch(scratch, lanes, gl), norm_raw, i};
  const Fp12R e = fexp_finish_exp(sc);                                   // E(previous value), fexp_stages.hip.h
  Fp12R x;
  if (stage == 5) {
    const Fp12R m = fx_get12(sc.ls, FX_M);
    const Fp12R m3 = mul(cyclotomic_square(m), m);
    const Fp12R c = fx_get12(sc.ls, FX_A);
    const Fp12R w = mul(frobenius(frobenius(c)), fit<1, 2>(conj(c)));
    fx_put12(sc.ls, FX_W, mul(w, m3));
    x = e;
  } else {
    const Fp12R t = (stage == 2) ? fit<1, 2>(conj(fx_get12(sc.ls, FX_M)))
                  : (stage == 3) ? fit<1, 2>(conj(fx_get12(sc.ls, FX_A)))
                                 : frobenius(fx_get12(sc.ls, FX_A));
    x = mul(e, t);
    fx_put12(sc.ls, FX_A, x);
  }
  fx_putx(sc.ls, compress(x));
}

namespace bls_launch {
void fexp_stage6_wide(uint64_t* out, uint8_t* flags, uint32_t* scratch, uint32_t* norm_raw, size_t n, size_t lanes, int form, int block, hipStream_t s) {
  hipLaunchKernelGGL(k_fexp_stage6_wide, dim3((unsigned)(lanes / block)), dim3(block), 0, s, out, flags, scratch, norm_raw, n, lanes, form);
}
void fexp_glue_wide(int stage, const uint64_t* in, uint32_t* scratch, uint32_t* norm_raw, size_t n, size_t lanes, int form, int block, hipStream_t s) {
  if (stage == 1) hipLaunchKernelGGL(k_fexp_glue1_wide, dim3((unsigned)(lanes / block)), dim3(block), 0, s, in, scratch, norm_raw, n, lanes, form);
  else hipLaunchKernelGGL(k_fexp_glue_wide, dim3((unsigned)(lanes / block)), dim3(block), 0, s, stage, scratch, norm_raw, n, lanes);
}
}  // namespace bls_launch
