// Kernel translation unit of libbls381_hip.so: Mode A with GENERIC Fq12 accumulators (the reference multiplies full Fq12 values,
// src/miller_loop_native_optimized.rs:91-97) compiled for ONE wave per SIMD with the Fq12 products inlined.  In k_ref_opt.hip the
// same source gets 256 registers and every product passes its operands and result through call frames in scratch; here the four
// products of an iteration keep their operands in registers (the same trade as the glue kernels of the final exponentiation,
// DESIGN 3.5).
#include <hip/hip_runtime.h>
#define BLS_COLD_INLINE 1
#include "../../include/bls381_hip.h"
#include "pairing.hip.h"
#include "launch.h"
using namespace bls;
#ifndef BLS_MAX_BLOCK
#define BLS_MAX_BLOCK 256
#endif

__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_miller_ref_opt_wide(const uint64_t* __restrict__ g1, const uint64_t* __restrict__ g2, uint64_t* __restrict__ out,
                      uint8_t* __restrict__ flags, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  const G1Jac p = g1_jac_load(g1 + 18 * i, form);
  const G2Jac q = g2_jac_load(g2 + 36 * i, form);
  uint8_t fl = 0;
  const Fp12R f = miller_ref_opt<true>(p, q, fl);
  fp12_store(out + 72 * i, f, form);
  if (flags && lane_part() == 0) flags[i] = fl;
}

namespace bls_launch {
void ref_opt_wide(const uint64_t* g1, const uint64_t* g2, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s) {
  hipLaunchKernelGGL(k_miller_ref_opt_wide, dim3((unsigned)((n * bls::LANES_PER_ELEM + block - 1) / block)), dim3(block), 0, s, g1, g2, out, flags, n, form);
}
}  // namespace bls_launch
