// Kernel translation unit of libbls381_hip.so: the final-exponentiation stage that has NO squaring chain (stage 6: decompress six
// powers, seven Fq12 products), compiled for ONE wave per SIMD.  In k_fexp.hip every routine gets 256 registers (two waves per
// SIMD pay off in the squaring chains); a stage that is only Fq12 products then passes every operand and result through scratch.
// Here the same source gets 512 registers: no private segment, the products stay in registers.
#include <hip/hip_runtime.h>
#define BLS_COLD_INLINE 1
#include "../../include/bls381_hip.h"
#include "pairing.cuh"
#include "fexp_stages.cuh"
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
  const FexpScratch sc{LaneScratch{scratch, lanes, gl}, norm_raw, i};
  fp12_store(out + 72 * i, fexp_stage6(sc), form);
  if (flags && lane_part() == 0) flags[i] = 0;
}

namespace bls_launch {
void fexp_stage6_wide(uint64_t* out, uint8_t* flags, uint32_t* scratch, uint32_t* norm_raw, size_t n, size_t lanes, int form, int block, hipStream_t s) {
  hipLaunchKernelGGL(k_fexp_stage6_wide, dim3((unsigned)(lanes / block)), dim3(block), 0, s, out, flags, scratch, norm_raw, n, lanes, form);
}
}  // namespace bls_launch
