// Kernel translation unit of libbls381_hip.so (see bls381_hip.hip for the C ABI).  The library is split into
// several TUs only to compile them in parallel; each exposes plain host launch functions (launch.h).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "../../include/bls381_hip.h"
#include "pairing.hip.h"
#include "launch.h"
using namespace bls;
#ifndef BLS_MAX_BLOCK
#define BLS_MAX_BLOCK 256
#endif
static inline unsigned grid_for(size_t n, int block) { return (unsigned)((n * bls::LANES_PER_ELEM + block - 1) / block); }

template <bool GENERIC>
__global__ void __launch_bounds__(BLS_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_miller_ref_opt(const uint64_t* __restrict__ g1, const uint64_t* __restrict__ g2, uint64_t* __restrict__ out,
                 uint8_t* __restrict__ flags, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  const G1Jac p = g1_jac_load(g1 + 18 * i, form);
  const G2Jac q = g2_jac_load(g2 + 36 * i, form);
  uint8_t fl = 0;
  const Fp12R f = miller_ref_opt<GENERIC>(p, q, fl);
  fp12_store(out + 72 * i, f, form);
  if (flags && lane_part() == 0) flags[i] = fl;
}

namespace bls_launch {
void ref_opt(bool fp2_shortcut, const uint64_t* g1, const uint64_t* g2, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s) {
  static const bool narrow = getenv("BLS381_NARROW_REF_OPT") != nullptr;      // A/B switch
  if (!fp2_shortcut && !narrow) { ref_opt_wide(g1, g2, out, flags, n, form, block, s); return; }
  if (fp2_shortcut) hipLaunchKernelGGL(k_miller_ref_opt<false>, dim3(grid_for(n, block)), dim3(block), 0, s, g1, g2, out, flags, n, form);
  else hipLaunchKernelGGL(k_miller_ref_opt<true>, dim3(grid_for(n, block)), dim3(block), 0, s, g1, g2, out, flags, n, form);
}
}  // namespace bls_launch
