// Kernel translation unit of libbls381_hip.so: Mode A with Fq12 accumulators (the reference multiplies full Fq12 values,
// src/miller_loop_native_optimized.rs:91-97) compiled for ONE wave per SIMD with the Fq12 products inlined.  In k_ref_opt.hip the
// same source gets 256 registers and every product passes its operands and result through call frames in scratch; here the
// products keep their operands in registers (the same trade as the glue kernels of the final exponentiation), and
// of the two accumulators f_num / f_den only the one being multiplied is in registers -- the other waits in LDS and they change places
// after every half-iteration (pairing.hip.h miller_ref_opt_generic_lds).
#include <hip/hip_runtime.h>
#define BLS_COLD_INLINE 1
#include "../../include/bls381_hip.h"
#include "pairing.hip.h"
#include "launch.h"
using namespace bls;

constexpr int REF_OPT_BLOCK = 64;         // one wave per workgroup: 6 LDS slots x 64 B per lane = 24 KB, four workgroups per CU

__global__ void __launch_bounds__(REF_OPT_BLOCK)
k_miller_ref_opt_wide(const uint64_t* __restrict__ g1, const uint64_t* __restrict__ g2, uint64_t* __restrict__ out,
                      uint8_t* __restrict__ flags, size_t n, int form) {
  __shared__ LdsQuad lds_mem[REF_OPT_BLOCK * (ref_opt_lds_bytes_per_lane() / sizeof(LdsQuad))];
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;                      // no barrier anywhere: every lane only touches its own LDS column
  const G1Jac p = g1_jac_load(g1 + 18 * i, form);
  uint8_t fl = 0;
  const LaneLds<REF_OPT_BLOCK> lds = lane_lds<REF_OPT_BLOCK>(lds_mem, threadIdx.x);
  const Fp12R f = miller_ref_opt_generic_lds(p, g2 + 36 * i, form, lds, fl);
  fp12_store(out + 72 * i, f, form);
  if (flags && lane_part() == 0) flags[i] = fl;
}

namespace bls_launch {
// the workgroup size is fixed (the LDS layout is compiled for it); `block` is accepted for symmetry with the other launchers
void ref_opt_wide(const uint64_t* g1, const uint64_t* g2, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s) {
  (void)block;
  hipLaunchKernelGGL(k_miller_ref_opt_wide, dim3((unsigned)((n * bls::LANES_PER_ELEM + REF_OPT_BLOCK - 1) / REF_OPT_BLOCK)), dim3(REF_OPT_BLOCK), 0, s,
                     g1, g2, out, flags, n, form);
}
}  // namespace bls_launch
