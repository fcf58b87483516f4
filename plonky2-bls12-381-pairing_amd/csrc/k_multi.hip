// Kernel translation unit of libbls381_hip.so (see bls381_hip.hip for the C ABI).  The library is split into
// several TUs only to compile them in parallel; each exposes plain host launch functions (launch.h).
#include <hip/hip_runtime.h>
#include "../../include/bls381_hip.h"
// one wave per SIMD: every instruction costs one issue slot, so the sparse line multiplication runs as lazy sums (tower.hip.h)
#ifndef BLS_LAZY_014
#define BLS_LAZY_014 1
#endif
#include "pairing.hip.h"
#include "launch.h"
using namespace bls;
#ifndef BLS_MAX_BLOCK
#define BLS_MAX_BLOCK 256
#endif
static inline unsigned grid_for(size_t n, int block) { return (unsigned)((n * bls::LANES_PER_ELEM + block - 1) / block); }

#ifndef BLS_K
#error "compile with -DBLS_K=1..3"
#endif
#define BLS_CAT2(a, b) a##b
#define BLS_CAT(a, b) BLS_CAT2(a, b)
// HOM: homogeneous-projective curve steps (pairing.hip.h) -- for launches that are followed by the final exponentiation
#if BLS_K == 1
template <int K, bool HOM>
__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_multi_miller(const uint64_t* __restrict__ g1, const uint64_t* __restrict__ g2, size_t stride, uint64_t* __restrict__ out,
               uint8_t* __restrict__ flags, int flags_or, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  G1Aff p[K]; G2Aff q[K]; bool skip[K];
  uint8_t fl = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    bool i1, i2;
    p[j] = g1_aff_load(g1 + 12 * (stride * i + j), form, i1);      // stride = pairs per element in the caller's arrays (>= K)
    q[j] = g2_aff_load(g2 + 24 * (stride * i + j), form, i2);
    skip[j] = i1 || i2;
    if (skip[j]) fl |= FLAG_INFINITY;
  }
  const Fp12R f = miller_ate<K, HOM>(p, q, skip);
  fp12_store(out + 72 * i, f, form);
  if (flags && lane_part() == 0) flags[i] = flags_or ? (uint8_t)(flags[i] | fl) : fl;
}
constexpr int MULTI_BLOCK = 0;            // any of the context's block sizes
#else
// Two and three pairs: the current pair's R in registers, the others' in LDS (pairing.hip.h miller_ate_lds).  One wave per
// workgroup -- the LDS layout is compiled for it -- and no barrier anywhere: every lane only touches its own LDS column.
constexpr int MULTI_BLOCK = 64;
template <int K, bool HOM>
__global__ void __launch_bounds__(MULTI_BLOCK)
k_multi_miller(const uint64_t* __restrict__ g1, const uint64_t* __restrict__ g2, size_t stride, uint64_t* __restrict__ out,
               uint8_t* __restrict__ flags, int flags_or, size_t n, int form) {
  __shared__ LdsQuad lds_mem[MULTI_BLOCK * (miller_lds_bytes_per_lane(K) / sizeof(LdsQuad))];
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  uint8_t fl = 0;
  const LaneLds<MULTI_BLOCK> lds = lane_lds<MULTI_BLOCK>(lds_mem, threadIdx.x);
  const Fp12R f = miller_ate_lds<K, HOM>(g1 + 12 * stride * i, g2 + 24 * stride * i, form, lds, fl);   // stride = pairs per element in the caller's arrays (>= K)
  fp12_store(out + 72 * i, f, form);
  if (flags && lane_part() == 0) flags[i] = flags_or ? (uint8_t)(flags[i] | fl) : fl;
}
#endif

namespace bls_launch {
// A full pairing is this Miller kernel (one wave per SIMD, all 512 registers per lane, 0-88 B of private segment for one pair: DESIGN 3.4 table) followed by a final
// exponentiation in place on `out` (k_fexp.hip: its own translation unit, compiled for two waves per SIMD, where it
// runs ~15 % faster than inside a fused kernel).  The intermediate Fq12 costs 2 x 576 B of HBM traffic per pairing.
void BLS_CAT(multi_k, BLS_K)(bool hom, const uint64_t* g1, const uint64_t* g2, size_t stride, uint64_t* out, uint8_t* flags, bool flags_or,
                             size_t n, int form, int block, hipStream_t s) {
  if (MULTI_BLOCK) block = MULTI_BLOCK;
  if (hom) hipLaunchKernelGGL((k_multi_miller<BLS_K, true>), dim3(grid_for(n, block)), dim3(block), 0, s, g1, g2, stride, out, flags, (int)flags_or, n, form);
  else hipLaunchKernelGGL((k_multi_miller<BLS_K, false>), dim3(grid_for(n, block)), dim3(block), 0, s, g1, g2, stride, out, flags, (int)flags_or, n, form);
}
}  // namespace bls_launch
