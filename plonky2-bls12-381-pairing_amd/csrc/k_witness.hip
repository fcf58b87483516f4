// Kernel translation unit: batched witness-generator backend (witness.hip.h).
#include <hip/hip_runtime.h>
#include "../../include/bls381_hip.h"
#include "witness.hip.h"
#include "launch.h"
using namespace bls;
#ifndef BLS_MAX_BLOCK
#define BLS_MAX_BLOCK 256
#endif
static inline unsigned grid_for(size_t n, int block) { return (unsigned)((n * bls::LANES_PER_ELEM + block - 1) / block); }

// op: 0 fq12 inverse (MyFq12 order), 1 fq6 inverse (MyFq6 order), 2 fq2 inverse
__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_witness_inverse(int op, const uint64_t* __restrict__ x, uint64_t* __restrict__ out, uint8_t* __restrict__ flags, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  uint8_t fl = 0;
  if (op == 0) {
    const Fp12R a = fp12_load_my(x + 72 * i);
    if (is_zero(a)) fl |= WFLAG_NO_INVERSE;                 // `x.inverse().unwrap()` panics (fq12_target.rs:362)
    fp12_store_my(out + 72 * i, inv(a));
  } else if (op == 1) {
    const Fp6R a = fp6_load_my(x + 36 * i);
    if (is_zero(a)) fl |= WFLAG_NO_INVERSE;                 // fq6_target.rs:406
    fp6_store_my(out + 36 * i, inv(a));
  } else {
    const Fp2R a = fp2_load(x + 12 * i, IO_CANONICAL);
    fp2_store(out + 12 * i, inv(a), IO_CANONICAL);          // None -> zero (fq2_target.rs:340-343)
  }
  if (flags && lane_part() == 0) flags[i] = fl;
}

// op: 0 fq sqrt, 1 fq2 sqrt
__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_witness_sqrt(int op, const uint64_t* __restrict__ x, const uint8_t* __restrict__ sgn, uint64_t* __restrict__ out,
               uint8_t* __restrict__ flags, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  uint8_t fl = 0;
  const bool want = (sgn[i] & 1) != 0;                      // `sgn_val.to_canonical_u64() % 2 == 1`
  if (op == 0) {
    const FpR a = fp_load(x + 6 * i, IO_CANONICAL);
    const FpR r = witness_fq_sqrt(a, want, fl);
    fp_store(out + 6 * i, r, IO_CANONICAL);
  } else {
    const Fp2R a = fp2_load(x + 12 * i, IO_CANONICAL);
    const Fp2R r = witness_fq2_sqrt(a, want, fl);
    fp2_store(out + 12 * i, r, IO_CANONICAL);
  }
  if (flags && lane_part() == 0) flags[i] = fl;
}

// sign byte: bit 0 = requested sgn0 of y, bit 1 = the point at infinity (x ignored, all-zero output, BLS381_FLAG_INFINITY)
__global__ void __launch_bounds__(BLS_MAX_BLOCK)
k_decompress(int group, const uint64_t* __restrict__ x, const uint8_t* __restrict__ sgn, uint64_t* __restrict__ out,
             uint8_t* __restrict__ flags, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  uint8_t fl = 0;
  const bool want = (sgn[i] & 1) != 0, inf = (sgn[i] & 2) != 0;
  if (group == 1) {
    const FpR a = fp_load(x + 6 * i, form);
    const FpR y = decompress_y(a, want, fl);
    const bool bad = inf || (fl & WFLAG_NOT_SQUARE);
    fp_store(out + 12 * i, select(bad, FpR(fp_zero()), a), form);
    fp_store(out + 12 * i + 6, select(bad, FpR(fp_zero()), y), form);
  } else {
    const Fp2R a = fp2_load(x + 12 * i, form);
    const Fp2R y = decompress_y(a, want, fl);
    const bool bad = inf || (fl & WFLAG_NOT_SQUARE);
    fp2_store(out + 24 * i, select(bad, Fp2R(fp2_zero()), a), form);
    fp2_store(out + 24 * i + 12, select(bad, Fp2R(fp2_zero()), y), form);
  }
  if (inf) fl = BLS381_FLAG_INFINITY;
  if (flags && lane_part() == 0) flags[i] = fl;
}

namespace bls_launch {
void decompress(int group, const uint64_t* x, const uint8_t* sgn, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s) {
  hipLaunchKernelGGL(k_decompress, dim3(grid_for(n, block)), dim3(block), 0, s, group, x, sgn, out, flags, n, form);
}
void witness_inverse(int op, const uint64_t* x, uint64_t* out, uint8_t* flags, size_t n, int block, hipStream_t s) {
  hipLaunchKernelGGL(k_witness_inverse, dim3(grid_for(n, block)), dim3(block), 0, s, op, x, out, flags, n);
}
void witness_sqrt(int op, const uint64_t* x, const uint8_t* sgn, uint64_t* out, uint8_t* flags, size_t n, int block, hipStream_t s) {
  hipLaunchKernelGGL(k_witness_sqrt, dim3(grid_for(n, block)), dim3(block), 0, s, op, x, sgn, out, flags, n);
}
}  // namespace bls_launch
