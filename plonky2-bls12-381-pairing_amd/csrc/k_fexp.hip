// Kernel translation unit of libbls381_hip.so: the final exponentiation, alone in its file ON PURPOSE.
// Register budgets propagate through the call graph of a translation unit: the out-of-line tower routines shared
// with a 512-register kernel would be compiled for 512 registers and drag this kernel back to one wave per SIMD.
// Here every kernel asks for two waves per SIMD, so the routines are compiled for 256 registers.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "../../include/bls381_hip.h"
#include "pairing.hip.h"
#include "fexp_stages.hip.h"
#include "launch.h"
using namespace bls;
#ifndef BLS_MAX_BLOCK
#define BLS_MAX_BLOCK 256
#endif
static inline unsigned grid_for(size_t n, int block) { return (unsigned)((n * bls::LANES_PER_ELEM + block - 1) / block); }

// Final exponentiation, in place allowed (in == out: each lane pair reads its element before writing it).
// Two waves per SIMD: the cyclotomic-squaring runs need ~250 registers, and with two waves the SIMD issues a MAD
// every ~4.2 cycles instead of ~5.6 (profiles/r01_v3: 44-46 ms vs 52 ms per 2^18 elements).
//
// k_final_exp is the self-contained form (one Fermat inversion per lane, ~9 % of the kernel).  The split form amortises
// that inversion over the batch:  k_fexp_norm writes the Fp norm under f^-1 for every element,  k_fp_batch_inv
// (k_misc.hip) inverts them sixteen per lane with Montgomery's trick,  k_final_exp_given finishes f^-1 from the supplied
// norm inverse and runs the rest.  Same field values, hence the same output bits.
__global__ void __launch_bounds__(BLS_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_final_exp(const uint64_t* in, uint64_t* out, uint8_t* flags, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  const Fp12R f = fp12_load(in + 72 * i, form);
  fp12_store(out + 72 * i, final_exponentiation(f), form);
  if (flags && lane_part() == 0) flags[i] = 0;
}

__global__ void __launch_bounds__(BLS_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_fexp_norm(const uint64_t* __restrict__ in, uint32_t* __restrict__ raw, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  const Fp12R f = fp12_load(in + 72 * i, form);
  const FpR nrm = fp12_inv_norm_call(f);                       // identical on both lanes of the pair
  if (lane_part() == 0) raw_fp_store(raw + i * RAW_FP_STRIDE, nrm);
}

__global__ void __launch_bounds__(BLS_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_final_exp_given(const uint64_t* in, uint64_t* out, uint8_t* flags, const uint32_t* __restrict__ raw, size_t n, int form) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES_PER_ELEM;
  if (i >= n) return;
  const Fp12R f = fp12_load(in + 72 * i, form);
  const FpR norm_inv = raw_fp_load(raw + i * RAW_FP_STRIDE);
  const Fp12R f_inv = fp12_inv_given_call(f, norm_inv);
  fp12_store(out + 72 * i, final_exponentiation_given_inverse(f, f_inv), form);
  if (flags && lane_part() == 0) flags[i] = 0;
}

// Staged form (fexp_stages.hip.h): Karabina-compressed squaring chains, one batch-wide inversion between stages.
// `lanes` = number of lanes of the launch (the lane-sliced scratch is laid out for exactly that many).
template <int STAGE>
__global__ void __launch_bounds__(BLS_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_fexp_stage(const uint64_t* in, uint64_t* out, uint8_t* flags, uint32_t* __restrict__ scratch, uint32_t* __restrict__ norm_raw,
             size_t n, size_t lanes, int form) {
  const size_t gl = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = gl / LANES_PER_ELEM;
  if (i >= n) return;
  const FexpScratch sc{fx_lane_scratch(scratch, lanes, gl), norm_raw, i};
  if constexpr (STAGE == 1) fexp_stage1(fp12_load(in + 72 * i, form), sc);
  else if constexpr (STAGE == 2) fexp_stage2(sc);
  else if constexpr (STAGE == 3) fexp_stage3(sc);
  else if constexpr (STAGE == 4) fexp_stage4(sc);
  else if constexpr (STAGE == 5) fexp_stage5(sc);
  else {
    fp12_store(out + 72 * i, fexp_stage6(sc), form);
    if (flags && lane_part() == 0) flags[i] = 0;
  }
}

// Norm pass of the staged form: like k_fexp_norm, and parks the intermediate values of the inversion for stage 1 (fexp_norm_pass)
__global__ void __launch_bounds__(BLS_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_fexp_norm_park(const uint64_t* __restrict__ in, uint32_t* __restrict__ scratch, uint32_t* __restrict__ raw, size_t n, size_t lanes, int form) {
  const size_t gl = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = gl / LANES_PER_ELEM;
  if (i >= n) return;
  const FpR nrm = fexp_norm_pass(fp12_load(in + 72 * i, form), fx_lane_scratch(scratch, lanes, gl));
  if (lane_part() == 0) raw_fp_store(raw + i * RAW_FP_STRIDE, nrm);
}

// The squaring chain alone (split launch, after k_fexp_glue*_wide parked its input): ~200 registers, no private segment.
__global__ void __launch_bounds__(BLS_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_fexp_chain(uint32_t* __restrict__ scratch, uint32_t* __restrict__ norm_raw, size_t n, size_t lanes) {
  const size_t gl = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = gl / LANES_PER_ELEM;
  if (i >= n) return;
  const FexpScratch sc{fx_lane_scratch(scratch, lanes, gl), norm_raw, i};
  fexp_chain(fx_getx(sc.ls), sc);
}

namespace bls_launch {
size_t fexp_stage_scratch_bytes(size_t n_chunk) {      // for a chunk of n_chunk elements, any supported block size
  const size_t lanes = (n_chunk * bls::LANES_PER_ELEM + BLS_MAX_BLOCK - 1) / BLS_MAX_BLOCK * BLS_MAX_BLOCK;
  return lanes * (size_t)FX_DWORDS_PER_LANE * sizeof(uint32_t);
}
// Staged final exponentiation over chunks of at most `chunk` elements; raw_scratch: chunk x RAW_FP_STRIDE dwords,
// stage_scratch: fexp_stage_scratch_bytes(chunk)
hipError_t final_exp_staged(const uint64_t* in, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s,
                            uint32_t* raw_scratch, uint32_t* stage_scratch, size_t chunk,
                            const hipEvent_t* wait_before, const hipEvent_t* record_after) {
  hipError_t err = hipSuccess;     // a failed wait would turn the pipelines' ordering into a race: reported, not swallowed
  for (size_t first = 0; first < n; first += chunk) {
    const size_t m = (n - first < chunk) ? n - first : chunk;
    const dim3 grid(grid_for(m, block)), blk(block);
    const size_t lanes = (size_t)grid.x * block;
    const uint64_t* ci = in + 72 * first;
    uint64_t* co = out + 72 * first;
    uint8_t* cf = flags ? flags + first : nullptr;
    // events tie the stage kernels of the first pass to another pipeline's (see bls381_hip.hip)
    auto note = [&](hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; };
    auto before = [&](int k) { if (first == 0 && wait_before && wait_before[k]) note(hipStreamWaitEvent(s, wait_before[k], 0)); };
    auto after = [&](int k) { if (first == 0 && record_after && record_after[k]) note(hipEventRecord(record_after[k], s)); };
    hipLaunchKernelGGL(k_fexp_norm_park, grid, blk, 0, s, ci, stage_scratch, raw_scratch, m, lanes, form);
    fp_batch_inv_raw(raw_scratch, m, s);
    // stages 1..5: glue + squaring chain, fused in one two-wave kernel or as a one-wave glue kernel plus the chain kernel
    static const bool fused = getenv("BLS381_FEXP_FUSED_STAGES") != nullptr;    // A/B switch (tools/collect_profile.sh)
    for (int st = 1; st <= 5; ++st) {
      before(st - 1);
      if (fused) {
        switch (st) {
          case 1: hipLaunchKernelGGL(k_fexp_stage<1>, grid, blk, 0, s, ci, co, cf, stage_scratch, raw_scratch, m, lanes, form); break;
          case 2: hipLaunchKernelGGL(k_fexp_stage<2>, grid, blk, 0, s, ci, co, cf, stage_scratch, raw_scratch, m, lanes, form); break;
          case 3: hipLaunchKernelGGL(k_fexp_stage<3>, grid, blk, 0, s, ci, co, cf, stage_scratch, raw_scratch, m, lanes, form); break;
          case 4: hipLaunchKernelGGL(k_fexp_stage<4>, grid, blk, 0, s, ci, co, cf, stage_scratch, raw_scratch, m, lanes, form); break;
          default: hipLaunchKernelGGL(k_fexp_stage<5>, grid, blk, 0, s, ci, co, cf, stage_scratch, raw_scratch, m, lanes, form); break;
        }
      } else {
        fexp_glue_wide(st, ci, stage_scratch, raw_scratch, m, lanes, form, block, s);
        hipLaunchKernelGGL(k_fexp_chain, grid, blk, 0, s, stage_scratch, raw_scratch, m, lanes);
      }
      after(st - 1);
      fp_batch_inv_raw(raw_scratch, m, s);
    }
    before(5);
    static const bool narrow6 = getenv("BLS381_NARROW_STAGE6") != nullptr;      // A/B switch (tools/collect_profile.sh)
    if (narrow6) hipLaunchKernelGGL(k_fexp_stage<6>, grid, blk, 0, s, ci, co, cf, stage_scratch, raw_scratch, m, lanes, form);
    else fexp_stage6_wide(co, cf, stage_scratch, raw_scratch, m, lanes, form, block, s);
    after(5);
  }
  return err;
}
// `raw_scratch`: n x RAW_FP_STRIDE dwords of device memory for the split form, or nullptr for the self-contained kernel
void final_exp(const uint64_t* in, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s, uint32_t* raw_scratch) {
  const dim3 grid(grid_for(n, block)), blk(block);
  if (!raw_scratch) {
    hipLaunchKernelGGL(k_final_exp, grid, blk, 0, s, in, out, flags, n, form);
    return;
  }
  hipLaunchKernelGGL(k_fexp_norm, grid, blk, 0, s, in, raw_scratch, n, form);
  fp_batch_inv_raw(raw_scratch, n, s);
  hipLaunchKernelGGL(k_final_exp_given, grid, blk, 0, s, in, out, flags, (const uint32_t*)raw_scratch, n, form);
}
}  // namespace bls_launch
