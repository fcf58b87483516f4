// Host-side launch functions implemented by the kernel translation units (k_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
namespace bls_launch {
// Mode A with generic Fq12 accumulators, one wave per SIMD, products inlined (k_ref_opt_wide.hip)
void ref_opt_wide(const uint64_t* g1, const uint64_t* g2, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s);
void ref_opt(bool fp2_shortcut, const uint64_t* g1, const uint64_t* g2, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s);
// k in 1..3 pairs per launch, read from arrays that hold `stride` >= k pairs per element (g1 / g2 point at the first pair of
// element 0 to use); returns false for an unsupported k.  flags_or: OR into flags[] instead of overwriting it.
// Miller loops only; a full pairing = this, then one of the final-exponentiation launches in place on `out`
// hom: homogeneous-projective curve steps (only for results that go through the final exponentiation)
bool multi(size_t k, bool hom, const uint64_t* g1, const uint64_t* g2, size_t stride, uint64_t* out, uint8_t* flags, bool flags_or,
           size_t n, int form, int block, hipStream_t s);
void multi_k1(bool hom, const uint64_t*, const uint64_t*, size_t, uint64_t*, uint8_t*, bool, size_t, int, int, hipStream_t);
void multi_k2(bool hom, const uint64_t*, const uint64_t*, size_t, uint64_t*, uint8_t*, bool, size_t, int, int, hipStream_t);
void multi_k3(bool hom, const uint64_t*, const uint64_t*, size_t, uint64_t*, uint8_t*, bool, size_t, int, int, hipStream_t);
// raw_scratch: n x 16 dwords for the split form (batched inversion), or nullptr for the self-contained kernel
void final_exp(const uint64_t* in, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s, uint32_t* raw_scratch);
void fp_batch_inv_raw(uint32_t* raw, size_t n, hipStream_t s);
// staged form (Karabina-compressed squaring chains, csrc/fexp_stages.hip.h): processes the batch in chunks of `chunk` elements;
// raw_scratch: chunk x 16 dwords, stage_scratch: fexp_stage_scratch_bytes(chunk) bytes; wait_before[k] / record_after[k]
// (arrays of six, entries or the arrays may be null) are waited for before / recorded behind stage kernel k+1 of the first pass
size_t fexp_stage_scratch_bytes(size_t n_chunk);
// returns the first HIP error of the event waits / records that order the stage kernels against other pipelines (hipSuccess otherwise)
hipError_t final_exp_staged(const uint64_t* in, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s,
                      uint32_t* raw_scratch, uint32_t* stage_scratch, size_t chunk,
                      const hipEvent_t* wait_before, const hipEvent_t* record_after);
// stage 6 of the staged form compiled for one wave per SIMD (k_fexp_wide.hip); same arguments as the stage kernels
// glue of stage 1..5 compiled for one wave per SIMD (split launch: followed by k_fexp_chain); `in` is read by stage 1 only
void fexp_glue_wide(int stage, const uint64_t* in, uint32_t* scratch, uint32_t* norm_raw, size_t n, size_t lanes, int form, int block, hipStream_t s);
void fexp_stage6_wide(uint64_t* out, uint8_t* flags, uint32_t* scratch, uint32_t* norm_raw, size_t n, size_t lanes, int form, int block, hipStream_t s);
void fp_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int form, int block, hipStream_t s);
void fp2_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int form, int block, hipStream_t s);
void fp6_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int form, int block, hipStream_t s);
void fp12_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int form, int block, hipStream_t s);
void line_ref_opt(const uint64_t* q1, const uint64_t* q2, const uint64_t* p, uint64_t* out, size_t n, int form, int block, hipStream_t s);
// group 1 = G1 (n x 3 x 6 u64), 2 = G2 (n x 3 x 2 x 6)
void point_op(int group, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s);
void fill_one(uint64_t* out, size_t n, int form, hipStream_t s);
// Miller kernel with LDS-staged fixed-G2 coefficients (no final exponentiation); hom: homogeneous curve steps for the variable pair
bool prepared(int nvar, int nfixed, bool hom, const uint64_t* g1, const uint64_t* g2_var, const uint64_t* prep, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s);
void fp_pow(const uint64_t* a, const uint64_t* exp, int nwords, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s);
void g2_prepare(const uint64_t* g2, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s);
void witness_inverse(int op, const uint64_t* x, uint64_t* out, uint8_t* flags, size_t n, int block, hipStream_t s);
// group 1: x n x 6 u64 -> out n x 12; group 2: x n x 12 -> out n x 24; sgn: n bytes (bit 0 sign of y, bit 1 infinity)
void decompress(int group, const uint64_t* x, const uint8_t* sgn, uint64_t* out, uint8_t* flags, size_t n, int form, int block, hipStream_t s);
void witness_sqrt(int op, const uint64_t* x, const uint8_t* sgn, uint64_t* out, uint8_t* flags, size_t n, int block, hipStream_t s);
}  // namespace bls_launch
