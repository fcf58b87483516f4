/* bls381_hip.h -- C ABI of libbls381_hip.so: batched BLS12-381 pairing engine for AMD MI355X (gfx950).
 *
 * This is the drop-in boundary for the native pairing path of NikolayKostadinov21/plonky2-bls12-381-pairing.
 * The reference has no FFI of its own; each entry point below replaces one Rust item (paths relative to the
 * reference repository) and is what a Rust `extern "C"` block would bind:
 *
 *   bls381_miller_ref_opt_batch   <- pub fn optimized_miller_loop(P: G1Projective, Q: G2Projective) -> Fq12
 *                                    src/miller_loop_native_optimized.rs:81-127 (line fn :8-78)
 *   bls381_multi_miller_batch     <- pub fn multi_miller_loop(terms: &[(&G1Affine, &G2Affine)]) -> MillerLoopResult
 *                                    src/miller_loop_native.rs:154-212, loop :89-116, steps :27-87, `ell` :118-152
 *   bls381_multi_pairing_batch    <- the same followed by MillerLoopResult::final_exponentiation
 *                                    src/fields_as_trees/miller_loop.rs:128-178
 *   bls381_final_exp_batch        <- final_exponentiation / f_conversion, same file :128-178
 *   bls381_fp_op_batch            <- Fq / Bls12_381Base operators, src/fields/bls12_381base.rs:174-268, :118-125
 *   bls381_fp12_op_batch          <- Fq12 operators used at src/miller_loop_native_optimized.rs:91-102,118 and
 *                                    the tower routines of src/fields_as_trees/fq12_target_tree.rs:53-176
 *
 * Conventions
 *   - All buffers are caller-owned; no allocation crosses the boundary.  `_batch` entry points take HOST
 *     pointers and copy; `_batch_dev` twins take DEVICE pointers (hipMalloc'ed on the context's device) and
 *     enqueue on the context's stream without synchronising (bls381_sync waits).
 *   - Fp element = 6 x uint64_t, little-endian limbs (48 bytes).  Default encoding is the arkworks in-memory
 *     Montgomery form a * 2^384 mod p [ark-ff 0.4.2]; BLS381_OPT_IO_FORM = BLS381_IO_CANONICAL switches to the
 *     canonical limbs of `Bls12_381Base(pub [u64; 6])` (src/fields/bls12_381base.rs:20-21).
 *     Fp2 = (c0, c1); Fp6 = (c0, c1, c2); Fp12 = (c0, c1): 12 Fp in the order c0.c0.c0, c0.c0.c1, c0.c1.c0, ...
 *     G1 Jacobian = (x, y, z) 3 Fp; G2 Jacobian = (x, y, z) 3 Fp2; affine = (x, y).  An affine point whose
 *     coordinates are all zero denotes the point at infinity (arkworks' `infinity` flag).
 *   - Return value 0 = success, otherwise a BLS381_E_* code (bls381_strerror).  Where the reference would
 *     panic, the element's output is zeroed and its flags[i] byte says why; flags may be NULL.
 *   - Results are bit-identical for every batch size, device and option setting.
 *   - A context is bound to ONE device; multi-GPU use = one context per GPU with the batch split in contiguous
 *     ranges -- elements are independent, nothing is exchanged between devices: either one process per GPU
 *     (bls381_amd/sharding.py, torch.distributed) or one process with a bls381_multi (end of this file).
 */
#ifndef BLS381_HIP_H
#define BLS381_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bls381_ctx bls381_ctx;

enum {
  BLS381_OK = 0,
  BLS381_E_INVALID = 1,     /* bad argument (NULL buffer, k = 0, unknown op/option) */
  BLS381_E_NO_DEVICE = 2,   /* no usable gfx950 device / HIP runtime failure at create */
  BLS381_E_HIP = 3,         /* HIP runtime error during a call (bls381_last_hip_error) */
  BLS381_E_NOMEM = 4        /* device allocation failed */
};

/* per-element flag bits */
enum {
  BLS381_FLAG_DEN_ZERO = 1, /* Mode A: f_den == 0, the reference panics at opt.rs:102; output zeroed */
  BLS381_FLAG_INFINITY = 2, /* an input point at infinity: pair skipped (result 1 if all pairs skipped); point ops: the result
                               (predicates: the first operand) is the point at infinity */
  BLS381_FLAG_TRUE = 32     /* point predicates (BLS381_POINT_EQ, BLS381_POINT_IS_ON_CURVE): the predicate holds */
};

enum { BLS381_OPT_IO_FORM = 0, BLS381_OPT_REF_OPT_FP2_SHORTCUT = 1,
       BLS381_OPT_BLOCK_SIZE = 2 /* lanes per workgroup: 64 (default), 128 or 256.  No measurable effect (nothing is shared between lanes);
          the kernels that park state in LDS -- Miller loops of two and three pairs, Mode A with Fq12 accumulators -- always run one
          wave per workgroup, and the prepared-coefficient kernels 256 lanes */,
       BLS381_OPT_FEXP_MODE = 3 /* how the final exponentiation (src/fields_as_trees/miller_loop.rs:128-178) is run; same output bits */,
       BLS381_OPT_FEXP_PIPELINES = 4 /* staged mode: 1..4 part-batch pipelines on internal streams; 0 (default) = automatic: 2 from 2^18
          elements, 1 below; an explicit setting applies from 2^14 elements */,
       BLS381_OPT_PAIRING_HOMOGENEOUS_MILLER = 5 /* 1 (default): bls381_multi_pairing_batch (and the per-element pair of the prepared /
          Groth16-shaped products) walks its Miller loop with homogeneous-
          projective curve steps (3 M + 6 S per doubling instead of Alg. 26's 3 M + 8 S, src/miller_loop_native.rs:27-55); the Miller
          value differs by an Fp2 factor that the final exponentiation kills, the pairing is bit-identical FOR POINTS ON THE CURVES
          (what arkworks' affine types hold); for other coordinate values only 0 reproduces the reference's arithmetic.
          Miller-only entry points always follow Alg. 26/27. */,
       BLS381_OPT_FEXP_CHUNK_LOG2 = 6 /* staged final exponentiation: a batch of more than 2^value elements runs in passes of 2^value
          (10..22, default 20: ~6.1 GB of stage scratch).  Lower it to cap the scratch; the tests lower it to run the multi-pass path. */ };
/* Read-only through bls381_get_option: what the most recent pairing / final-exponentiation call on this context actually ran
 * (the automatic choices depend on the batch size and on whether the stage scratch could be allocated); -1 before the first call. */
enum { BLS381_INFO_LAST_FEXP_MODE = 100       /* BLS381_FEXP_SELF_CONTAINED / _BATCHED_INVERSION / _STAGED; -1: no final exponentiation */,
       BLS381_INFO_LAST_FEXP_PIPELINES = 101  /* part-batch pipelines of the staged form (1 for the other forms) */,
       BLS381_INFO_LAST_MILLER_HOMOGENEOUS = 102 /* 1: the Miller loops walked homogeneous-projective curve steps, 0: Alg. 26/27 */,
       BLS381_INFO_LAST_MILLER_GROUP = 103    /* pairs per Miller launch (k for k <= 3, else 3: groups of three multiplied on device) */ };
enum { BLS381_FEXP_SELF_CONTAINED = 0,  /* one kernel, one Fermat inversion per element */
       BLS381_FEXP_BATCHED_INVERSION = 1, /* the Fp inversion under f^-1 (:175) from a batch-wide simultaneous inversion: three launches,
                                             64 B of context-owned device scratch per element */
       BLS381_FEXP_STAGED = 2,          /* additionally Karabina-compressed cyclotomic squarings, their decompression denominators
                                             inverted batch-wide: six stage kernels with an inversion kernel before each,
                                             ~5.8 KB of context-owned device scratch per element (batches above 2^20 elements
                                             run in passes of 2^20) */
       BLS381_FEXP_AUTO = 3,            /* STAGED for batches of at least 2^10 elements (final exponentiation alone: 4.8 vs 5.4 ms at 2^10,
                                             5.4 vs 6.4 at 2^15, 17.8 vs 21.8 at 2^17, 130 vs 162 at 2^20), BATCHED_INVERSION
                                             below, and for a call whose stage
                                             scratch cannot be allocated */
       BLS381_FEXP_DEFAULT = BLS381_FEXP_AUTO };
enum { BLS381_IO_ARK_MONT = 0, BLS381_IO_CANONICAL = 1 };

/* Fp ops (first-kernel surface).  b may be NULL for unary ops. */
enum { BLS381_FP_ADD = 0, BLS381_FP_SUB = 1, BLS381_FP_MUL = 2, BLS381_FP_SQR = 3, BLS381_FP_INV = 4,
       BLS381_FP_NEG = 5, BLS381_FP_TO_MONT = 6, BLS381_FP_FROM_MONT = 7,
       BLS381_FP_DIV = 8 /* a * b^-1 (src/fields/bls12_381base.rs:259-268); b = 0 gives 0 */,
       BLS381_FP_LEGENDRE = 9  /* a^((p-1)/2): 0, 1 or p-1 (FqTarget::legendre, src/fields/fq_target.rs:271-274) */,
       BLS381_FP_IS_SQUARE = 10 /* legendre == 1 -> field element 1, else 0 (fq_target.rs:276-280) */,
       BLS381_FP_SGN0 = 11      /* parity of the canonical integer as field element 0/1 (sgn0_fq, src/fields/helpers.rs:159-167) */ };
/* Fp2 ops (src/fields_as_trees/fq2_target_tree.rs:54-142) */
enum { BLS381_FP2_ADD = 0, BLS381_FP2_SUB = 1, BLS381_FP2_MUL = 2, BLS381_FP2_SQR = 3, BLS381_FP2_INV = 4 /* zero -> zero */,
       BLS381_FP2_NEG = 5, BLS381_FP2_CONJ = 6, BLS381_FP2_MUL_BY_NONRESIDUE = 7 /* x (1 + u) */ };
/* Fp6 ops (src/fields_as_trees/fq6_target_tree.rs:59-293).  For MUL_BY_01 / MUL_BY_1 operand b carries the sparse
 * factor's Fp2 coefficients (c0, c1) / (c1) in its first two / one Fp2 slots, the rest of the slot is ignored. */
enum { BLS381_FP6_ADD = 0, BLS381_FP6_SUB = 1, BLS381_FP6_MUL = 2, BLS381_FP6_SQR = 3, BLS381_FP6_INV = 4, BLS381_FP6_NEG = 5,
       BLS381_FP6_MUL_BY_NONRESIDUE = 6 /* x v */, BLS381_FP6_FROBENIUS = 7, BLS381_FP6_MUL_BY_01 = 8, BLS381_FP6_MUL_BY_1 = 9 };
/* Fp12 ops */
enum { BLS381_FP12_MUL = 0, BLS381_FP12_SQR = 1, BLS381_FP12_INV = 2, BLS381_FP12_FROBENIUS = 3,
       BLS381_FP12_CYCLOTOMIC_SQR = 4, BLS381_FP12_CONJ = 5, BLS381_FP12_CYCLOTOMIC_EXP = 6,
       BLS381_FP12_MUL_BY_014 = 7 /* sparse line multiplication (commented `mul_by_014`, src/miller_loop_native.rs:118-137; live twin
          src/fields_as_trees/fq12_target_tree.rs:157-176): a * (c0, c1, 0 | 0, c4, 0) with (c0, c1, c4) read from b's slots c0.c0, c0.c1,
          c1.c1 -- in the form the Miller kernels run it (six lazy three-product sums, one reduction each) */,
       BLS381_FP12_MUL_BY_014_KARATSUBA = 8 /* the same value by Karatsuba over Fq6 as the reference writes it */ };

/* Point ops on Jacobian points (arkworks `Projective`: x = X/Z^2, y = Y/Z^3; Z = 0 is the point at infinity) */
enum { BLS381_POINT_ADD = 0        /* ark-ec 0.4.2 `Projective += &Projective` (add-2007-bl with the equal -> double and
                                      infinity fallbacks) [third party]: the `R + R`, `R + Q` of src/miller_loop_native_optimized.rs:93,98 */,
       BLS381_POINT_DOUBLE = 1     /* `double_in_place` (dbl-2009-l, a = 0) */,
       BLS381_POINT_NORMALIZE = 2  /* the Z = 1 representative (`.into()` to affine and back, opt.rs:93,98;
                                      `From<G2AffineTarget> for G2ProjectiveTarget`, src/fields_as_trees/g2_curve.rs:29-50);
                                      infinity -> (1, 1, 0) */,
       BLS381_POINT_EQ = 3         /* point equality, flags bit BLS381_FLAG_TRUE (`is_point_equal_to`, g2_curve.rs:129-149,
                                      g1_curve.rs:79-110: both at infinity, or neither and equal coordinates) */,
       BLS381_POINT_IS_ON_CURVE = 4 /* Y^2 == X^3 + b Z^6, b = 4 (G1) / 4(1+u) (G2); flags bit BLS381_FLAG_TRUE */,
       BLS381_POINT_IS_IN_SUBGROUP = 5 /* on the curve AND [r] P == infinity (r at src/miller_loop_native_optimized.rs:110): what
                                          arkworks checks when it deserialises a `G1Affine` / `G2Affine` [third party]; the pairing
                                          entry points assume it.  flags bit BLS381_FLAG_TRUE */ };

/* lifecycle ------------------------------------------------------------------------------------------ */
int bls381_create(bls381_ctx** out, int device_id);
void bls381_destroy(bls381_ctx* ctx);
int bls381_set_option(bls381_ctx* ctx, int option, int value);
int bls381_get_option(const bls381_ctx* ctx, int option, int* value);
/* Adopt an existing hipStream_t (e.g. the caller framework's current stream): the `_dev` entry points enqueue behind the work
 * already queued on it.  NULL is a stream like any other -- the legacy default stream (handle 0, what torch calls its
 * default stream).  A context starts on a private non-blocking stream of its own; bls381_use_own_stream returns to it. */
int bls381_set_stream(bls381_ctx* ctx, void* hip_stream);
int bls381_use_own_stream(bls381_ctx* ctx);
int bls381_sync(bls381_ctx* ctx);
const char* bls381_strerror(int code);
int bls381_last_hip_error(const bls381_ctx* ctx);
/* Device time in ms of the most recent pairing / Miller-loop / final-exponentiation call on this context (every launch of the
 * call, between two HIP events recorded on the context's stream); waits for the call to finish; < 0 if there was none. */
float bls381_last_kernel_ms(const bls381_ctx* ctx);
/* The same call split at the boundary between its Miller loops (phase 0: the k_multi_miller / k_prepared_miller launches) and its
 * final exponentiation (phase 1); < 0 when the most recent call was not a pairing / pairing-product call. */
float bls381_last_phase_ms(const bls381_ctx* ctx, int phase);

/* Mode A: optimized_miller_loop over a batch (src/miller_loop_native_optimized.rs:81) -- the same Fq12 value bit for bit, quirks
 * included (LSB-first walk, numerator / denominator pair, Jacobian coordinates read as homogeneous, one final squaring; any
 * coordinate values, on a curve or not).  The accumulators are full Fq12 values; where the reference spells a generic `*` or `+`
 * whose operands have structure -- `f * f`, `f * Fq12::new(Fq6::new(v, 0, 0), 0)`, `R + R` -- the kernel runs the squaring formula,
 * six Fp2 products, the doubling formulas (same field element in each case).  BLS381_OPT_REF_OPT_FP2_SHORTCUT = 1 keeps the
 * accumulators in Fp2, where they provably live.
 * g1_jac: n x 3 x 6 u64, g2_jac: n x 3 x 2 x 6, out_fq12: n x 12 x 6, flags: n bytes or NULL (bit BLS381_FLAG_DEN_ZERO: the
 * reference would panic at :102, output zeroed). */
int bls381_miller_ref_opt_batch(bls381_ctx* ctx, const uint64_t* g1_jac, const uint64_t* g2_jac,
                                uint64_t* out_fq12, uint8_t* flags, size_t n);
int bls381_miller_ref_opt_batch_dev(bls381_ctx* ctx, const uint64_t* g1_jac, const uint64_t* g2_jac,
                                    uint64_t* out_fq12, uint8_t* flags, size_t n);

/* Mode B: product over k pairs of optimal-ate Miller loops -- `multi_miller_loop(terms: &[(&G1Affine, &G2Affine)])`,
 * src/miller_loop_native.rs:154 -- for a slice of ANY length k >= 1.  Up to three pairs share one loop (one squaring per
 * iteration for all of them, :100-106); longer products run in groups of three whose Miller values are multiplied on the
 * device (the same field element: squaring distributes over the product), with one final exponentiation at the end.
 * g1_aff: n x k x 2 x 6, g2_aff: n x k x 2 x 2 x 6, out: n x 12 x 6. */
int bls381_multi_miller_batch(bls381_ctx* ctx, const uint64_t* g1_aff, const uint64_t* g2_aff, size_t k,
                              uint64_t* out_fq12, uint8_t* flags, size_t n);
int bls381_multi_miller_batch_dev(bls381_ctx* ctx, const uint64_t* g1_aff, const uint64_t* g2_aff, size_t k,
                                  uint64_t* out_fq12, uint8_t* flags, size_t n);
/* ... followed by the final exponentiation: full pairings / pairing products (Groth16 shape: k = 3) */
int bls381_multi_pairing_batch(bls381_ctx* ctx, const uint64_t* g1_aff, const uint64_t* g2_aff, size_t k,
                               uint64_t* out_gt, uint8_t* flags, size_t n);
int bls381_multi_pairing_batch_dev(bls381_ctx* ctx, const uint64_t* g1_aff, const uint64_t* g2_aff, size_t k,
                                   uint64_t* out_gt, uint8_t* flags, size_t n);
/* k = 1 conveniences */
int bls381_miller_ate_batch(bls381_ctx* ctx, const uint64_t* g1_aff, const uint64_t* g2_aff,
                            uint64_t* out_fq12, uint8_t* flags, size_t n);
int bls381_pairing_batch(bls381_ctx* ctx, const uint64_t* g1_aff, const uint64_t* g2_aff,
                         uint64_t* out_gt, uint8_t* flags, size_t n);

/* final exponentiation alone (src/fields_as_trees/miller_loop.rs:166-178) */
int bls381_final_exp_batch(bls381_ctx* ctx, const uint64_t* in_fq12, uint64_t* out_gt, uint8_t* flags, size_t n);
int bls381_final_exp_batch_dev(bls381_ctx* ctx, const uint64_t* in_fq12, uint64_t* out_gt, uint8_t* flags, size_t n);

/* field-level surfaces */
int bls381_fp_op_batch(bls381_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int bls381_fp2_op_batch(bls381_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);   /* n x 2 x 6 */
int bls381_fp6_op_batch(bls381_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);   /* n x 6 x 6 */
int bls381_fp12_op_batch(bls381_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);

/* Groth16-verify shape with FIXED G2 points (BASELINE config 4): out[i] = e(P0_i, Q0_i) e(P1_i, Q1) e(P2_i, Q2), where
 * Q1, Q2 are shared by the batch and given as their prepared coefficients -- the output of bls381_g2_prepare_batch for
 * the two points, 2 x 68 x 3 x 2 x 6 u64 (`G2PreparedTarget`, src/miller_loop_target.rs:23-76).  The coefficients are
 * staged in LDS once per workgroup; only Q0 pays doubling/addition steps.  Same value as
 * bls381_multi_pairing_batch(k = 3) on (P0,Q0),(P1,Q1),(P2,Q2).  g1_aff: n x 3 x 2 x 6, g2_var: n x 2 x 2 x 6. */
int bls381_groth16_pairing_batch(bls381_ctx* ctx, const uint64_t* g1_aff, const uint64_t* g2_var,
                                 const uint64_t* prepared_fixed, uint64_t* out_gt, uint8_t* flags, size_t n);
int bls381_groth16_pairing_batch_dev(bls381_ctx* ctx, const uint64_t* g1_aff, const uint64_t* g2_var,
                                     const uint64_t* prepared_fixed, uint64_t* out_gt, uint8_t* flags, size_t n);

/* General form: n_var in {0, 1} pairs with a per-element G2 point followed by n_fixed in {1, 2} pairs whose G2 point is
 * fixed for the batch (prepared coefficients, n_fixed x 68 x 3 x 2 x 6 u64).  g1_aff: n x (n_var + n_fixed) x 2 x 6 in that
 * order; g2_var: n x 2 x 2 x 6 (ignored when n_var = 0).  (1, 2) is the Groth16 shape above; (0, 1) pairs many G1 points
 * against ONE G2 point with no curve arithmetic at all in the Miller loop. */
int bls381_prepared_pairing_batch(bls381_ctx* ctx, const uint64_t* g1_aff, const uint64_t* g2_var, const uint64_t* prepared_fixed,
                                  size_t n_var, size_t n_fixed, uint64_t* out_gt, uint8_t* flags, size_t n);
int bls381_prepared_pairing_batch_dev(bls381_ctx* ctx, const uint64_t* g1_aff, const uint64_t* g2_var, const uint64_t* prepared_fixed,
                                      size_t n_var, size_t n_fixed, uint64_t* out_gt, uint8_t* flags, size_t n);

/* pow_fq (src/fields/helpers.rs:176-195): out[i] = a[i]^exp for one exponent of exp_words (1..8) little-endian u64
 * words shared by the batch.  Quirks kept: exp == 0 returns a[i] itself; where the exponent's non-adjacent form has
 * a -1 digit the reference divides by a[i], so a[i] == 0 sets BLS381_FLAG_NO_INVERSE (value 0). */
int bls381_fp_pow_batch(bls381_ctx* ctx, const uint64_t* a, const uint64_t* exp, size_t exp_words, uint64_t* out,
                        uint8_t* flags, size_t n);

/* G2 precomputation: the 68 line-coefficient triples of each point in Miller-loop order
 * (63 doubling + 5 addition steps; `G2PreparedTarget::from`, src/miller_loop_target.rs:23-76, len == 68 at :69).
 * g2_aff: n x 2 x 2 x 6, coeffs: n x 68 x 3 x 2 x 6 u64 (19 584 bytes per point; triple = (a, b, c) with
 * ell(f) = f.mul_by_014(c, b * P.x, a * P.y)).  An infinite input is replaced by the generator and flagged
 * BLS381_FLAG_INFINITY, as the reference does (:58-60). */
int bls381_g2_prepare_batch(bls381_ctx* ctx, const uint64_t* g2_aff, uint64_t* coeffs, uint8_t* flags, size_t n);

/* G1 / G2 point ops (src/curves/, src/fields_as_trees/g{1,2}_curve.rs; arkworks group law at opt.rs:93,98).
 * a, b, out: n x 3 x 6 u64 (G1) / n x 3 x 2 x 6 (G2) Jacobian points; b is read by ADD and EQ only.  ADD / DOUBLE / NORMALIZE
 * write `out` (flags optional: BLS381_FLAG_INFINITY when the result is the point at infinity); EQ / IS_ON_CURVE / IS_IN_SUBGROUP
 * answer in `flags` (required) and leave `out` untouched (may be NULL). */
int bls381_g1_op_batch(bls381_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* flags, size_t n);
int bls381_g2_op_batch(bls381_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* flags, size_t n);

/* Point decompression, from the reference's own pieces: out[i] = the AFFINE point (x, y) with the given abscissa,
 * y^2 = x^3 + 4 (G1) / x^3 + 4 (1 + u) (G2), whose ordinate satisfies sgn0(y) == sign[i] & 1 -- sgn0_fq / sgn0_fq2 of
 * src/fields/helpers.rs:159-174 (parity of the canonical integer; for Fq2 of c0, or of c1 when c0 == 0), the square roots of
 * FqSqrtGenerator / Fq2SqrtGenerator (src/fields/fq_target.rs:330-343, src/fields/fq2_target.rs:385-410).  sign[i] & 2 marks the
 * point at infinity (all-zero output, BLS381_FLAG_INFINITY).  x that is not an abscissa of the curve: all-zero output and
 * BLS381_FLAG_NOT_SQUARE.  Subgroup membership is a separate question (BLS381_POINT_IS_IN_SUBGROUP).
 * x: n x 6 u64 (G1) / n x 2 x 6 (G2) in the context's encoding; out_aff: n x 2 x 6 / n x 2 x 2 x 6; sign: n bytes. */
int bls381_g1_decompress_batch(bls381_ctx* ctx, const uint64_t* x, const uint8_t* sign, uint64_t* out_aff, uint8_t* flags, size_t n);
int bls381_g2_decompress_batch(bls381_ctx* ctx, const uint64_t* x, const uint8_t* sign, uint64_t* out_aff, uint8_t* flags, size_t n);

/* Witness-generator backend: batch forms of the `SimpleGenerator::run_once` hooks through which
 * plonky2 invokes native arithmetic while filling a witness.  Elements are CANONICAL integers as 12 x u32
 * little-endian digits per Fp (the BigUint limb format of FqTarget, src/fields/fq_target.rs:289-313; identical in
 * memory to 6 x u64), Fq12 / Fq6 in the flat MyFq12 / MyFq6 coefficient orders (src/fields/helpers.rs:39-41,
 * src/fields/my_fq6.rs:25).  Independent of BLS381_OPT_IO_FORM.  Extra flag bits: */
enum {
  BLS381_FLAG_NO_INVERSE = 4,          /* zero input where the generator does `inverse().unwrap()` */
  BLS381_FLAG_NOT_SQUARE = 8,          /* `sqrt().unwrap()` would panic */
  BLS381_FLAG_SIGN_UNSATISFIABLE = 16  /* root is 0 but sgn = 1 requested: the generator's assert_eq! fails */
};
/* Fq12InverseGenerator::run_once, src/fields/fq12_target.rs:351-374.  x, inv: n x 12 x 12 u32 */
int bls381_witness_fq12_inverse_batch(bls381_ctx* ctx, const uint32_t* x, uint32_t* inv, uint8_t* flags, size_t n);
/* Fq6InverseGenerator::run_once, src/fields/fq6_target.rs:395-418.  x, inv: n x 6 x 12 u32 */
int bls381_witness_fq6_inverse_batch(bls381_ctx* ctx, const uint32_t* x, uint32_t* inv, uint8_t* flags, size_t n);
/* Fq2InverseGenerator::run_once, src/fields/fq2_target.rs:331-352 (zero maps to zero).  x, inv: n x 2 x 12 u32 */
int bls381_witness_fq2_inverse_batch(bls381_ctx* ctx, const uint32_t* x, uint32_t* inv, uint8_t* flags, size_t n);
/* FqSqrtGenerator::run_once, src/fields/fq_target.rs:330-343: the root whose sgn0 (parity) equals sgn[i] & 1 */
int bls381_witness_fq_sqrt_batch(bls381_ctx* ctx, const uint32_t* x, const uint8_t* sgn, uint32_t* sqrt, uint8_t* flags, size_t n);
/* Fq2SqrtGenerator::run_once, src/fields/fq2_target.rs:385-410 (sgn0_fq2, src/fields/helpers.rs:169-174) */
int bls381_witness_fq2_sqrt_batch(bls381_ctx* ctx, const uint32_t* x, const uint8_t* sgn, uint32_t* sqrt, uint8_t* flags, size_t n);

/* Coefficient-order helpers, host only (no context, no device): arkworks tree order of the ABI <-> the flat MyFq12 order
 * of src/fields/helpers.rs:22-76 (coeffs[k] = w^k u^0 part, coeffs[k + 6] = w^k u^1 part).  in != out. */
void bls381_fq12_ark_to_myfq12(const uint64_t* in, uint64_t* out, size_t n);
void bls381_fq12_myfq12_to_ark(const uint64_t* in, uint64_t* out, size_t n);

/* Test surface for the private `optimized_line_function` (src/miller_loop_native_optimized.rs:8-78): all three
 * branches, including the vertical one (:62-77) the loop never reaches on subgroup inputs.
 * q1_jac, q2_jac: n x 3 x 2 x 6, p_jac: n x 3 x 6, out_nd: n x (n, d) = n x 2 x 2 x 6. */
int bls381_debug_line_ref_opt_batch(bls381_ctx* ctx, const uint64_t* q1_jac, const uint64_t* q2_jac,
                                    const uint64_t* p_jac, uint64_t* out_nd, size_t n);

/* What src/miller_loop_native.rs:154-212 returns TODAY for any input (its line steps are commented out):
 * Fq12::one().  Host-only, here so that a caller can reproduce the literal reference behaviour. */
int bls381_multi_miller_literal(bls381_ctx* ctx, uint64_t* out_fq12, size_t n);

/* Several devices behind ONE host process (BASELINE configs "2^23 pairings / 3 x 2^18 proofs across 8 GPUs" for a
 * caller that is a single program, like the reference's).  A bls381_multi owns one context per listed device
 * (device_ids == NULL or n_devices <= 0: every visible device; an id may be listed more than once).  The `_sharded` calls cut the
 * batch into contiguous ranges [g n / G, (g + 1) n / G), one per context, and run each range through that device's host-pointer
 * entry point on its own host thread -- double-buffered copies on the context's copy streams, kernels on its compute stream.
 * Elements are independent, so nothing is exchanged between devices; results are bit-identical to the one-device call.
 * With a single context this IS the one-device call. */
typedef struct bls381_multi bls381_multi;
int bls381_create_multi(bls381_multi** out, const int* device_ids, int n_devices);
void bls381_destroy_multi(bls381_multi* m);
int bls381_multi_device_count(const bls381_multi* m);
bls381_ctx* bls381_multi_context(bls381_multi* m, int index);           /* e.g. to query one device's last_kernel_ms */
int bls381_multi_set_option(bls381_multi* m, int option, int value);    /* on every context */
/* multi_miller_loop over n independent k-pair products followed by the final exponentiation (src/miller_loop_native.rs:154,
 * src/fields_as_trees/miller_loop.rs:128-178); same buffers as bls381_multi_pairing_batch */
int bls381_multi_pairing_batch_sharded(bls381_multi* m, const uint64_t* g1_aff, const uint64_t* g2_aff, size_t k,
                                       uint64_t* out_gt, uint8_t* flags, size_t n);
int bls381_multi_miller_batch_sharded(bls381_multi* m, const uint64_t* g1_aff, const uint64_t* g2_aff, size_t k,
                                      uint64_t* out_fq12, uint8_t* flags, size_t n);
/* optimized_miller_loop (src/miller_loop_native_optimized.rs:81); same buffers as bls381_miller_ref_opt_batch */
int bls381_miller_ref_opt_batch_sharded(bls381_multi* m, const uint64_t* g1_jac, const uint64_t* g2_jac,
                                        uint64_t* out_fq12, uint8_t* flags, size_t n);
/* the Groth16-verify shape with its two fixed G2 points prepared (BASELINE config 4 for a single-process caller): same buffers as
 * bls381_groth16_pairing_batch; every device stages the 2 x 68 coefficient triples (src/miller_loop_target.rs:23-76) itself */
int bls381_groth16_pairing_batch_sharded(bls381_multi* m, const uint64_t* g1_aff, const uint64_t* g2_var, const uint64_t* prepared_fixed,
                                         uint64_t* out_gt, uint8_t* flags, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* BLS381_HIP_H */
