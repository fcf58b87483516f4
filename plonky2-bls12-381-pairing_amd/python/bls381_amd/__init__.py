"""ctypes binding of libbls381_hip.so (include/bls381_hip.h) -- the Python-side twin of the Rust
`extern "C"` shim shown in INTEGRATION.md.  It mirrors the reference's two public functions

    optimized_miller_loop(P: G1Projective, Q: G2Projective) -> Fq12      src/miller_loop_native_optimized.rs:81
    multi_miller_loop(terms: &[(&G1Affine, &G2Affine)]) -> MillerLoopResult   src/miller_loop_native.rs:154

as batch calls over numpy uint64 buffers (6 little-endian limbs per Fp).  There is no CPU fallback: if the
HIP library or a GPU is missing, construction raises.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libbls381_hip.so"))

IO_ARK_MONT, IO_CANONICAL = 0, 1
OPT_IO_FORM, OPT_REF_OPT_FP2_SHORTCUT, OPT_BLOCK_SIZE, OPT_FEXP_MODE, OPT_FEXP_PIPELINES, OPT_PAIRING_HOMOGENEOUS_MILLER = 0, 1, 2, 3, 4, 5
OPT_FEXP_CHUNK_LOG2 = 6
# read-only (get_option): what the most recent call ran
INFO_LAST_FEXP_MODE, INFO_LAST_FEXP_PIPELINES, INFO_LAST_MILLER_HOMOGENEOUS, INFO_LAST_MILLER_GROUP = 100, 101, 102, 103
FEXP_SELF_CONTAINED, FEXP_BATCHED_INVERSION, FEXP_STAGED, FEXP_AUTO = 0, 1, 2, 3
FLAG_DEN_ZERO, FLAG_INFINITY = 1, 2
FLAG_TRUE = 32
POINT_ADD, POINT_DOUBLE, POINT_NORMALIZE, POINT_EQ, POINT_IS_ON_CURVE, POINT_IS_IN_SUBGROUP = range(6)
FLAG_NO_INVERSE, FLAG_NOT_SQUARE, FLAG_SIGN_UNSATISFIABLE = 4, 8, 16
FP_ADD, FP_SUB, FP_MUL, FP_SQR, FP_INV, FP_NEG, FP_TO_MONT, FP_FROM_MONT, FP_DIV, FP_LEGENDRE, FP_IS_SQUARE, FP_SGN0 = range(12)
FP2_ADD, FP2_SUB, FP2_MUL, FP2_SQR, FP2_INV, FP2_NEG, FP2_CONJ, FP2_MUL_BY_NONRESIDUE = range(8)
FP6_ADD, FP6_SUB, FP6_MUL, FP6_SQR, FP6_INV, FP6_NEG, FP6_MUL_BY_NONRESIDUE, FP6_FROBENIUS, FP6_MUL_BY_01, FP6_MUL_BY_1 = range(10)
FP12_MUL, FP12_SQR, FP12_INV, FP12_FROBENIUS, FP12_CYCLOTOMIC_SQR, FP12_CONJ, FP12_CYCLOTOMIC_EXP, FP12_MUL_BY_014, FP12_MUL_BY_014_KARATSUBA = range(9)

# every symbol include/bls381_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "bls381_create", "bls381_destroy", "bls381_set_option", "bls381_get_option", "bls381_set_stream", "bls381_use_own_stream",
    "bls381_sync", "bls381_strerror", "bls381_last_hip_error", "bls381_last_kernel_ms", "bls381_last_phase_ms",
    "bls381_miller_ref_opt_batch", "bls381_miller_ref_opt_batch_dev",
    "bls381_multi_miller_batch", "bls381_multi_miller_batch_dev",
    "bls381_multi_pairing_batch", "bls381_multi_pairing_batch_dev",
    "bls381_miller_ate_batch", "bls381_pairing_batch",
    "bls381_final_exp_batch", "bls381_final_exp_batch_dev",
    "bls381_fp_op_batch", "bls381_fp2_op_batch", "bls381_fp6_op_batch", "bls381_fp12_op_batch", "bls381_multi_miller_literal",
    "bls381_debug_line_ref_opt_batch", "bls381_g2_prepare_batch", "bls381_fp_pow_batch",
    "bls381_groth16_pairing_batch", "bls381_groth16_pairing_batch_dev",
    "bls381_prepared_pairing_batch", "bls381_prepared_pairing_batch_dev",
    "bls381_fq12_ark_to_myfq12", "bls381_fq12_myfq12_to_ark",
    "bls381_witness_fq12_inverse_batch", "bls381_witness_fq6_inverse_batch", "bls381_witness_fq2_inverse_batch",
    "bls381_witness_fq_sqrt_batch", "bls381_witness_fq2_sqrt_batch",
    "bls381_g1_op_batch", "bls381_g2_op_batch", "bls381_g1_decompress_batch", "bls381_g2_decompress_batch",
    "bls381_create_multi", "bls381_destroy_multi", "bls381_multi_device_count", "bls381_multi_context", "bls381_multi_set_option",
    "bls381_multi_pairing_batch_sharded", "bls381_multi_miller_batch_sharded", "bls381_miller_ref_opt_batch_sharded",
    "bls381_groth16_pairing_batch_sharded",
]


class Bls381Error(RuntimeError):
    pass


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise Bls381Error(f"{path} not found: build it with `python __graft_entry__.py` (hipcc, gfx950)")
    lib = ctypes.CDLL(path)
    u64p, u8p, sz, vp = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t, ctypes.c_void_p
    lib.bls381_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
    lib.bls381_destroy.argtypes = [vp]
    lib.bls381_destroy.restype = None
    lib.bls381_set_option.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    lib.bls381_get_option.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    lib.bls381_set_stream.argtypes = [vp, vp]
    lib.bls381_use_own_stream.argtypes = [vp]
    lib.bls381_sync.argtypes = [vp]
    lib.bls381_strerror.argtypes = [ctypes.c_int]
    lib.bls381_strerror.restype = ctypes.c_char_p
    lib.bls381_last_hip_error.argtypes = [vp]
    lib.bls381_last_kernel_ms.argtypes = [vp]
    lib.bls381_last_kernel_ms.restype = ctypes.c_float
    lib.bls381_last_phase_ms.argtypes = [vp, ctypes.c_int]
    lib.bls381_last_phase_ms.restype = ctypes.c_float
    for name in ("bls381_miller_ref_opt_batch", "bls381_miller_ate_batch", "bls381_pairing_batch"):
        getattr(lib, name).argtypes = [vp, u64p, u64p, u64p, u8p, sz]
    for name in ("bls381_multi_miller_batch", "bls381_multi_pairing_batch"):
        getattr(lib, name).argtypes = [vp, u64p, u64p, sz, u64p, u8p, sz]
    lib.bls381_final_exp_batch.argtypes = [vp, u64p, u64p, u8p, sz]
    lib.bls381_fp_op_batch.argtypes = [vp, ctypes.c_int, u64p, u64p, u64p, sz]
    lib.bls381_fp12_op_batch.argtypes = [vp, ctypes.c_int, u64p, u64p, u64p, sz]
    lib.bls381_fp2_op_batch.argtypes = [vp, ctypes.c_int, u64p, u64p, u64p, sz]
    lib.bls381_fp6_op_batch.argtypes = [vp, ctypes.c_int, u64p, u64p, u64p, sz]
    lib.bls381_multi_miller_literal.argtypes = [vp, u64p, sz]
    for name in ("bls381_fq12_ark_to_myfq12", "bls381_fq12_myfq12_to_ark"):
        getattr(lib, name).argtypes = [u64p, u64p, sz]
        getattr(lib, name).restype = None
    lib.bls381_debug_line_ref_opt_batch.argtypes = [vp, u64p, u64p, u64p, u64p, sz]
    lib.bls381_g2_prepare_batch.argtypes = [vp, u64p, u64p, u8p, sz]
    lib.bls381_fp_pow_batch.argtypes = [vp, u64p, u64p, sz, u64p, u8p, sz]
    lib.bls381_g1_op_batch.argtypes = [vp, ctypes.c_int, u64p, u64p, u64p, u8p, sz]
    lib.bls381_g2_op_batch.argtypes = [vp, ctypes.c_int, u64p, u64p, u64p, u8p, sz]
    lib.bls381_groth16_pairing_batch.argtypes = [vp, u64p, u64p, u64p, u64p, u8p, sz]
    lib.bls381_groth16_pairing_batch_dev.argtypes = [vp, vp, vp, vp, vp, vp, sz]
    lib.bls381_prepared_pairing_batch.argtypes = [vp, u64p, u64p, u64p, sz, sz, u64p, u8p, sz]
    lib.bls381_prepared_pairing_batch_dev.argtypes = [vp, vp, vp, vp, sz, sz, vp, vp, sz]
    u32p = ctypes.POINTER(ctypes.c_uint32)
    for name in ("bls381_witness_fq12_inverse_batch", "bls381_witness_fq6_inverse_batch", "bls381_witness_fq2_inverse_batch"):
        getattr(lib, name).argtypes = [vp, u32p, u32p, u8p, sz]
    for name in ("bls381_witness_fq_sqrt_batch", "bls381_witness_fq2_sqrt_batch"):
        getattr(lib, name).argtypes = [vp, u32p, u8p, u32p, u8p, sz]
    lib.bls381_g1_decompress_batch.argtypes = [vp, u64p, u8p, u64p, u8p, sz]
    lib.bls381_g2_decompress_batch.argtypes = [vp, u64p, u8p, u64p, u8p, sz]
    lib.bls381_create_multi.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.bls381_destroy_multi.argtypes = [vp]
    lib.bls381_destroy_multi.restype = None
    lib.bls381_multi_device_count.argtypes = [vp]
    lib.bls381_multi_context.argtypes = [vp, ctypes.c_int]
    lib.bls381_multi_context.restype = vp
    lib.bls381_multi_set_option.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    lib.bls381_multi_pairing_batch_sharded.argtypes = [vp, u64p, u64p, sz, u64p, u8p, sz]
    lib.bls381_multi_miller_batch_sharded.argtypes = [vp, u64p, u64p, sz, u64p, u8p, sz]
    lib.bls381_miller_ref_opt_batch_sharded.argtypes = [vp, u64p, u64p, u64p, u8p, sz]
    lib.bls381_groth16_pairing_batch_sharded.argtypes = [vp, u64p, u64p, u64p, u64p, u8p, sz]
    # device-pointer twins take raw addresses
    lib.bls381_miller_ref_opt_batch_dev.argtypes = [vp, vp, vp, vp, vp, sz]
    lib.bls381_multi_miller_batch_dev.argtypes = [vp, vp, vp, sz, vp, vp, sz]
    lib.bls381_multi_pairing_batch_dev.argtypes = [vp, vp, vp, sz, vp, vp, sz]
    lib.bls381_final_exp_batch_dev.argtypes = [vp, vp, vp, vp, sz]
    return lib


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


class MultiEngine:
    """Several devices behind one process (`bls381_multi`): contiguous ranges of the batch, one per device."""

    def __init__(self, device_ids=None, lib_path: str = LIB_PATH):
        self.lib = load_library(lib_path)
        self._m = ctypes.c_void_p()
        ids = None
        if device_ids is not None:
            ids = (ctypes.c_int * len(device_ids))(*device_ids)
        rc = self.lib.bls381_create_multi(ctypes.byref(self._m), ids, 0 if device_ids is None else len(device_ids))
        if rc != 0:
            raise Bls381Error(f"bls381_create_multi({device_ids}): {self.lib.bls381_strerror(rc).decode()}")

    def close(self):
        if getattr(self, "_m", None):
            self.lib.bls381_destroy_multi(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_count(self):
        return self.lib.bls381_multi_device_count(self._m)

    def set_option(self, opt, value):
        rc = self.lib.bls381_multi_set_option(self._m, opt, value)
        if rc != 0:
            raise Bls381Error("bls381_multi_set_option: " + self.lib.bls381_strerror(rc).decode())

    def _run(self, fn, name, a, b, w1, w2, k=None, out=None):
        a, pa = _u64(a)
        b, pb = _u64(b)
        n = a.size // w1
        assert a.size == n * w1 and b.size == n * w2, "ragged input"
        if out is None:
            out = np.empty(n * 72, dtype=np.uint64)
        assert out.size == n * 72 and out.dtype == np.uint64 and out.flags.c_contiguous
        flags = np.zeros(n, dtype=np.uint8)
        po, pf = out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
        rc = fn(self._m, pa, pb, po, pf, n) if k is None else fn(self._m, pa, pb, k, po, pf, n)
        if rc != 0:
            raise Bls381Error(f"{name}: {self.lib.bls381_strerror(rc).decode()}")
        return out, flags

    def multi_pairing(self, g1_aff, g2_aff, k=1, out=None):
        return self._run(self.lib.bls381_multi_pairing_batch_sharded, "bls381_multi_pairing_batch_sharded", g1_aff, g2_aff, 12 * k, 24 * k, k, out)

    def multi_miller_loop(self, g1_aff, g2_aff, k=1, out=None):
        return self._run(self.lib.bls381_multi_miller_batch_sharded, "bls381_multi_miller_batch_sharded", g1_aff, g2_aff, 12 * k, 24 * k, k, out)

    def optimized_miller_loop(self, g1_jac, g2_jac, out=None):
        return self._run(self.lib.bls381_miller_ref_opt_batch_sharded, "bls381_miller_ref_opt_batch_sharded", g1_jac, g2_jac, 18, 36, None, out)

    def groth16_pairing(self, g1_aff, g2_var, prepared_fixed):
        """e(P0,Q0) e(P1,Q1) e(P2,Q2) per element with Q1, Q2 fixed for the batch (Engine.g2_prepare output), sharded over the devices"""
        a, pa = _u64(g1_aff)
        b, pb = _u64(g2_var)
        c, pc = _u64(prepared_fixed)
        n = a.size // 36
        assert a.size == n * 36 and b.size == n * 24 and c.size == 2 * 68 * 36, "ragged input"
        out = np.zeros(n * 72, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        rc = self.lib.bls381_groth16_pairing_batch_sharded(self._m, pa, pb, pc, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                                           flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n)
        if rc != 0:
            raise Bls381Error("bls381_groth16_pairing_batch_sharded: " + self.lib.bls381_strerror(rc).decode())
        return out, flags


class Engine:
    """One context on one GPU (`bls381_ctx`)."""

    def __init__(self, device: int = 0, lib_path: str = LIB_PATH):
        self.lib = load_library(lib_path)
        self._ctx = ctypes.c_void_p()
        rc = self.lib.bls381_create(ctypes.byref(self._ctx), device)
        if rc != 0:
            raise Bls381Error(f"bls381_create(device={device}): {self.lib.bls381_strerror(rc).decode()}")
        self.device = device

    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.bls381_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise Bls381Error(f"{what}: {self.lib.bls381_strerror(rc).decode()} (hip error {self.lib.bls381_last_hip_error(self._ctx)})")

    # options ---------------------------------------------------------------
    def set_option(self, opt, value):
        self._check(self.lib.bls381_set_option(self._ctx, opt, value), "bls381_set_option")

    def get_option(self, opt):
        v = ctypes.c_int()
        self._check(self.lib.bls381_get_option(self._ctx, opt, ctypes.byref(v)), "bls381_get_option")
        return v.value

    def set_stream(self, hip_stream: int | None):
        """Adopt a hipStream_t handle (0 = the legacy default stream); None returns to the context's own stream."""
        if hip_stream is None:
            self._check(self.lib.bls381_use_own_stream(self._ctx), "bls381_use_own_stream")
        else:
            self._check(self.lib.bls381_set_stream(self._ctx, ctypes.c_void_p(hip_stream)), "bls381_set_stream")

    def sync(self):
        self._check(self.lib.bls381_sync(self._ctx), "bls381_sync")

    def last_kernel_ms(self) -> float:
        return float(self.lib.bls381_last_kernel_ms(self._ctx))

    def last_phase_ms(self, phase: int) -> float:
        """Device time of the Miller loops (phase 0) / the final exponentiation (phase 1) of the most recent pairing call."""
        return float(self.lib.bls381_last_phase_ms(self._ctx, phase))

    # host-buffer batch calls ------------------------------------------------
    def optimized_miller_loop(self, g1_jac, g2_jac):
        """Batch form of `optimized_miller_loop` (src/miller_loop_native_optimized.rs:81).
        g1_jac: n*18 u64, g2_jac: n*36 u64 -> (n*72 u64 Fq12, n flag bytes)."""
        a, pa = _u64(g1_jac)
        b, pb = _u64(g2_jac)
        n = a.size // 18
        assert a.size == n * 18 and b.size == n * 36, "ragged input"
        out = np.zeros(n * 72, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        self._check(self.lib.bls381_miller_ref_opt_batch(self._ctx, pa, pb, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                                         flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n),
                    "bls381_miller_ref_opt_batch")
        return out, flags

    def _multi(self, fn, name, g1_aff, g2_aff, k, out=None):
        a, pa = _u64(g1_aff)
        b, pb = _u64(g2_aff)
        n = a.size // (12 * k)
        assert a.size == n * 12 * k and b.size == n * 24 * k, "ragged input"
        if out is None:
            out = np.zeros(n * 72, dtype=np.uint64)
        assert out.size == n * 72 and out.dtype == np.uint64 and out.flags.c_contiguous
        flags = np.zeros(n, dtype=np.uint8)
        self._check(fn(self._ctx, pa, pb, k, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                       flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n), name)
        return out, flags

    def multi_miller_loop(self, g1_aff, g2_aff, k=1):
        """Batch form of `multi_miller_loop` (src/miller_loop_native.rs:154) with the line steps live:
        n products of k Miller loops each."""
        return self._multi(self.lib.bls381_multi_miller_batch, "bls381_multi_miller_batch", g1_aff, g2_aff, k)

    def multi_pairing(self, g1_aff, g2_aff, k=1, out=None):
        """n products of k pairings; `out` (n*72 u64) lets a caller reuse its result buffer like a Rust caller reuses a Vec."""
        return self._multi(self.lib.bls381_multi_pairing_batch, "bls381_multi_pairing_batch", g1_aff, g2_aff, k, out)

    def final_exponentiation(self, f):
        a, pa = _u64(f)
        n = a.size // 72
        out = np.zeros(n * 72, dtype=np.uint64)
        self._check(self.lib.bls381_final_exp_batch(self._ctx, pa, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), None, n),
                    "bls381_final_exp_batch")
        return out

    def fp_op(self, op, a, b=None):
        a, pa = _u64(a)
        pb = None
        if b is not None:
            b, pb = _u64(b)
        n = a.size // 6
        out = np.zeros(n * 6, dtype=np.uint64)
        self._check(self.lib.bls381_fp_op_batch(self._ctx, op, pa, pb, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n),
                    "bls381_fp_op_batch")
        return out

    def _tower_op(self, fn, name, width, op, a, b):
        a, pa = _u64(a)
        pb = None
        if b is not None:
            b, pb = _u64(b)
        n = a.size // width
        out = np.zeros(n * width, dtype=np.uint64)
        self._check(fn(self._ctx, op, pa, pb, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n), name)
        return out

    def fp2_op(self, op, a, b=None):
        """Fp2 arithmetic (src/fields_as_trees/fq2_target_tree.rs:54-142) on n x 12 u64."""
        return self._tower_op(self.lib.bls381_fp2_op_batch, "bls381_fp2_op_batch", 12, op, a, b)

    def fp6_op(self, op, a, b=None):
        """Fp6 arithmetic (src/fields_as_trees/fq6_target_tree.rs:59-293) on n x 36 u64."""
        return self._tower_op(self.lib.bls381_fp6_op_batch, "bls381_fp6_op_batch", 36, op, a, b)

    def fp12_op(self, op, a, b=None):
        a, pa = _u64(a)
        pb = None
        if b is not None:
            b, pb = _u64(b)
        n = a.size // 72
        out = np.zeros(n * 72, dtype=np.uint64)
        self._check(self.lib.bls381_fp12_op_batch(self._ctx, op, pa, pb, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n),
                    "bls381_fp12_op_batch")
        return out

    def fp_pow(self, a, exponent: int):
        """Batch `pow_fq` (src/fields/helpers.rs:176-195): a[i]^exponent, one exponent (< 2^512) for the batch."""
        a, pa = _u64(a)
        n = a.size // 6
        nwords = max(1, (exponent.bit_length() + 63) // 64)
        e = np.array([(exponent >> (64 * i)) & (2 ** 64 - 1) for i in range(nwords)], dtype=np.uint64)
        out = np.zeros(n * 6, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        self._check(self.lib.bls381_fp_pow_batch(self._ctx, pa, e.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nwords,
                                                 out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                                 flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n), "bls381_fp_pow_batch")
        return out, flags

    def groth16_pairing(self, g1_aff, g2_var, prepared_fixed):
        """e(P0,Q0) e(P1,Q1) e(P2,Q2) per element with Q1, Q2 fixed for the batch and given as `g2_prepare` output
        (2 x 68 x 36 u64).  g1_aff: n*36 u64, g2_var: n*24 u64."""
        a, pa = _u64(g1_aff)
        b, pb = _u64(g2_var)
        c, pc = _u64(prepared_fixed)
        n = a.size // 36
        assert a.size == n * 36 and b.size == n * 24 and c.size == 2 * 68 * 36, "ragged input"
        out = np.zeros(n * 72, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        self._check(self.lib.bls381_groth16_pairing_batch(self._ctx, pa, pb, pc, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                                          flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n), "bls381_groth16_pairing_batch")
        return out, flags

    def prepared_pairing(self, g1_aff, g2_var, prepared_fixed, n_var, n_fixed):
        """Pairing products with n_fixed (1..2) batch-wide fixed G2 points given as `g2_prepare` output and n_var (0..1)
        per-element G2 points; g1_aff holds n x (n_var + n_fixed) points, variable pair first."""
        a, pa = _u64(g1_aff)
        c, pc = _u64(prepared_fixed)
        n = a.size // (12 * (n_var + n_fixed))
        pb = None
        if n_var:
            b, pb = _u64(g2_var)
            assert b.size == n * 24, "ragged input"
        assert a.size == n * 12 * (n_var + n_fixed) and c.size == n_fixed * 68 * 36, "ragged input"
        out = np.zeros(n * 72, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        self._check(self.lib.bls381_prepared_pairing_batch(self._ctx, pa, pb, pc, n_var, n_fixed,
                                                           out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                                           flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n), "bls381_prepared_pairing_batch")
        return out, flags

    def prepared_pairing_dev(self, g1, g2_var, prepared, n_var, n_fixed, out, flags, n):
        self._check(self.lib.bls381_prepared_pairing_batch_dev(self._ctx, g1, g2_var, prepared, n_var, n_fixed, out, flags, n),
                    "bls381_prepared_pairing_batch_dev")

    def groth16_pairing_dev(self, g1, g2_var, prepared, out, flags, n):
        self._check(self.lib.bls381_groth16_pairing_batch_dev(self._ctx, g1, g2_var, prepared, out, flags, n), "bls381_groth16_pairing_batch_dev")

    def g2_prepare(self, g2_aff):
        """68 line-coefficient triples per G2 point (`G2PreparedTarget::from`, src/miller_loop_target.rs:23-76)."""
        a, pa = _u64(g2_aff)
        n = a.size // 24
        assert a.size == n * 24, "ragged input"
        out = np.zeros(n * 68 * 36, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        self._check(self.lib.bls381_g2_prepare_batch(self._ctx, pa, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                                     flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n), "bls381_g2_prepare_batch")
        return out, flags

    def point_op(self, group, op, a, b=None):
        """bls381_g1_op_batch (group 1, 18 u64 per point) / bls381_g2_op_batch (group 2, 36 u64 per point) on Jacobian points.
        Returns (out, flags); predicates (POINT_EQ, POINT_IS_ON_CURVE) answer in flags & FLAG_TRUE and return out = None."""
        w = 18 if group == 1 else 36
        a, pa = _u64(a)
        n = a.size // w
        assert a.size == n * w, "ragged input"
        pb = None
        if b is not None:
            b, pb = _u64(b)
            assert b.size == a.size
        predicate = op in (POINT_EQ, POINT_IS_ON_CURVE, POINT_IS_IN_SUBGROUP)
        out = None if predicate else np.zeros(n * w, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        fn = self.lib.bls381_g1_op_batch if group == 1 else self.lib.bls381_g2_op_batch
        self._check(fn(self._ctx, op, pa, pb, None if predicate else out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                       flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n), "bls381_g%d_op_batch" % group)
        return out, flags

    def decompress(self, group, x, sign):
        """bls381_g1_decompress_batch (group 1: x n*6 u64) / bls381_g2_decompress_batch (group 2: x n*12 u64): the affine point
        with abscissa x whose ordinate has sgn0 == sign & 1 (sign & 2: infinity).  Returns (points n*12 / n*24 u64, flags)."""
        w = 6 if group == 1 else 12
        x, px = _u64(x)
        n = x.size // w
        sign = np.ascontiguousarray(sign, dtype=np.uint8)
        assert x.size == n * w and sign.size == n, "ragged input"
        out = np.zeros(n * 2 * w, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        fn = self.lib.bls381_g1_decompress_batch if group == 1 else self.lib.bls381_g2_decompress_batch
        u8p = ctypes.POINTER(ctypes.c_uint8)
        self._check(fn(self._ctx, px, sign.ctypes.data_as(u8p), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), flags.ctypes.data_as(u8p), n),
                    "bls381_g%d_decompress_batch" % group)
        return out, flags

    def line_ref_opt(self, q1_jac, q2_jac, p_jac):
        """Test surface for `optimized_line_function` (opt.rs:8-78): returns n x (n, d) Fp2 pairs."""
        a, pa = _u64(q1_jac)
        b, pb = _u64(q2_jac)
        c, pc = _u64(p_jac)
        n = a.size // 36
        out = np.zeros(n * 24, dtype=np.uint64)
        self._check(self.lib.bls381_debug_line_ref_opt_batch(self._ctx, pa, pb, pc, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n),
                    "bls381_debug_line_ref_opt_batch")
        return out

    # witness-generator backend: canonical 12 x u32 digits per Fp -------------------------------
    def _witness_inverse(self, fn_name, fp_per_elem, x):
        x = np.ascontiguousarray(x, dtype=np.uint32)
        n = x.size // (12 * fp_per_elem)
        assert x.size == n * 12 * fp_per_elem, "ragged input"
        out = np.zeros_like(x)
        flags = np.zeros(n, dtype=np.uint8)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        self._check(getattr(self.lib, fn_name)(self._ctx, x.ctypes.data_as(u32p), out.ctypes.data_as(u32p),
                                               flags.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n), fn_name)
        return out, flags

    def witness_fq12_inverse(self, x):
        """Fq12InverseGenerator::run_once (src/fields/fq12_target.rs:351-374), MyFq12 coefficient order."""
        return self._witness_inverse("bls381_witness_fq12_inverse_batch", 12, x)

    def witness_fq6_inverse(self, x):
        """Fq6InverseGenerator::run_once (src/fields/fq6_target.rs:395-418), MyFq6 coefficient order."""
        return self._witness_inverse("bls381_witness_fq6_inverse_batch", 6, x)

    def witness_fq2_inverse(self, x):
        """Fq2InverseGenerator::run_once (src/fields/fq2_target.rs:331-352): zero maps to zero."""
        return self._witness_inverse("bls381_witness_fq2_inverse_batch", 2, x)

    def _witness_sqrt(self, fn_name, fp_per_elem, x, sgn):
        x = np.ascontiguousarray(x, dtype=np.uint32)
        sgn = np.ascontiguousarray(sgn, dtype=np.uint8)
        n = x.size // (12 * fp_per_elem)
        assert x.size == n * 12 * fp_per_elem and sgn.size == n, "ragged input"
        out = np.zeros_like(x)
        flags = np.zeros(n, dtype=np.uint8)
        u32p, u8p = ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint8)
        self._check(getattr(self.lib, fn_name)(self._ctx, x.ctypes.data_as(u32p), sgn.ctypes.data_as(u8p), out.ctypes.data_as(u32p),
                                               flags.ctypes.data_as(u8p), n), fn_name)
        return out, flags

    def witness_fq_sqrt(self, x, sgn):
        """FqSqrtGenerator::run_once (src/fields/fq_target.rs:330-343)."""
        return self._witness_sqrt("bls381_witness_fq_sqrt_batch", 1, x, sgn)

    def witness_fq2_sqrt(self, x, sgn):
        """Fq2SqrtGenerator::run_once (src/fields/fq2_target.rs:385-410)."""
        return self._witness_sqrt("bls381_witness_fq2_sqrt_batch", 2, x, sgn)

    def multi_miller_loop_literal(self, n):
        """What the reference's multi_miller_loop returns today: Fq12::one()."""
        out = np.zeros(n * 72, dtype=np.uint64)
        self._check(self.lib.bls381_multi_miller_literal(self._ctx, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n),
                    "bls381_multi_miller_literal")
        return out

    # device-pointer calls (addresses of hipMalloc'ed buffers, e.g. torch tensor .data_ptr()) ----
    def optimized_miller_loop_dev(self, g1, g2, out, flags, n):
        self._check(self.lib.bls381_miller_ref_opt_batch_dev(self._ctx, g1, g2, out, flags, n), "bls381_miller_ref_opt_batch_dev")

    def multi_miller_loop_dev(self, g1, g2, k, out, flags, n):
        self._check(self.lib.bls381_multi_miller_batch_dev(self._ctx, g1, g2, k, out, flags, n), "bls381_multi_miller_batch_dev")

    def multi_pairing_dev(self, g1, g2, k, out, flags, n):
        self._check(self.lib.bls381_multi_pairing_batch_dev(self._ctx, g1, g2, k, out, flags, n), "bls381_multi_pairing_batch_dev")

    def final_exponentiation_dev(self, f, out, flags, n):
        self._check(self.lib.bls381_final_exp_batch_dev(self._ctx, f, out, flags, n), "bls381_final_exp_batch_dev")
