"""ctypes loader for oracle/libbls381_oracle.so -- the C restatement of the reference path (TEST INFRASTRUCTURE: checker and
CPU baseline only; never imported by the product).  Buffers as in include/bls381_hip.h, arkworks Montgomery form."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _ptr(a, ty=ctypes.c_uint64):
    return a.ctypes.data_as(ctypes.POINTER(ty)) if a is not None else None


class COracle:
    def __init__(self):
        override = os.environ.get("BLS381_ORACLE_SO")          # an instrumented build (tools/sanitize_host.sh)
        so = override or os.path.join(HERE, "libbls381_oracle.so")
        src = os.path.join(HERE, "bls381_oracle.c")
        if not override and (not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so))):
            subprocess.check_call(["make", "-C", HERE])
        self.lib = ctypes.CDLL(so)

    def miller_ref_opt(self, g1, g2, threads=1):
        n = len(g1) // 18
        out = np.zeros(n * 72, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        self.lib.oracle_miller_ref_opt_batch(_ptr(g1), _ptr(g2), _ptr(out), _ptr(flags, ctypes.c_uint8), ctypes.c_size_t(n), threads)
        return out, flags

    def multi(self, g1, g2, k=1, final_exp=True, threads=1):
        n = len(g1) // (12 * k)
        out = np.zeros(n * 72, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        fn = self.lib.oracle_multi_pairing_batch if final_exp else self.lib.oracle_multi_miller_batch
        rc = fn(_ptr(g1), _ptr(g2), ctypes.c_size_t(k), _ptr(out), _ptr(flags, ctypes.c_uint8), ctypes.c_size_t(n), threads)
        assert rc == 0
        return out, flags

    def final_exp(self, f, threads=1):
        n = len(f) // 72
        out = np.zeros(n * 72, dtype=np.uint64)
        self.lib.oracle_final_exp_batch(_ptr(f), _ptr(out), ctypes.c_size_t(n), threads)
        return out

    def fp12_op(self, op, a, b=None):
        n = len(a) // 72
        out = np.zeros(n * 72, dtype=np.uint64)
        assert self.lib.oracle_fp12_op(op, _ptr(a), _ptr(b), _ptr(out), ctypes.c_size_t(n)) == 0
        return out

    def fp_op(self, op, a, b=None):
        n = len(a) // 6
        out = np.zeros(n * 6, dtype=np.uint64)
        assert self.lib.oracle_fp_op(op, _ptr(a), _ptr(b), _ptr(out), ctypes.c_size_t(n)) == 0
        return out
