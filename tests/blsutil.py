"""Shared helpers for the test-suite: encoding between the oracle's Python ints and the C-ABI buffers
(6 x u64 little-endian per Fp), seeded synthetic inputs, and library loaders.  Test infrastructure only."""
from __future__ import annotations

import ctypes
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "plonky2-bls12-381-pairing_amd", "python"))

import pyref as O  # noqa: E402

IO_ARK_MONT, IO_CANONICAL = 0, 1


# ------------------------------------------------------------------ encoding
def enc_fp(v, form=IO_ARK_MONT):
    return O.fp_to_limbs(v, mont=(form == IO_ARK_MONT))


def dec_fp(limbs, form=IO_ARK_MONT):
    return O.fp_from_limbs([int(x) for x in limbs], mont=(form == IO_ARK_MONT))


def enc_fps(vals, form=IO_ARK_MONT):
    """flat list of Fp ints -> uint64 array of 6 limbs each"""
    out = np.empty(len(vals) * 6, dtype=np.uint64)
    for i, v in enumerate(vals):
        out[6 * i:6 * i + 6] = np.array(enc_fp(v, form), dtype=np.uint64)
    return out


def dec_fps(arr, form=IO_ARK_MONT):
    a = np.asarray(arr, dtype=np.uint64).reshape(-1, 6)
    return [dec_fp(row, form) for row in a]


def flat_g1_jac(p):
    return [p[0], p[1], p[2]]


def flat_g2_jac(q):
    return [c for co in q for c in co]


def flat_g1_aff(p):
    return [0, 0] if p is None else [p[0], p[1]]


def flat_g2_aff(q):
    return [0, 0, 0, 0] if q is None else [q[0][0], q[0][1], q[1][0], q[1][1]]


def enc_fp12s(vals, form=IO_ARK_MONT):
    flat = []
    for v in vals:
        flat += O.fp12_flatten(v)
    return enc_fps(flat, form)


def dec_fp12s(arr, form=IO_ARK_MONT):
    f = dec_fps(arr, form)
    return [O.fp12_unflatten(f[12 * i:12 * i + 12]) for i in range(len(f) // 12)]


# ------------------------------------------------------------------ synthetic inputs
_POOL = {}


def point_pool(k=16):
    """(i+1)*G1, (i+1)*G2 for i < k (affine), built by repeated addition."""
    if k in _POOL:
        return _POOL[k]
    g1j = O.jac_from_affine(O._F1, O.G1_GEN)
    g2j = O.jac_from_affine(O._F2, O.G2_GEN)
    a1, a2 = g1j, g2j
    p1, p2 = [], []
    for _ in range(k):
        p1.append(O.jac_to_affine(O._F1, a1))
        p2.append(O.jac_to_affine(O._F2, a2))
        a1 = O.jac_add(O._F1, a1, g1j)
        a2 = O.jac_add(O._F2, a2, g2j)
    _POOL[k] = (p1, p2)
    return _POOL[k]


def rand_fp(rng):
    return rng.randrange(1, O.P)


def rand_fp2(rng):
    return (rng.randrange(O.P), rng.randrange(O.P))


def rand_fp12(rng):
    return O.fp12_unflatten([rng.randrange(O.P) for _ in range(12)])


def random_pairs(n, seed, jacobian=True):
    """n seeded (P, Q) pairs from the pool; Jacobian representatives are re-randomised with a random
    lambda (general Z, like the reference test's G*Projective::rand inputs, opt.rs:141-142)."""
    rng = random.Random(seed)
    p1, p2 = point_pool(16)
    out = []
    for _ in range(n):
        P_aff = p1[rng.randrange(len(p1))]
        Q_aff = p2[rng.randrange(len(p2))]
        if jacobian:
            lp, lq = rand_fp(rng), (rand_fp(rng), rand_fp(rng))
            Pj = O.jac_rerandomise(O._F1, O.jac_from_affine(O._F1, P_aff), lp)
            Qj = O.jac_rerandomise(O._F2, O.jac_from_affine(O._F2, Q_aff), lq)
            out.append((Pj, Qj))
        else:
            out.append((P_aff, Q_aff))
    return out


# ------------------------------------------------------------------ libraries
def _ptr(a, ty=ctypes.c_uint64):
    return a.ctypes.data_as(ctypes.POINTER(ty)) if a is not None else None


def build_hostsim():
    """g++ build of the device headers for the host (tests/hostsim); returns the ctypes handle."""
    if os.environ.get("BLS381_HOSTSIM_SO"):            # an instrumented build (tools/sanitize_host.sh)
        return ctypes.CDLL(os.environ["BLS381_HOSTSIM_SO"])
    src = os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp")
    so = os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")
    inc = os.path.join(ROOT, "plonky2-bls12-381-pairing_amd", "csrc")
    deps = [src] + [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith((".hip.h", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        import subprocess
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I", inc, src, "-o", so])
    return ctypes.CDLL(so)


class HostSim:
    """Thin caller for tests/hostsim/libhostsim.so (same buffer conventions as the C ABI)."""

    def __init__(self):
        self.lib = build_hostsim()

    def fp_op(self, op, a, b=None, form=IO_ARK_MONT):
        n = len(a) // 6
        out = np.zeros(n * 6, dtype=np.uint64)
        rc = self.lib.hostsim_fp_op(op, form, _ptr(a), _ptr(b), _ptr(out), ctypes.c_size_t(n))
        assert rc == 0
        return out

    def fp2_op(self, op, a, b=None, form=IO_ARK_MONT):
        n = len(a) // 12
        out = np.zeros(n * 12, dtype=np.uint64)
        rc = self.lib.hostsim_fp2_op(op, form, _ptr(a), _ptr(b), _ptr(out), ctypes.c_size_t(n))
        assert rc == 0
        return out

    def fp12_op(self, op, a, b=None, form=IO_ARK_MONT):
        n = len(a) // 72
        out = np.zeros(n * 72, dtype=np.uint64)
        rc = self.lib.hostsim_fp12_op(op, form, _ptr(a), _ptr(b), _ptr(out), ctypes.c_size_t(n))
        assert rc == 0
        return out

    def miller_ref_opt(self, g1, g2, generic=True, form=IO_ARK_MONT):
        n = len(g1) // 18
        out = np.zeros(n * 72, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        rc = self.lib.hostsim_miller_ref_opt(form, int(generic), _ptr(g1), _ptr(g2), _ptr(out),
                                             _ptr(flags, ctypes.c_uint8), ctypes.c_size_t(n))
        assert rc == 0
        return out, flags

    def miller_ref_opt_lds(self, g1, g2, form=IO_ARK_MONT):
        """Mode A as k_miller_ref_opt_wide runs it: one Fq12 accumulator in registers, the other in (here: a plain buffer standing in for) LDS"""
        n = len(g1) // 18
        out = np.zeros(n * 72, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        rc = self.lib.hostsim_miller_ref_opt_lds(form, _ptr(g1), _ptr(g2), _ptr(out), _ptr(flags, ctypes.c_uint8), ctypes.c_size_t(n))
        assert rc == 0
        return out, flags

    def prepared_pairing(self, g1, g2_var, prepared, n_var, n_fixed, hom=True, form=IO_ARK_MONT):
        """k_prepared_miller's loop (pairing.hip.h miller_prepared) + final exponentiation; buffers as bls381_prepared_pairing_batch"""
        n = len(g1) // (12 * (n_var + n_fixed))
        out = np.zeros(n * 72, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        rc = self.lib.hostsim_prepared_pairing(form, n_var, n_fixed, int(hom), _ptr(g1), _ptr(g2_var) if n_var else None, _ptr(prepared),
                                               _ptr(out), _ptr(flags, ctypes.c_uint8), ctypes.c_size_t(n))
        assert rc == 0
        return out, flags

    def multi_miller_lds(self, g1, g2, k, final_exp=False, form=IO_ARK_MONT):
        """products of k = 2, 3 pairs as k_multi_miller<2|3> runs them: the idle pairs' R in the slot store (miller_ate_lds)"""
        n = len(g1) // (12 * k)
        out = np.zeros(n * 72, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        rc = self.lib.hostsim_multi_miller_lds(form, int(final_exp), _ptr(g1), _ptr(g2), ctypes.c_size_t(k), _ptr(out),
                                               _ptr(flags, ctypes.c_uint8), ctypes.c_size_t(n))
        assert rc == 0
        return out, flags

    def multi_miller(self, g1, g2, k=1, final_exp=False, form=IO_ARK_MONT):
        n = len(g1) // (12 * k)
        out = np.zeros(n * 72, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        rc = self.lib.hostsim_multi_miller(form, int(final_exp), _ptr(g1), _ptr(g2), ctypes.c_size_t(k), _ptr(out),
                                           _ptr(flags, ctypes.c_uint8), ctypes.c_size_t(n))
        assert rc == 0
        return out, flags

    def fp_pow(self, a, exponent, form=IO_ARK_MONT):
        n = len(a) // 6
        nwords = max(1, (exponent.bit_length() + 63) // 64)
        e = np.array([(exponent >> (64 * i)) & (2 ** 64 - 1) for i in range(nwords)], dtype=np.uint64)
        out = np.zeros(n * 6, dtype=np.uint64)
        assert self.lib.hostsim_fp_pow(form, _ptr(a), _ptr(e), nwords, _ptr(out), ctypes.c_size_t(n)) == 0
        return out

    def g2_prepare(self, g2, form=IO_ARK_MONT):
        n = len(g2) // 24
        out = np.zeros(n * 68 * 36, dtype=np.uint64)
        assert self.lib.hostsim_g2_prepare(form, _ptr(g2), _ptr(out), ctypes.c_size_t(n)) == 0
        return out

    def witness_inverse(self, op, x):
        per = {0: 72, 1: 36, 2: 12}[op]
        n = len(x) // per
        out = np.zeros(n * per, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        assert self.lib.hostsim_witness_inverse(op, _ptr(x), _ptr(out), _ptr(flags, ctypes.c_uint8), ctypes.c_size_t(n)) == 0
        return out, flags

    def witness_sqrt(self, op, x, sgn):
        per = {0: 6, 1: 12}[op]
        n = len(x) // per
        out = np.zeros(n * per, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        sgn = np.ascontiguousarray(sgn, dtype=np.uint8)
        assert self.lib.hostsim_witness_sqrt(op, _ptr(x), _ptr(sgn, ctypes.c_uint8), _ptr(out), _ptr(flags, ctypes.c_uint8), ctypes.c_size_t(n)) == 0
        return out, flags

    def point_op(self, group, op, a, b=None, form=IO_ARK_MONT):
        w = 18 if group == 1 else 36
        n = len(a) // w
        out = np.zeros(n * w, dtype=np.uint64)
        flags = np.zeros(n, dtype=np.uint8)
        rc = self.lib.hostsim_point_op(group, op, form, _ptr(a), _ptr(b), _ptr(out), flags.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n))
        assert rc == 0
        return out, flags

    def line_ref_opt(self, q1, q2, p, form=IO_ARK_MONT):
        n = len(q1) // 36
        out = np.zeros(n * 24, dtype=np.uint64)
        rc = self.lib.hostsim_line_ref_opt(form, _ptr(q1), _ptr(q2), _ptr(p), _ptr(out), ctypes.c_size_t(n))
        assert rc == 0
        return out


# ------------------------------------------------------------------ golden fixtures
import json  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def hx(lst):
    return [int(x, 16) for x in lst]


from coracle import COracle  # noqa: E402,F401  (oracle/coracle.py: the C restatement, checker only)


def inversion_cases():
    """Values that steer the binary-GCD inversion (csrc/fp.hip.h fp_inv) through its corners: zero, one, p - 1, powers of two
    and their neighbours (long runs of halvings, approximation windows of every length), multiples near p, any 384-bit pattern."""
    import random
    rng = random.Random(2020972)
    vals = [0, 1, 2, 3, O.P - 1, O.P - 2, (O.P + 1) // 2, (O.P - 1) // 2, O.P, O.P + 1, 2 * O.P - 1, 2 ** 384 - 1, 2 ** 383, 2 ** 381]
    vals += [1 << k for k in range(1, 384, 7)] + [(1 << k) - 1 for k in range(2, 384, 5)] + [O.P - (1 << k) for k in range(1, 380, 11)]
    vals += [rng.randrange(1 << rng.randrange(1, 384)) for _ in range(300)] + [rng.randrange(O.P) for _ in range(300)]
    # bit length 30 t + 1: the top 30-bit limb of the first approximation window is exactly 1 (the un-clamped shift of fp_inv)
    vals += [(1 << (30 * t)) + r for t in range(3, 13) for r in (0, 1, rng.randrange(1 << (30 * t)))]
    return vals


def synthetic_batch(n, seed=O.SEED, jacobian=False, k=1, pool=64):
    """Small seeded batch in ABI layout for the oracle-sized tests: `pool` (x k) DISTINCT seeded pairs (multiples of the
    generators; Jacobian inputs with random representatives) tiled to n elements -- element i equals element i mod pool.
    Returns (g1, g2) uint64 arrays.  The full-size tests and bench.py use bls381_amd/synthetic.py instead (SURVEY 8d:
    4096-point pools built on the GPU, element i = (P[i mod 4096], Q[(i * 2654435761 >> 12) mod 4096]))."""
    base = random_pairs(pool * k, seed, jacobian=jacobian)
    if jacobian:
        g1 = enc_fps([c for P, Q in base for c in flat_g1_jac(P)])
        g2 = enc_fps([c for P, Q in base for c in flat_g2_jac(Q)])
        w1, w2 = 18, 36
    else:
        g1 = enc_fps([c for P, Q in base for c in flat_g1_aff(P)])
        g2 = enc_fps([c for P, Q in base for c in flat_g2_aff(Q)])
        w1, w2 = 12 * k, 24 * k
    reps = (n + pool - 1) // pool
    return np.tile(g1, reps)[: n * w1].copy(), np.tile(g2, reps)[: n * w2].copy()
