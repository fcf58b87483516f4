"""Seeded cases for the witness-generator backend (shared by the host-simulation and GPU tests)."""
import random

import numpy as np
from blsutil import O, enc_fps, dec_fps, IO_CANONICAL, rand_fp, rand_fp2, rand_fp12


def cases(seed=2024, n=12):
    rng = random.Random(seed)
    c = {}
    x12 = [O.fp12_to_myfq12(rand_fp12(rng)) for _ in range(n)] + [[0] * 12, [1] + [0] * 11]
    c["fq12"] = (x12, [O.witness_fq12_inverse(x) for x in x12])
    x6 = [[rng.randrange(O.P) for _ in range(6)] for _ in range(n)] + [[0] * 6, [0, 0, 0, 0, 5, 0]]
    c["fq6"] = (x6, [O.witness_fq6_inverse(x) for x in x6])
    x2 = [rand_fp2(rng) for _ in range(n)] + [(0, 0), (1, 0), (0, 1), (O.P - 1, 0)]
    c["fq2"] = (x2, [O.witness_fq2_inverse(x) for x in x2])
    # square roots: squares (both signs requested), non-squares, zero with both signs
    xs, sg = [], []
    for _ in range(n):
        r = rand_fp(rng)
        xs += [(r * r) % O.P, (r * r) % O.P]; sg += [0, 1]
    nonsq = next(v for v in range(2, 50) if O.fp_sqrt_any(v) is None)
    xs += [nonsq, 0, 0, 1, 4]; sg += [0, 0, 1, 1, 0]
    c["sqrt"] = (xs, sg, [O.witness_fq_sqrt(x, s) for x, s in zip(xs, sg)])
    xs2, sg2 = [], []
    for _ in range(n):
        r = rand_fp2(rng)
        sq = O.fp2_sqr(r)
        xs2 += [sq, sq]; sg2 += [0, 1]
    # b == 0 branches: a square in Fp, a non-square in Fp (root is purely imaginary), zero, and a non-square of Fp2
    xs2 += [(4, 0), (nonsq, 0), ((-nonsq) % O.P, 0), (0, 0), (0, 0)]; sg2 += [1, 0, 1, 0, 1]
    ns2 = next((a, 1) for a in range(1, 60) if O.fp2_sqrt_any((a, 1)) is None)
    xs2 += [ns2, (0, 9), (0, 2)]; sg2 += [0, 1, 0]
    c["sqrt2"] = (xs2, sg2, [O.witness_fq2_sqrt(x, s) for x, s in zip(xs2, sg2)])
    return c


def enc_can(vals):
    """flat Fp ints -> uint64 canonical limbs (== 12 x u32 digits in memory)"""
    return enc_fps(vals, IO_CANONICAL)


def as_u32(a):
    return np.ascontiguousarray(a, dtype=np.uint64).view(np.uint32)
