"""Shared cases for the point surface (bls381_g1_op_batch / bls381_g2_op_batch): inputs and the oracle's expected
outputs, used by the host-simulation test (CPU) and the GPU parity test.  Covers the group law's fallbacks as arkworks
implements them (equal operands -> doubling, opposite operands -> infinity, infinity on either side) with general-Z
representatives, normalisation, equality across representatives and curve membership."""
import random

from blsutil import O, enc_fps, dec_fps, point_pool, rand_fp

ADD, DOUBLE, NORMALIZE, EQ, ON_CURVE, IN_SUBGROUP = range(6)
FLAG_INFINITY, FLAG_TRUE = 2, 32


def _field(group):
    return O._F1 if group == 1 else O._F2


def _flat(group, pt):
    return list(pt) if group == 1 else [c for co in pt for c in co]


def _unflat(group, v):
    if group == 1:
        return [tuple(v[3 * i:3 * i + 3]) for i in range(len(v) // 3)]
    return [tuple((v[6 * i + 2 * j], v[6 * i + 2 * j + 1]) for j in range(3)) for i in range(len(v) // 6)]


def _rand_el(group, rng):
    return rand_fp(rng) if group == 1 else (rand_fp(rng), rand_fp(rng))


def build(group, seed=2024):
    """-> dict op -> (a_points, b_points) as lists of Jacobian tuples"""
    F = _field(group)
    rng = random.Random(seed + group)
    pool = point_pool(16)[group - 1]
    jac = lambda k: O.jac_rerandomise(F, O.jac_from_affine(F, pool[k]), _rand_el(group, rng))
    neg = lambda pt: (pt[0], F.neg(pt[1]), pt[2])
    inf = (_rand_el(group, rng), _rand_el(group, rng), F.zero)            # any (X, Y, 0) is the point at infinity
    inf2 = (F.one, F.one, F.zero)
    a = [jac(0), jac(3), jac(5), jac(7), inf, jac(9), inf, O.jac_from_affine(F, pool[2])]
    b = [jac(1), jac(3), neg(a[2]), neg(jac(7)), jac(4), inf2, inf2, O.jac_from_affine(F, pool[6])]
    bad = (a[0][0], F.add(a[0][1], F.one), a[0][2])                       # off the curve
    return {
        IN_SUBGROUP: ([a[0], bad, inf, jac(13), outside_subgroup(group, 1), O.jac_rerandomise(F, outside_subgroup(group, 2), _rand_el(group, rng))], None),
        ADD: (a, b),
        DOUBLE: ([jac(2), jac(11), inf, O.jac_from_affine(F, pool[1])], None),
        NORMALIZE: ([jac(4), inf, O.jac_from_affine(F, pool[8]), jac(15)], None),
        EQ: ([a[1], a[0], inf, inf, a[3], a[2]], [b[1], b[0], inf2, jac(1), neg(b[3]), b[2]]),
        ON_CURVE: ([a[0], bad, inf, jac(12), O.jac_from_affine(F, pool[0])], None),
    }


def outside_subgroup(group, which):
    """A point ON the curve that is NOT in the order-r subgroup (the cofactors are far from 1, so the first abscissa tried
    works): smallest x = which, which + 1, ... with x^3 + b a square; checked with the oracle's scalar multiplication."""
    F = _field(group)
    x = which
    while True:
        xx = x if group == 1 else (x, 1)
        pt, fl = O.decompress_point(group, xx, 0)
        if pt is not None:
            j = O.jac_from_affine(F, pt)
            assert O.jac_on_curve(F, j)
            if not F.is_zero(O.jac_scalar_mul(F, j, O.R_ORDER)[2]):
                return j
        x += 1


def expected(group, op, a, b):
    """-> (points or None, flags)"""
    F = _field(group)
    if op == ADD:
        r = [O.jac_add(F, x, y) for x, y in zip(a, b)]
    elif op == DOUBLE:
        r = [O.jac_double(F, x) for x in a]
    elif op == NORMALIZE:
        r = [O.jac_normalize(F, x) for x in a]
    elif op == EQ:
        return None, [(FLAG_INFINITY if F.is_zero(x[2]) else 0) | (FLAG_TRUE if O.jac_eq(F, x, y) else 0) for x, y in zip(a, b)]
    elif op == IN_SUBGROUP:
        return None, [(FLAG_INFINITY if F.is_zero(x[2]) else 0) | (FLAG_TRUE if O.jac_in_subgroup(F, x) else 0) for x in a]
    else:
        return None, [(FLAG_INFINITY if F.is_zero(x[2]) else 0) | (FLAG_TRUE if O.jac_on_curve(F, x) else 0) for x in a]
    return r, [FLAG_INFINITY if F.is_zero(p[2]) else 0 for p in r]


def encode(group, pts):
    return enc_fps([c for p in pts for c in _flat(group, p)])


def decode(group, arr):
    return _unflat(group, dec_fps(arr))


def check(run, group):
    """run(group, op, a_encoded, b_encoded_or_None) -> (out array or None, flags array); asserts oracle parity for every op"""
    cases = build(group)
    seen = set()
    for op, (a, b) in cases.items():
        out, fl = run(group, op, encode(group, a), encode(group, b) if b is not None else None)
        exp_pts, exp_fl = expected(group, op, a, b)
        assert list(fl) == exp_fl, (group, op, list(fl), exp_fl)
        if exp_pts is not None:
            got = decode(group, out)
            assert got == [tuple(p) for p in exp_pts], (group, op)
        seen.update(exp_fl)
    assert {0, FLAG_INFINITY, FLAG_TRUE, FLAG_INFINITY | FLAG_TRUE} <= seen      # every flag combination exercised
