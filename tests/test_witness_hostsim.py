"""Witness-generator backend -- device routines compiled for the host vs the oracle."""
from blsutil import O, dec_fps, IO_CANONICAL
from witness_cases import cases, enc_can


def test_witness_inverses(hostsim):
    c = cases()
    for key, op, width in (("fq12", 0, 12), ("fq6", 1, 6), ("fq2", 2, 2)):
        xs, exp = c[key]
        out, fl = hostsim.witness_inverse(op, enc_can([v for x in xs for v in x]))
        got = dec_fps(out, IO_CANONICAL)
        if key == "fq2":
            assert [tuple(got[2 * i:2 * i + 2]) for i in range(len(xs))] == [tuple(e) for e in exp]
            assert not fl.any()
        else:
            assert [got[width * i:width * (i + 1)] for i in range(len(xs))] == [e[0] for e in exp], key
            assert list(fl) == [e[1] for e in exp], key


def test_witness_sqrts(hostsim):
    c = cases()
    xs, sg, exp = c["sqrt"]
    out, fl = hostsim.witness_sqrt(0, enc_can(xs), sg)
    assert dec_fps(out, IO_CANONICAL) == [e[0] for e in exp] and list(fl) == [e[1] for e in exp]
    assert {e[1] for e in exp} == {0, O.WFLAG_NOT_SQUARE, O.WFLAG_SIGN_UNSATISFIABLE}
    xs, sg, exp = c["sqrt2"]
    out, fl = hostsim.witness_sqrt(1, enc_can([v for x in xs for v in x]), sg)
    got = dec_fps(out, IO_CANONICAL)
    assert [tuple(got[2 * i:2 * i + 2]) for i in range(len(xs))] == [tuple(e[0]) for e in exp]
    assert list(fl) == [e[1] for e in exp]
    for (x, s, (r, f)) in zip(xs, sg, exp):                    # the oracle's roots really are roots with the asked sign
        if f == 0:
            assert O.fp2_sqr(r) == (x[0] % O.P, x[1] % O.P) and O.sgn0_fq2(r) == bool(s)


def test_g2_prepare_and_fp_div(hostsim):
    import random
    from blsutil import enc_fps, random_pairs, flat_g2_aff, rand_fp
    pts = [O.G2_GEN] + [q for _, q in random_pairs(3, 808, jacobian=False)]
    out = dec_fps(hostsim.g2_prepare(enc_fps([c for q in pts for c in flat_g2_aff(q)])))
    exp = [v for q in pts for (a, b, c) in O.g2_prepare(q) for v in (a[0], a[1], b[0], b[1], c[0], c[1])]
    assert out == exp and len(out) == len(pts) * 68 * 6
    rng = random.Random(3)
    xs = [rand_fp(rng) for _ in range(8)] + [5]; ys = [rand_fp(rng) for _ in range(8)] + [0]
    got = dec_fps(hostsim.fp_op(8, enc_fps(xs), enc_fps(ys)))
    assert got == [(x * pow(y, O.P - 2, O.P)) % O.P for x, y in zip(xs, ys)]


def _pow_cases():
    import random
    rng = random.Random(17)
    xs = [rng.randrange(1, O.P) for _ in range(6)] + [0, 1, O.P - 1]
    exps = [0, 1, 2, 3, 0xd201000000010000, (O.P - 1) // 2, O.P - 2, (1 << 128) - 1, 2 ** 200 + 12345]
    return xs, exps


def test_pow_fq_legendre_sgn0(hostsim):
    from blsutil import enc_fps
    xs, exps = _pow_cases()
    a = enc_fps(xs)
    for e in exps:
        words = [(e >> (64 * i)) & (2 ** 64 - 1) for i in range(max(1, (e.bit_length() + 63) // 64))]
        exp = [O.pow_fq(x, words) for x in xs]
        # the NAF walk equals plain exponentiation (and returns the base for exponent 0)
        assert [v for v, f in exp if f == 0] == [(x if e == 0 else pow(x, e, O.P)) for x, (v, f) in zip(xs, exp) if f == 0]
        got = dec_fps(hostsim.fp_pow(a, e))
        assert got == [v if f == 0 else (x if e == 0 else pow(x, e, O.P)) for x, (v, f) in zip(xs, exp)], hex(e)
    assert dec_fps(hostsim.fp_op(9, a)) == [O.legendre(x) for x in xs]
    assert dec_fps(hostsim.fp_op(10, a)) == [1 if O.legendre(x) == 1 else 0 for x in xs]
    assert dec_fps(hostsim.fp_op(11, a)) == [1 if O.sgn0_fq(x) else 0 for x in xs]
