"""Device algorithms compiled for the host (tests/hostsim) against the golden fixtures and the oracle, with
every static limb/value bound of the lazy-reduction types verified at run time (BLS_BOUND_CHECK).
This is how kernel logic is validated in the GPU-less container; the GPU parity tests proper are test_gpu_*.py."""
import random

import numpy as np
from blsutil import (O, enc_fps, dec_fps, enc_fp12s, dec_fp12s, load_golden, hx, rand_fp, rand_fp2, rand_fp12, inversion_cases,
                     random_pairs, flat_g1_jac, flat_g2_jac, IO_ARK_MONT, IO_CANONICAL)


def _fp12(h):
    return O.fp12_unflatten(hx(h))


def test_fp_ops_both_encodings(hostsim):
    rng = random.Random(7)
    xs = [rand_fp(rng) for _ in range(24)] + [0, 1, O.P - 1, O.P - 1]
    ys = [rand_fp(rng) for _ in range(24)] + [5, O.P - 1, O.P - 1, 1]
    for form in (IO_ARK_MONT, IO_CANONICAL):
        a, b = enc_fps(xs, form), enc_fps(ys, form)
        assert dec_fps(hostsim.fp_op(0, a, b, form), form) == [(x + y) % O.P for x, y in zip(xs, ys)]
        assert dec_fps(hostsim.fp_op(1, a, b, form), form) == [(x - y) % O.P for x, y in zip(xs, ys)]
        assert dec_fps(hostsim.fp_op(2, a, b, form), form) == [(x * y) % O.P for x, y in zip(xs, ys)]
        assert dec_fps(hostsim.fp_op(3, a, None, form), form) == [(x * x) % O.P for x in xs]
        assert dec_fps(hostsim.fp_op(5, a, None, form), form) == [(-x) % O.P for x in xs]
        assert dec_fps(hostsim.fp_op(4, a, None, form), form) == [pow(x, O.P - 2, O.P) for x in xs]
    # to/from Montgomery (KAT-3 shape): canonical -> ark form and back
    can = enc_fps(xs, IO_CANONICAL)
    mont = hostsim.fp_op(6, can)
    assert np.array_equal(mont, enc_fps(xs, IO_ARK_MONT))
    assert np.array_equal(hostsim.fp_op(7, mont), can)


def test_unreduced_inputs_are_reduced(hostsim):
    """Inputs >= p (any 384-bit pattern) are accepted and reduced, like a `from_biguint`-style load."""
    vals = [O.P, O.P + 1, 2 ** 384 - 1, 2 * O.P + 5]
    a = np.array([(v >> (64 * i)) & (2 ** 64 - 1) for v in vals for i in range(6)], dtype=np.uint64)
    one = enc_fps([1] * len(vals), IO_CANONICAL)
    out = dec_fps(hostsim.fp_op(2, a, one, IO_CANONICAL), IO_CANONICAL)
    assert out == [v % O.P for v in vals]


def test_fp_inverse_binary_gcd_against_fermat_and_pow(hostsim):
    vals = inversion_cases()
    a = np.array([(v >> (64 * i)) & (2 ** 64 - 1) for v in vals for i in range(6)], dtype=np.uint64)
    got = dec_fps(hostsim.fp_op(4, a, None, IO_CANONICAL), IO_CANONICAL)
    assert got == [pow(v % O.P, O.P - 2, O.P) for v in vals]
    assert dec_fps(hostsim.fp_op(12, a, None, IO_CANONICAL), IO_CANONICAL) == got      # a^(p-2) by square-and-multiply


def test_fp2_ops(hostsim):
    rng = random.Random(8)
    a2 = [rand_fp2(rng) for _ in range(8)] + [(0, 0), (1, 0), (0, 1)]
    b2 = [rand_fp2(rng) for _ in range(8)] + [(5, 7), (O.P - 1, 3), (O.P - 1, O.P - 1)]
    A = enc_fps([c for v in a2 for c in v]); B = enc_fps([c for v in b2 for c in v])
    pair = lambda r: [(r[2 * i], r[2 * i + 1]) for i in range(len(r) // 2)]
    assert pair(dec_fps(hostsim.fp2_op(0, A, B))) == [O.fp2_mul(x, y) for x, y in zip(a2, b2)]
    assert pair(dec_fps(hostsim.fp2_op(1, A))) == [O.fp2_sqr(x) for x in a2]
    assert pair(dec_fps(hostsim.fp2_op(3, A))) == [O.fp2_mul_by_nonresidue(x) for x in a2]
    assert pair(dec_fps(hostsim.fp2_op(4, A))) == [O.fp2_conj(x) for x in a2]
    inv = pair(dec_fps(hostsim.fp2_op(2, A)))
    assert inv == [O.fp2_inv(x) or (0, 0) for x in a2]


def test_tower_golden(hostsim):
    cases = load_golden("tower_ops.json")["cases"]
    X = enc_fp12s([_fp12(c["x"]) for c in cases]); Y = enc_fp12s([_fp12(c["y"]) for c in cases])
    for op, key in ((0, "mul"), (1, "sqr"), (2, "inv"), (3, "frobenius"), (4, "cyclotomic_square"), (5, "conj"), (7, "final_exp")):
        assert dec_fp12s(hostsim.fp12_op(op, X, Y)) == [_fp12(c[key]) for c in cases], key
    assert dec_fp12s(hostsim.fp12_op(6, X)) == [O.cyclotomic_exp(_fp12(c["x"])) for c in cases]


def test_mul_by_014_lazy_sums_and_karatsuba(hostsim):
    """The sparse line multiplication in the two forms of csrc/tower.hip.h -- six three-product sums with one reduction each
    (what the Miller kernels compile) and Karatsuba over Fp6 -- against the oracle's mul_by_014
    (fields_as_trees/fq12_target_tree.rs:157-176), bound checks on."""
    rng = random.Random(14)
    fs = [rand_fp12(rng) for _ in range(9)]
    ys = [rand_fp12(rng) for _ in range(9)]
    ys[3] = O.fp12_unflatten([0] * 12)
    fs[5] = O.fp12_unflatten([O.P - 1] * 12)
    X = enc_fp12s(fs); Y = enc_fp12s(ys)
    exp = [O.fp12_mul_by_014(f, y[0][0], y[0][1], y[1][1]) for f, y in zip(fs, ys)]
    assert dec_fp12s(hostsim.fp12_op(10, X, Y)) == exp
    assert dec_fp12s(hostsim.fp12_op(11, X, Y)) == exp


def test_twice_square_sqr3_over_fp4(hostsim):
    """sqr_twice (tower.hip.h): 2 f^2 by the Chung-Hasan SQR3 formula over Fp4 (the A/B alternative for the squaring of the
    homogeneous-step Miller loops, BLS_HOM_SQR3) against the big-integer tower: random elements, a sparse one (a line value),
    zero, one and all-(p-1)."""
    rng = random.Random(3303)
    xs = [rand_fp12(rng) for _ in range(40)]
    xs += [O.FP12_ONE, O.fp12_unflatten([0] * 12), O.fp12_unflatten([O.P - 1] * 12)]
    line = [[rand_fp2(rng), rand_fp2(rng), (0, 0)], [(0, 0), rand_fp2(rng), (0, 0)]]
    xs.append(tuple(tuple(c) for c in line))
    got = dec_fp12s(hostsim.fp12_op(13, enc_fp12s(xs)))
    for x, g in zip(xs, got):
        assert g == O.fp12_mul(O.fp12_sqr(x), O.fp12_from_fp2((2, 0)))


def test_final_exp_split_with_batched_inversion(hostsim):
    """The three-kernel form of the final exponentiation (norms -> simultaneous inversion in chunks of eight -> finish)
    gives the oracle's bits for every element, whatever the chunk alignment; zero elements stay zero and do not disturb
    the others of their chunk."""
    rng = random.Random(77)
    vals = [rand_fp12(rng) for _ in range(11)]          # 8 + a ragged tail of 3
    vals[2] = O.fp12_unflatten([0] * 12)                # zero: no inverse (the reference would panic); stays zero
    X = enc_fp12s(vals)
    exp = [O.final_exponentiation(v) or vals[2] for v in vals]
    assert dec_fp12s(hostsim.fp12_op(8, X)) == exp
    assert dec_fp12s(hostsim.fp12_op(7, X)) == exp
    for n in (1, 7, 8, 9):
        assert dec_fp12s(hostsim.fp12_op(8, X[:72 * n])) == exp[:n], n


def test_final_exp_staged_with_compressed_squarings(hostsim):
    """The six-stage final exponentiation (Karabina-compressed squaring chains, their decompression denominators inverted
    batch-wide between the stages; csrc/fexp_stages.hip.h) gives the oracle's bits: random Fq12 inputs, the identity (whose
    compressed form is all zero: the zero-denominator path), an element of GT, and a zero input (stays zero)."""
    rng = random.Random(78)
    vals = [rand_fp12(rng) for _ in range(10)]
    zero = O.fp12_unflatten([0] * 12)
    vals[1] = O.fp12_unflatten([1] + [0] * 11)          # one
    vals[4] = zero
    from blsutil import point_pool
    p1, p2 = point_pool(4)
    vals[6] = O.miller_loop(p1[1], p2[2])               # a genuine Miller value
    vals[7] = O.pairing(p1[0], p2[0])                   # already in GT
    X = enc_fp12s(vals)
    exp = [O.final_exponentiation(v) or zero for v in vals]
    assert dec_fp12s(hostsim.fp12_op(9, X)) == exp
    assert dec_fp12s(hostsim.fp12_op(12, X)) == exp          # glue and chain as separate passes (the default launch)
    for n in (1, 3):
        assert dec_fp12s(hostsim.fp12_op(9, X[:72 * n])) == exp[:n], n


def test_mode_a_golden_both_variants(hostsim):
    cases = load_golden("mode_a.json")["cases"]
    g1 = enc_fps([v for c in cases for v in hx(c["g1_jac"])])
    g2 = enc_fps([v for c in cases for v in hx(c["g2_jac"])])
    for generic in (False, True):
        out, fl = hostsim.miller_ref_opt(g1, g2, generic=generic)
        assert dec_fp12s(out) == [_fp12(c["out"]) for c in cases], generic
        assert list(fl) == [c["flags"] for c in cases]


def test_mode_a_lds_form_golden(hostsim):
    """miller_ref_opt_generic_lds -- what k_miller_ref_opt_wide runs: one Fq12 accumulator in registers, the other in the slot store,
    squaring formula, Fp2-scalar line multiplication, R + R by the doubling formulas -- on the golden cases (infinity, garbage and
    all-zero inputs included), with bound checks."""
    cases = load_golden("mode_a.json")["cases"]
    g1 = enc_fps([v for c in cases for v in hx(c["g1_jac"])])
    g2 = enc_fps([v for c in cases for v in hx(c["g2_jac"])])
    out, fl = hostsim.miller_ref_opt_lds(g1, g2)
    assert dec_fp12s(out) == [_fp12(c["out"]) for c in cases]
    assert list(fl) == [c["flags"] for c in cases]


def test_multi_pair_lds_form_against_the_oracle(hostsim):
    """miller_ate_lds -- what k_multi_miller<2|3> run: the idle pairs' R in the slot store, changing places with the current pair's
    after every step -- against the C oracle: Miller products (Alg. 26/27 steps) and full pairing products (homogeneous steps),
    with a pair at infinity in every position."""
    from blsutil import COracle, synthetic_batch
    orc = COracle()
    for k in (2, 3):
        n = 6
        g1, g2 = synthetic_batch(n, seed=700 + k, jacobian=False, k=k, pool=n)
        g1 = g1.copy(); g2 = g2.copy()
        for j in range(k):                                   # element j: pair j at infinity (G1 side, then G2 side on the last element)
            g1.reshape(n, k, 12)[j, j] = 0
        g2.reshape(n, k, 24)[n - 1, k - 1] = 0
        for fe in (False, True):
            exp, efl = orc.multi(g1, g2, k, final_exp=fe, threads=4)
            out, fl = hostsim.multi_miller_lds(g1, g2, k, final_exp=fe)
            assert np.array_equal(out, exp) and np.array_equal(fl, efl), (k, fe)
            assert fl[:k].all() and fl[n - 1]


def test_prepared_pairing_loop_against_the_oracle(hostsim):
    """miller_prepared (pairing.hip.h) -- the loop of k_prepared_miller: the variable pair walks its curve steps, the fixed pairs take
    G2Prepared coefficients (here from a plain array, on the device from LDS) -- for the Groth16 shape (1, 2) and the shapes
    (1, 1), (0, 1), (0, 2), homogeneous and Alg. 26/27 steps, against the generic product of the C oracle; a G1 point at infinity
    in a fixed slot and the variable G2 point at infinity included."""
    from blsutil import COracle, synthetic_batch
    orc = COracle()
    n = 4
    g1, g2 = synthetic_batch(n, seed=4100, jacobian=False, k=3, pool=n)
    g1 = g1.copy().reshape(n, 3, 12); g2 = g2.copy().reshape(n, 3, 24)
    g2[:, 1] = g2[0, 1]; g2[:, 2] = g2[0, 2]                       # pairs 1 and 2: the same G2 points on every element
    g1[1, 2] = 0                                                    # P at infinity in a fixed slot
    g2[2, 0] = 0                                                    # the variable Q at infinity
    prep = hostsim.g2_prepare(np.ascontiguousarray(g2[0, 1:3]).reshape(-1))
    for n_var, n_fixed in ((1, 2), (1, 1), (0, 1), (0, 2)):
        cols = list(range(0 if n_var else 1, 1 + n_fixed))         # which of the three pairs take part
        a = np.ascontiguousarray(g1[:, cols]).reshape(-1); b = np.ascontiguousarray(g2[:, cols]).reshape(-1)
        exp, efl = orc.multi(a, b, len(cols), final_exp=True, threads=4)
        pc = prep[: n_fixed * 68 * 36]
        for hom in (True, False):
            out, fl = hostsim.prepared_pairing(a, np.ascontiguousarray(g2[:, 0]).reshape(-1), pc, n_var, n_fixed, hom=hom)
            assert np.array_equal(out, exp) and np.array_equal(fl, efl), (n_var, n_fixed, hom)


def test_line_function_three_branches(hostsim):
    cases = load_golden("line_fn.json")["cases"]
    assert sorted(c["branch"] for c in cases) == ["A", "T", "V"]
    q1 = enc_fps([v for c in cases for v in hx(c["q1"])]); q2 = enc_fps([v for c in cases for v in hx(c["q2"])])
    p = enc_fps([v for c in cases for v in hx(c["p"])])
    out = dec_fps(hostsim.line_ref_opt(q1, q2, p))
    exp = [v for c in cases for v in hx(c["n"]) + hx(c["d"])]
    assert out == exp


def test_mode_b_golden(hostsim):
    g = load_golden("mode_b.json")
    s = g["single"]
    g1 = enc_fps([v for c in s for v in hx(c["g1"])]); g2 = enc_fps([v for c in s for v in hx(c["g2"])])
    out, fl = hostsim.multi_miller(g1, g2, 1, final_exp=False)
    assert dec_fp12s(out) == [_fp12(c["miller"]) for c in s] and list(fl) == [c["flags"] for c in s]
    out, fl = hostsim.multi_miller(g1, g2, 1, final_exp=True)
    assert dec_fp12s(out) == [_fp12(c["pairing"]) for c in s]
    m = g["multi"]
    g1 = enc_fps([v for c in m for p in c["g1"] for v in hx(p)]); g2 = enc_fps([v for c in m for q in c["g2"] for v in hx(q)])
    out, fl = hostsim.multi_miller(g1, g2, 3, final_exp=True)
    assert dec_fp12s(out) == [_fp12(c["pairing"]) for c in m] and list(fl) == [c["flags"] for c in m]


def test_canonical_io_form(hostsim):
    c = load_golden("mode_a.json")["cases"][0]
    g1 = enc_fps(hx(c["g1_jac"]), IO_CANONICAL); g2 = enc_fps(hx(c["g2_jac"]), IO_CANONICAL)
    out, _ = hostsim.miller_ref_opt(g1, g2, generic=False, form=IO_CANONICAL)
    assert dec_fp12s(out, IO_CANONICAL) == [_fp12(c["out"])]


def test_bound_checks_ran(hostsim):
    assert hostsim.lib.hostsim_bound_checks() > 100000


def test_point_surface_g1_g2(hostsim):
    """Jacobian add / double / normalise / equality / on-curve for G1 and G2 (A4 of SURVEY 8a: the arkworks group law at
    opt.rs:93,98, with its equal-operand, opposite-operand and infinity branches) against the oracle, bit for bit."""
    import point_cases

    def run(group, op, a, b):
        out, fl = hostsim.point_op(group, op, a, b)
        return out, fl
    for group in (1, 2):
        point_cases.check(run, group)
