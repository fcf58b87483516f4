"""GPU parity tests proper: the HIP path, called through the C ABI (libbls381_hip.so via ctypes), against the
oracle and the committed golden fixtures -- bit-exact (integer arithmetic).  `pytest -m gpu` on an MI355X."""
import os
import random

import numpy as np
import pytest

from blsutil import (O, enc_fps, dec_fps, enc_fp12s, dec_fp12s, load_golden, hx, rand_fp, rand_fp2, rand_fp12, random_pairs,
                     flat_g1_jac, flat_g2_jac, flat_g1_aff, flat_g2_aff, synthetic_batch, IO_ARK_MONT, IO_CANONICAL)
import bls381_amd as B

pytestmark = pytest.mark.gpu


def _fp12(h):
    return O.fp12_unflatten(hx(h))


def test_fp_ops_and_kat3_montgomery_conversion(engine):
    rng = random.Random(7)
    xs = [rand_fp(rng) for _ in range(300)] + [0, 1, O.P - 1, O.P - 1]
    ys = [rand_fp(rng) for _ in range(300)] + [5, O.P - 1, O.P - 1, 1]
    a, b = enc_fps(xs), enc_fps(ys)
    assert dec_fps(engine.fp_op(B.FP_ADD, a, b)) == [(x + y) % O.P for x, y in zip(xs, ys)]
    assert dec_fps(engine.fp_op(B.FP_SUB, a, b)) == [(x - y) % O.P for x, y in zip(xs, ys)]
    assert dec_fps(engine.fp_op(B.FP_MUL, a, b)) == [(x * y) % O.P for x, y in zip(xs, ys)]
    assert dec_fps(engine.fp_op(B.FP_SQR, a)) == [(x * x) % O.P for x in xs]
    assert dec_fps(engine.fp_op(B.FP_NEG, a)) == [(-x) % O.P for x in xs]
    assert dec_fps(engine.fp_op(B.FP_INV, a)) == [pow(x, O.P - 2, O.P) for x in xs]
    # KAT-3: the reference's hard-coded generator limbs are to_mont(canonical generator coordinates)
    k = load_golden("reference_kats.json")["kat3_generators_montgomery"]
    gen = [O.G1_GEN[0], O.G1_GEN[1], *O.G2_GEN[0], *O.G2_GEN[1]]
    mont = engine.fp_op(B.FP_TO_MONT, enc_fps(gen, IO_CANONICAL)).reshape(-1, 6)
    got = [sum(int(l) << (64 * i) for i, l in enumerate(row)) for row in mont]
    assert got == [k["g1_x"], k["g1_y"], *k["g2_x"], *k["g2_y"]]
    assert np.array_equal(engine.fp_op(B.FP_FROM_MONT, mont.reshape(-1)), enc_fps(gen, IO_CANONICAL))


def test_fp_inverse_corner_cases_on_device(engine):
    """bls381_fp_op_batch(INV) -- the binary-GCD inversion of csrc/fp.hip.h -- on zero, one, p - 1, powers of two and their
    neighbours, values near p and arbitrary 384-bit patterns (canonical I/O), against pow(x, p - 2, p); division shares it."""
    from blsutil import inversion_cases
    vals = inversion_cases()
    a = np.array([(v >> (64 * i)) & (2 ** 64 - 1) for v in vals for i in range(6)], dtype=np.uint64)
    engine.set_option(B.OPT_IO_FORM, IO_CANONICAL)
    try:
        got = dec_fps(engine.fp_op(B.FP_INV, a), IO_CANONICAL)
        one = enc_fps([1] * len(vals), IO_CANONICAL)
        div = dec_fps(engine.fp_op(B.FP_DIV, one, a), IO_CANONICAL)
    finally:
        engine.set_option(B.OPT_IO_FORM, IO_ARK_MONT)
    exp = [pow(v % O.P, O.P - 2, O.P) for v in vals]
    assert got == exp and div == exp


def test_kat1_kat2_on_device(engine):
    K = load_golden("reference_kats.json")
    engine.set_option(B.OPT_IO_FORM, IO_CANONICAL)     # the reference feeds raw limbs to a canonical-limb type
    try:
        for name, op in (("kat1_fp2_add", B.FP_ADD), ("kat2_fp2_sub", B.FP_SUB)):
            k = K[name]
            out = dec_fps(engine.fp_op(op, enc_fps(k["a"], IO_CANONICAL), enc_fps(k["b"], IO_CANONICAL)), IO_CANONICAL)
            assert out == k["c"], name
    finally:
        engine.set_option(B.OPT_IO_FORM, IO_ARK_MONT)


def test_tower_ops_golden(engine):
    cases = load_golden("tower_ops.json")["cases"]
    X = enc_fp12s([_fp12(c["x"]) for c in cases]); Y = enc_fp12s([_fp12(c["y"]) for c in cases])
    for op, key in ((B.FP12_MUL, "mul"), (B.FP12_SQR, "sqr"), (B.FP12_INV, "inv"), (B.FP12_FROBENIUS, "frobenius"),
                    (B.FP12_CYCLOTOMIC_SQR, "cyclotomic_square"), (B.FP12_CONJ, "conj")):
        assert dec_fp12s(engine.fp12_op(op, X, Y)) == [_fp12(c[key]) for c in cases], key
    assert dec_fp12s(engine.final_exponentiation(X)) == [_fp12(c["final_exp"]) for c in cases]
    assert dec_fp12s(engine.fp12_op(B.FP12_CYCLOTOMIC_EXP, X)) == [O.cyclotomic_exp(_fp12(c["x"])) for c in cases]


def test_sparse_line_multiplication_both_forms(engine):
    """mul_by_014 (fields_as_trees/fq12_target_tree.rs:157-176) as the Miller kernels run it -- six lazy three-product sums with one
    reduction each -- and as Karatsuba over Fq6, against the oracle; zero, all-(p-1) and random operands, a ragged batch."""
    import random
    from blsutil import rand_fp12
    rng = random.Random(1414)
    n = 133
    fs = [rand_fp12(rng) for _ in range(n)]
    ys = [rand_fp12(rng) for _ in range(n)]
    ys[3] = O.fp12_unflatten([0] * 12); fs[4] = O.fp12_unflatten([0] * 12)
    fs[5] = O.fp12_unflatten([O.P - 1] * 12); ys[5] = O.fp12_unflatten([O.P - 1] * 12)
    X = enc_fp12s(fs); Y = enc_fp12s(ys)
    exp = [O.fp12_mul_by_014(f, y[0][0], y[0][1], y[1][1]) for f, y in zip(fs, ys)]
    assert dec_fp12s(engine.fp12_op(B.FP12_MUL_BY_014, X, Y)) == exp
    assert dec_fp12s(engine.fp12_op(B.FP12_MUL_BY_014_KARATSUBA, X, Y)) == exp


def test_final_exp_modes_agree_with_the_oracle(engine):
    """BLS381_OPT_FEXP_MODE: the self-contained kernel, the split form (norm kernel, simultaneous inversion eight per lane,
    finishing kernel) and the staged form (Karabina-compressed squaring chains, six stages) give the oracle's bits, for
    ragged batch sizes (chunk tails, context scratch growing between calls), with zero elements inside a chunk (zero
    stays zero, neighbours intact) and with the identity (all-zero compressed form: the zero-denominator path)."""
    rng = random.Random(401)
    vals = [rand_fp12(rng) for _ in range(21)]
    zero = O.fp12_unflatten([0] * 12)
    vals[3] = zero; vals[16] = zero
    vals[5] = O.fp12_unflatten([1] + [0] * 11)
    exp = [O.final_exponentiation(v) or zero for v in vals]
    X = enc_fp12s(vals)
    default = engine.get_option(B.OPT_FEXP_MODE)
    assert default == B.FEXP_AUTO
    try:
        for mode in (B.FEXP_STAGED, B.FEXP_BATCHED_INVERSION, B.FEXP_SELF_CONTAINED):
            engine.set_option(B.OPT_FEXP_MODE, mode)
            for n in (1, 7, 8, 9, 21):
                assert dec_fp12s(engine.final_exponentiation(X[: 72 * n])) == exp[:n], (mode, n)
        with pytest.raises(B.Bls381Error):
            engine.set_option(B.OPT_FEXP_MODE, 4)
    finally:
        engine.set_option(B.OPT_FEXP_MODE, default)


def test_pairing_same_bits_in_every_fexp_mode(engine, coracle):
    n = 300                                                               # not a multiple of the chunk or the wave
    g1, g2 = synthetic_batch(n, jacobian=False, pool=16)
    exp, efl = coracle.multi(g1, g2, 1, final_exp=True, threads=16)
    default = engine.get_option(B.OPT_FEXP_MODE)
    try:
        for mode in (B.FEXP_STAGED, B.FEXP_BATCHED_INVERSION, B.FEXP_SELF_CONTAINED):
            engine.set_option(B.OPT_FEXP_MODE, mode)
            out, fl = engine.multi_pairing(g1, g2, 1)
            assert np.array_equal(out, exp) and np.array_equal(fl, efl), mode
            for blk in (128, 256):                                        # the lane-sliced scratch follows the launch shape
                engine.set_option(B.OPT_BLOCK_SIZE, blk)
                out, fl = engine.multi_pairing(g1, g2, 1)
                assert np.array_equal(out, exp), (mode, blk)
            engine.set_option(B.OPT_BLOCK_SIZE, 64)
    finally:
        engine.set_option(B.OPT_FEXP_MODE, default); engine.set_option(B.OPT_BLOCK_SIZE, 64)


@pytest.mark.parametrize("pipes", [1, 2, 3, 4])
def test_staged_fexp_pipelines(engine, coracle, pipes):
    """Staged final exponentiation as 1..4 part-batch pipelines on the context's internal streams (fork/join around the
    caller's stream, one share of the scratch each): a ragged batch just above the pipelining threshold, against the oracle."""
    n = (1 << 14) + 77
    g1, g2 = synthetic_batch(n, seed=600 + pipes, jacobian=False, pool=64)
    exp, efl = coracle.multi(g1, g2, 1, final_exp=True, threads=16)
    default = engine.get_option(B.OPT_FEXP_PIPELINES)
    try:
        engine.set_option(B.OPT_FEXP_MODE, B.FEXP_STAGED)
        engine.set_option(B.OPT_FEXP_PIPELINES, pipes)
        out, fl = engine.multi_pairing(g1, g2, 1)
        assert np.array_equal(out, exp) and np.array_equal(fl, efl)
    finally:
        engine.set_option(B.OPT_FEXP_PIPELINES, default)


def test_staged_fexp_fused_stage_launches(coracle, tmp_path):
    """The A/B launch forms -- stages 1..5 of the staged final exponentiation as fused two-wave kernels (BLS381_FEXP_FUSED_STAGES),
    the two-wave stage 6 (BLS381_NARROW_STAGE6), the two-wave generic Mode-A kernel (BLS381_NARROW_REF_OPT); read from the
    environment at the first such launch of a process -- in a fresh process, against the oracle."""
    import subprocess, sys
    n = (1 << 14) + 77
    g1, g2 = synthetic_batch(n, seed=4242, jacobian=False, pool=64)
    exp, efl = coracle.multi(g1, g2, 1, final_exp=True, threads=16)
    j1, j2 = synthetic_batch(301, seed=4243, jacobian=True, pool=43)
    expa, efla = coracle.miller_ref_opt(j1, j2, threads=16)
    for name, arr in (("g1", g1), ("g2", g2), ("j1", j1), ("j2", j2)):
        np.save(tmp_path / (name + ".npy"), arr)
    t = lambda name: str(tmp_path / (name + ".npy"))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import bls381_amd as B; e = B.Engine(0); "
            "e.set_option(B.OPT_FEXP_MODE, B.FEXP_STAGED); "
            "o, f = e.multi_pairing(np.load(%r), np.load(%r), 1); np.save(%r, o); np.save(%r, f); "
            "o, f = e.optimized_miller_loop(np.load(%r), np.load(%r)); np.save(%r, o); np.save(%r, f)"
            % (os.path.dirname(os.path.dirname(B.__file__)), t("g1"), t("g2"), t("out"), t("fl"), t("j1"), t("j2"), t("outa"), t("fla")))
    env = dict(os.environ, BLS381_FEXP_FUSED_STAGES="1", BLS381_NARROW_STAGE6="1", BLS381_NARROW_REF_OPT="1")
    subprocess.check_call([sys.executable, "-c", code], env=env)
    assert np.array_equal(np.load(t("out")), exp) and np.array_equal(np.load(t("fl")), efl)
    assert np.array_equal(np.load(t("outa")), expa) and np.array_equal(np.load(t("fla")), efla)


def test_homogeneous_miller_option_domain(engine, coracle):
    """BLS381_OPT_PAIRING_HOMOGENEOUS_MILLER: on curve points the pairing has the same bits either way (the lines differ by
    Fp2 factors the final exponentiation kills); on inputs that are NOT on the curves only the Alg. 26/27 path is the
    reference's arithmetic -- documented, and pinned here so that it stays true."""
    n = 64
    g1, g2 = synthetic_batch(n, seed=808, jacobian=False, pool=32)
    exp, _ = coracle.multi(g1, g2, 1, final_exp=True, threads=8)
    rs = np.random.default_rng(5)
    a = np.frombuffer(rs.bytes(n * 96), dtype=np.uint64).copy(); a[5::6] &= np.uint64(0x0fffffffffffffff)
    b = np.frombuffer(rs.bytes(n * 192), dtype=np.uint64).copy(); b[5::6] &= np.uint64(0x0fffffffffffffff)
    gexp, _ = coracle.multi(a, b, 1, final_exp=True, threads=8)
    assert engine.get_option(B.OPT_PAIRING_HOMOGENEOUS_MILLER) == 1
    try:
        for hom in (1, 0):
            engine.set_option(B.OPT_PAIRING_HOMOGENEOUS_MILLER, hom)
            assert np.array_equal(engine.multi_pairing(g1, g2, 1)[0], exp), hom
        assert np.array_equal(engine.multi_pairing(a, b, 1)[0], gexp)             # Alg. 26/27 on garbage == oracle
        assert np.array_equal(engine.multi_miller_loop(a, b, 1)[0], coracle.multi(a, b, 1, final_exp=False, threads=8)[0])
    finally:
        engine.set_option(B.OPT_PAIRING_HOMOGENEOUS_MILLER, 1)


def test_point_surface_g1_g2(engine):
    """bls381_g1_op_batch / bls381_g2_op_batch against the oracle (same cases as the host simulation): the arkworks group
    law of opt.rs:93,98 with its fallbacks, normalisation, equality across representatives, curve membership."""
    import point_cases
    for group in (1, 2):
        point_cases.check(engine.point_op, group)
    # argument checking: predicates need flags, ADD needs b (through the raw ABI)
    import ctypes
    a = point_cases.encode(2, point_cases.build(2)[point_cases.DOUBLE][0])
    pa = a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    assert engine.lib.bls381_g2_op_batch(engine._ctx, B.POINT_ADD, pa, None, pa, None, 4) == 1          # BLS381_E_INVALID
    assert engine.lib.bls381_g2_op_batch(engine._ctx, B.POINT_IS_ON_CURVE, pa, None, None, None, 4) == 1
    assert engine.lib.bls381_g2_op_batch(engine._ctx, 9, pa, None, pa, None, 4) == 1


def test_point_decompression(engine):
    """bls381_g1/g2_decompress_batch vs the oracle: both signs of pool points, the infinity marker, abscissas off the curve
    (flagged, zero output) and an on-curve abscissa outside the subgroup (decompression succeeds; the subgroup predicate says no)."""
    from blsutil import point_pool
    import point_cases
    p1, p2 = point_pool(12)
    for group, pool in ((1, p1), (2, p2)):
        w = 1 if group == 1 else 2
        flat = (lambda v: [v]) if group == 1 else (lambda v: list(v))
        xs, sg = [], []
        for pt in pool:
            s0 = int(O.sgn0_fq(pt[1]) if group == 1 else O.sgn0_fq2(pt[1]))
            xs += [pt[0], pt[0]]; sg += [s0, 1 - s0]
        xs.append(pool[0][0]); sg.append(2 | 1)                                         # infinity marker wins
        off = next(x for x in range(1, 200) if O.decompress_point(group, x if group == 1 else (x, 3), 0)[0] is None)
        xs.append(off if group == 1 else (off, 3)); sg.append(0)
        outside = O.jac_to_affine(O._F1 if group == 1 else O._F2, point_cases.outside_subgroup(group, 1))
        xs.append(outside[0]); sg.append(int(O.sgn0_fq(outside[1]) if group == 1 else O.sgn0_fq2(outside[1])))
        out, fl = engine.decompress(group, enc_fps([c for x in xs for c in flat(x)]), np.array(sg, dtype=np.uint8))
        got = dec_fps(out)
        for i, (x, s) in enumerate(zip(xs, sg)):
            pt, efl = O.decompress_point(group, x, s)
            exp = [0] * (2 * w) if pt is None else flat(pt[0]) + flat(pt[1])
            assert got[2 * w * i:2 * w * (i + 1)] == exp and fl[i] == efl, (group, i)
        assert got[-2 * w:] == flat(outside[0]) + flat(outside[1])
        jac = np.concatenate([out.reshape(len(xs), 12 * w)[[0, -1]], np.tile(enc_fps(flat(1 if group == 1 else (1, 0))), (2, 1))], axis=1).reshape(-1)
        _, sub = engine.point_op(group, B.POINT_IS_IN_SUBGROUP, jac)
        assert list(sub & B.FLAG_TRUE) == [B.FLAG_TRUE, 0]


def test_tower_identities_kat5(engine):
    """x^2 == x*x and (a+b) c^2 == c^2 a + c^2 b on device (the reference's test_fq12_arithmetic shape)."""
    rng = random.Random(55)
    n = 64
    xs = [rand_fp12(rng) for _ in range(n)]; ys = [rand_fp12(rng) for _ in range(n)]
    X, Y = enc_fp12s(xs), enc_fp12s(ys)
    assert np.array_equal(engine.fp12_op(B.FP12_SQR, X), engine.fp12_op(B.FP12_MUL, X, X))
    inv = engine.fp12_op(B.FP12_INV, X)
    one = enc_fp12s([O.FP12_ONE] * n)
    assert np.array_equal(engine.fp12_op(B.FP12_MUL, X, inv), one)
    assert dec_fp12s(engine.fp12_op(B.FP12_MUL, X, Y))[:8] == [O.fp12_mul(x, y) for x, y in zip(xs[:8], ys[:8])]


@pytest.mark.parametrize("shortcut", [0, 1])
def test_mode_a_golden(engine, shortcut):
    cases = load_golden("mode_a.json")["cases"]
    g1 = enc_fps([v for c in cases for v in hx(c["g1_jac"])]); g2 = enc_fps([v for c in cases for v in hx(c["g2_jac"])])
    engine.set_option(B.OPT_REF_OPT_FP2_SHORTCUT, shortcut)
    try:
        out, fl = engine.optimized_miller_loop(g1, g2)
    finally:
        engine.set_option(B.OPT_REF_OPT_FP2_SHORTCUT, 0)
    assert dec_fp12s(out) == [_fp12(c["out"]) for c in cases]
    assert list(fl) == [c["flags"] for c in cases]
    # only the c0.c0 slot is populated
    o = out.reshape(len(cases), 12, 6)
    assert not o[:, 2:, :].any()


def test_mode_a_vs_c_oracle_batch(engine, coracle):
    n = 1024 + 37                                  # ragged against every block size
    g1, g2 = synthetic_batch(n, seed=4242, jacobian=True, pool=128)
    exp, efl = coracle.miller_ref_opt(g1, g2, threads=16)
    for block in (64, 256):
        engine.set_option(B.OPT_BLOCK_SIZE, block)
        out, fl = engine.optimized_miller_loop(g1, g2)
        assert np.array_equal(out, exp) and np.array_equal(fl, efl), block
    engine.set_option(B.OPT_BLOCK_SIZE, 64)
    engine.set_option(B.OPT_REF_OPT_FP2_SHORTCUT, 1)
    out, fl = engine.optimized_miller_loop(g1, g2)
    engine.set_option(B.OPT_REF_OPT_FP2_SHORTCUT, 0)
    assert np.array_equal(out, exp)


def test_line_function_three_branches(engine):
    cases = load_golden("line_fn.json")["cases"]
    q1 = enc_fps([v for c in cases for v in hx(c["q1"])]); q2 = enc_fps([v for c in cases for v in hx(c["q2"])])
    p = enc_fps([v for c in cases for v in hx(c["p"])])
    assert dec_fps(engine.line_ref_opt(q1, q2, p)) == [v for c in cases for v in hx(c["n"]) + hx(c["d"])]


def test_mode_b_golden(engine):
    g = load_golden("mode_b.json")
    s = g["single"]
    g1 = enc_fps([v for c in s for v in hx(c["g1"])]); g2 = enc_fps([v for c in s for v in hx(c["g2"])])
    out, fl = engine.multi_miller_loop(g1, g2, 1)
    assert dec_fp12s(out) == [_fp12(c["miller"]) for c in s] and list(fl) == [c["flags"] for c in s]
    out, fl = engine.multi_pairing(g1, g2, 1)
    assert dec_fp12s(out) == [_fp12(c["pairing"]) for c in s] and list(fl) == [c["flags"] for c in s]
    m = g["multi"]
    g1 = enc_fps([v for c in m for p in c["g1"] for v in hx(p)]); g2 = enc_fps([v for c in m for q in c["g2"] for v in hx(q)])
    out, fl = engine.multi_pairing(g1, g2, 3)
    assert dec_fp12s(out) == [_fp12(c["pairing"]) for c in m] and list(fl) == [c["flags"] for c in m]
    out, fl = engine.multi_miller_loop(g1, g2, 3)
    assert dec_fp12s(out) == [_fp12(c["miller"]) for c in m]
    # miller then separate final exponentiation == fused pairing
    assert np.array_equal(engine.final_exponentiation(out), engine.multi_pairing(g1, g2, 3)[0])


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 7, 9])
def test_mode_b_vs_c_oracle_batch(engine, coracle, k):
    """Products of k pairs (`multi_miller_loop(terms: &[..])` takes any length, src/miller_loop_native.rs:154): one launch up to
    k = 3, groups of three multiplied on the device beyond (k = 4 is 3 + 1, k = 7 is 3 + 3 + 1); pairs at
    infinity in the first and in a later group (their flag must survive the later groups' launches)."""
    n = 300 + k
    g1, g2 = synthetic_batch(n, seed=9000 + k, jacobian=False, k=k, pool=50)
    g1 = g1.copy(); g2 = g2.copy()
    g1.reshape(n, k, 12)[3, 0] = 0                       # element 3: first pair at infinity
    g2.reshape(n, k, 24)[5, k - 1] = 0                   # element 5: last pair at infinity
    exp, efl = coracle.multi(g1, g2, k, final_exp=True, threads=16)
    out, fl = engine.multi_pairing(g1, g2, k)
    assert np.array_equal(out, exp) and np.array_equal(fl, efl)
    expm, _ = coracle.multi(g1, g2, k, final_exp=False, threads=16)
    outm, _ = engine.multi_miller_loop(g1, g2, k)
    assert np.array_equal(outm, expm)


def test_pairing_equals_textbook_ate_cubed_on_device(engine):
    """The GPU pairing against tests/textbook_ate.py directly (untwisted Q, affine lines over Fp12, one big power) -- a
    check that shares no curve or final-exponentiation formula with the kernels or with oracle/."""
    import textbook_ate as T
    from blsutil import point_pool
    p1, p2 = point_pool(16)
    idx = [(0, 0), (5, 9), (15, 2)]
    g1 = enc_fps([c for i, _ in idx for c in flat_g1_aff(p1[i])]); g2 = enc_fps([c for _, j in idx for c in flat_g2_aff(p2[j])])
    out, fl = engine.multi_pairing(g1, g2, 1)
    cube = lambda a: O.fp12_mul(O.fp12_mul(a, a), a)
    assert dec_fp12s(out) == [cube(T.textbook_ate(p1[i], p2[j])) for i, j in idx] and not fl.any()


def test_bilinearity_on_device(engine):
    """e(aP, bQ) == e(P, Q)^(ab) == e(abP, Q), computed on the GPU."""
    a, b = 11, 23
    P, Q = O.G1_GEN, O.G2_GEN
    ins = [(O.g1_mul(a), O.g2_mul(b)), (O.g1_mul(a * b), Q), (P, O.g2_mul(a * b)), (P, Q)]
    g1 = enc_fps([c for p, q in ins for c in flat_g1_aff(p)]); g2 = enc_fps([c for p, q in ins for c in flat_g2_aff(q)])
    out, _ = engine.multi_pairing(g1, g2, 1)
    e = dec_fp12s(out)
    assert e[0] == e[1] == e[2] == O.fp12_pow(e[3], a * b)
    assert not O.fp12_is_one(e[3]) and O.fp12_is_one(O.fp12_pow(e[3], O.R_ORDER))


def test_canonical_io_and_empty_and_errors(engine):
    c = load_golden("mode_b.json")["single"][0]
    engine.set_option(B.OPT_IO_FORM, IO_CANONICAL)
    try:
        out, _ = engine.multi_pairing(enc_fps(hx(c["g1"]), IO_CANONICAL), enc_fps(hx(c["g2"]), IO_CANONICAL), 1)
        assert dec_fp12s(out, IO_CANONICAL) == [_fp12(c["pairing"])]
    finally:
        engine.set_option(B.OPT_IO_FORM, IO_ARK_MONT)
    out, fl = engine.multi_pairing(np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint64), 1)   # empty batch
    assert out.size == 0 and fl.size == 0
    out, fl = engine.optimized_miller_loop(np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint64))
    assert out.size == 0
    import ctypes
    z = np.zeros(72, dtype=np.uint64); zp = z.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    assert engine.lib.bls381_multi_pairing_batch(engine._ctx, zp, zp, 0, zp, None, 1) == 1               # k = 0: BLS381_E_INVALID
    with pytest.raises(B.Bls381Error):
        engine.set_option(B.OPT_BLOCK_SIZE, 100)
    with pytest.raises(AssertionError):
        engine.multi_pairing(np.zeros(13, dtype=np.uint64), np.zeros(24, dtype=np.uint64), 1)           # ragged
    one = dec_fp12s(engine.multi_miller_loop_literal(3))
    assert all(O.fp12_is_one(x) for x in one)                                                            # config 0 / SURVEY F4


@pytest.fixture(scope="module")
def pools(engine):
    """4096-point pools (k+1) G1 / (k+1) G2 built on the GPU by the library's own point kernels."""
    from bls381_amd import synthetic
    return synthetic.build_pools(engine, 4096)


def _oracle_sample(coracle, g1, g2, out, n, count, w1=12, w2=24):
    idx = (np.arange(count, dtype=np.int64) * (n // count)) % n
    idx[-1] = n - 1                                            # always include the very last element
    exp, _ = coracle.multi(np.ascontiguousarray(g1.reshape(n, w1)[idx]).reshape(-1), np.ascontiguousarray(g2.reshape(n, w2)[idx]).reshape(-1),
                           1, final_exp=True, threads=16)
    return np.array_equal(out.reshape(n, 72)[idx].reshape(-1), exp)


def test_synthetic_pools_are_the_multiples_of_the_generators(engine, pools):
    """Pool entry k is (k+1) G: entries 0, 1, 2, 1000, 4095 against big-integer scalar multiplication."""
    p1, p2 = pools
    for k in (0, 1, 2, 1000, 4095):
        assert dec_fps(p1[k]) == list(O.g1_mul(k + 1)), k
        q = O.g2_mul(k + 1)
        assert dec_fps(p2[k]) == [q[0][0], q[0][1], q[1][0], q[1][1]], k


def test_full_size_batch_properties(engine, coracle, pools):
    """BASELINE size (2^20 full pairings) on 2^19 DISTINCT pairings: element i + 2^19 pairs the same multiples the other way
    round, e((a+1) G1, (b+1) G2) = e((b+1) G1, (a+1) G2), so (1) the two halves must be equal bit for bit -- 302 MB against
    302 MB, no two compared elements computed from the same inputs -- and (2) 4096 elements spread over the batch must equal
    the oracle."""
    from bls381_amd import synthetic
    n = 1 << 20
    g1, g2 = synthetic.mirrored_affine_batch(pools, n)
    out, fl = engine.multi_pairing(g1, g2, 1)
    assert not fl.any()
    o = out.reshape(2, n // 2, 72)
    assert np.array_equal(o[0], o[1])
    # really distinct pairings, not a tiled pool (e depends on (a+1)(b+1) mod r only, so some index pairs coincide in value)
    assert len(np.unique(o[0][:, :6], axis=0)) > (n // 2) * 0.8
    assert _oracle_sample(coracle, g1, g2, out, n, 4096)


def test_staged_fexp_multi_pass_small_chunks(engine, coracle):
    """The `first > 0` passes of the staged final exponentiation (batches above the chunk size) with the chunk lowered to
    2^12: n = 5 chunks + 77 (above 2^14, where an explicit pipeline count applies) at one, two and three pipelines, every element
    against the oracle."""
    n = 5 * 4096 + 77
    g1, g2 = synthetic_batch(n, seed=4242, jacobian=False, pool=64)
    exp, efl = coracle.multi(g1, g2, 1, final_exp=True, threads=16)
    try:
        engine.set_option(B.OPT_FEXP_MODE, B.FEXP_STAGED)
        engine.set_option(B.OPT_FEXP_CHUNK_LOG2, 12)
        for pipes in (1, 2, 3):
            engine.set_option(B.OPT_FEXP_PIPELINES, pipes)
            out, fl = engine.multi_pairing(g1, g2, 1)
            assert np.array_equal(out, exp) and np.array_equal(fl, efl), pipes
    finally:
        engine.set_option(B.OPT_FEXP_MODE, B.FEXP_AUTO); engine.set_option(B.OPT_FEXP_CHUNK_LOG2, 20); engine.set_option(B.OPT_FEXP_PIPELINES, 0)


def test_staged_fexp_above_one_chunk_full_size(engine, coracle, pools):
    """n = 2^20 + 2^19 + 77 with the default chunk of 2^20: the second pass and its pipelines at production size.  Checked
    against the batched-inversion form (a different code path, one pass) on every element and against the oracle on a sample."""
    from bls381_amd import synthetic
    n = (1 << 20) + (1 << 19) + 77
    g1, g2 = synthetic.affine_batch(pools, n, 1, first=12345)
    out, fl = engine.multi_pairing(g1, g2, 1)                               # AUTO -> staged, two passes
    try:
        engine.set_option(B.OPT_FEXP_MODE, B.FEXP_BATCHED_INVERSION)
        out2, fl2 = engine.multi_pairing(g1, g2, 1)
    finally:
        engine.set_option(B.OPT_FEXP_MODE, B.FEXP_AUTO)
    assert np.array_equal(out, out2) and not fl.any() and not fl2.any()
    assert _oracle_sample(coracle, g1, g2, out, n, 1024)


@pytest.mark.parametrize("n", [(1 << 10) - 64, 1 << 10, (1 << 18) - 64, 1 << 18])
def test_fexp_auto_both_sides_of_the_switch(engine, coracle, pools, n):
    """BLS381_FEXP_AUTO picks the batched-inversion form below 2^10 elements and the staged form from 2^10 on, one pipeline below
    2^18 elements and two from there on."""
    from bls381_amd import synthetic
    g1, g2 = synthetic.mirrored_affine_batch(pools, n, first=777)
    out, fl = engine.multi_pairing(g1, g2, 1)
    o = out.reshape(2, n // 2, 72)
    assert np.array_equal(o[0], o[1]) and not fl.any()
    assert _oracle_sample(coracle, g1, g2, out, n, 512)
    # the library's own record of what it ran (BLS381_INFO_*: bench.py builds its kernel labels from these)
    assert engine.get_option(B.INFO_LAST_FEXP_MODE) == (B.FEXP_STAGED if n >= (1 << 10) else B.FEXP_BATCHED_INVERSION)
    assert engine.get_option(B.INFO_LAST_FEXP_PIPELINES) == (2 if n >= (1 << 18) else 1)
    assert engine.get_option(B.INFO_LAST_MILLER_HOMOGENEOUS) == 1 and engine.get_option(B.INFO_LAST_MILLER_GROUP) == 1
    engine.multi_miller_loop(g1[: 12 * 64], g2[: 24 * 64], 1)
    assert engine.get_option(B.INFO_LAST_FEXP_MODE) == -1 and engine.get_option(B.INFO_LAST_MILLER_HOMOGENEOUS) == 0


def test_miller_only_64k_generic_mode_a_and_mode_b(engine, coracle, pools):
    """configs[1] at full size (2^16 Miller loops, no final exponentiation), every element against the oracle: Mode A as the
    reference computes it (generic Fq12 accumulators, general-Z inputs) and Mode B (Alg. 26/27)."""
    from bls381_amd import synthetic
    n = 1 << 16
    j1, j2 = synthetic.jacobian_batch(engine, pools, n, first=99)
    assert engine.get_option(B.OPT_REF_OPT_FP2_SHORTCUT) == 0
    out, fl = engine.optimized_miller_loop(j1, j2)
    exp, efl = coracle.miller_ref_opt(j1, j2, threads=16)
    assert np.array_equal(out, exp) and np.array_equal(fl, efl)
    a1, a2 = synthetic.affine_batch(pools, n, 1, first=99)
    outm, flm = engine.multi_miller_loop(a1, a2, 1)
    expm, eflm = coracle.multi(a1, a2, 1, final_exp=False, threads=16)
    assert np.array_equal(outm, expm) and np.array_equal(flm, eflm)
    # the Jacobian inputs are representatives of the same points: normalising them gives the affine batch back
    nrm, _ = engine.point_op(2, B.POINT_NORMALIZE, j2)
    assert np.array_equal(nrm.reshape(n, 36)[:, :24].reshape(-1), a2)


def test_mode_a_64k_batch_properties(engine, coracle):
    """configs[1] size (2^16 Miller loops, Mode A) with the Fp2 shortcut."""
    n = 1 << 16
    pool = 64
    g1, g2 = synthetic_batch(n, jacobian=True, pool=pool)
    engine.set_option(B.OPT_REF_OPT_FP2_SHORTCUT, 1)
    try:
        out, fl = engine.optimized_miller_loop(g1, g2)
    finally:
        engine.set_option(B.OPT_REF_OPT_FP2_SHORTCUT, 0)
    o = out.reshape(n // pool, pool * 72)
    exp, _ = coracle.miller_ref_opt(g1[: pool * 18], g2[: pool * 36], threads=16)
    assert np.array_equal(o[0], exp) and (o == o[0]).all() and not fl.any()


def test_host_pointer_pipeline_pieces_and_slot_reuse(engine, coracle, pools):
    """The host-pointer entry points move batches above 1.5 * 2^18 elements in pieces of 2^18 through two device slots
    (upload of piece i+1 and download of piece i-1 under the kernels of piece i): n = 2^20 + 77 is five pieces, so both slots
    are reused twice.  Every element against the resident `_dev` path, a sample against the oracle; a reused output buffer."""
    import torch
    from bls381_amd import synthetic
    n = (1 << 20) + 77
    g1, g2 = synthetic.affine_batch(pools, n, 1, first=31337)
    out = np.empty(n * 72, dtype=np.uint64)
    o2, fl = engine.multi_pairing(g1, g2, 1, out=out)
    assert o2 is out and not fl.any()
    dev = torch.device("cuda", 0)
    d1 = torch.from_numpy(g1.view(np.int64)).to(dev); d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    dout = torch.zeros(n * 72, dtype=torch.int64, device=dev)
    engine.multi_pairing_dev(d1.data_ptr(), d2.data_ptr(), 1, dout.data_ptr(), 0, n)
    engine.sync()
    assert np.array_equal(dout.cpu().numpy().view(np.uint64), out)
    assert _oracle_sample(coracle, g1, g2, out, n, 512)
    out[:] = 0
    engine.multi_pairing(g1[: 12 * 1000], g2[: 24 * 1000], 1)              # a small call in between re-uses slot 0
    engine.multi_pairing(g1, g2, 1, out=out)
    assert np.array_equal(dout.cpu().numpy().view(np.uint64), out)


def test_multi_device_context_shards_contiguous_ranges(coracle):
    """bls381_create_multi / *_sharded: one host process, one context per listed device, contiguous ranges on host threads.
    The box has one GPU, so it is listed twice and three times (two / three contexts, ranges of different sizes) and once by
    default (every visible device); n = 1 leaves contexts with empty ranges."""
    for ids, n, k in (([0, 0], 3001, 1), ([0, 0, 0], 500, 3), (None, 257, 1), ([0, 0], 1, 1)):
        m = B.MultiEngine(ids)
        try:
            assert m.device_count == (len(ids) if ids else 1)
            g1, g2 = synthetic_batch(n, seed=555 + n, jacobian=False, k=k, pool=40)
            exp, efl = coracle.multi(g1, g2, k, final_exp=True, threads=16)
            out, fl = m.multi_pairing(g1, g2, k)
            assert np.array_equal(out, exp) and np.array_equal(fl, efl), (ids, n, k)
            expm, _ = coracle.multi(g1, g2, k, final_exp=False, threads=16)
            outm, _ = m.multi_miller_loop(g1, g2, k)
            assert np.array_equal(outm, expm)
        finally:
            m.close()
    m = B.MultiEngine([0, 0, 0])
    try:                                                   # the Groth16 shape with prepared fixed points, three uneven ranges
        n = 1000
        g1, g2 = synthetic_batch(n, seed=8123, jacobian=False, k=3, pool=50)
        g2 = g2.copy().reshape(n, 3, 24); g2[:, 1] = g2[0, 1]; g2[:, 2] = g2[0, 2]
        eng0 = B.Engine(0)
        prep, _ = eng0.g2_prepare(np.ascontiguousarray(g2[0, 1:3]).reshape(-1))
        exp, efl = coracle.multi(g1, g2.reshape(-1), 3, final_exp=True, threads=16)
        out, fl = m.groth16_pairing(g1, np.ascontiguousarray(g2[:, 0]).reshape(-1), prep)
        assert np.array_equal(out, exp) and np.array_equal(fl, efl)
        eng0.close()
    finally:
        m.close()
    m = B.MultiEngine([0, 0])
    try:
        j1, j2 = synthetic_batch(301, seed=77, jacobian=True, pool=30)
        exp, efl = coracle.miller_ref_opt(j1, j2, threads=16)
        out, fl = m.optimized_miller_loop(j1, j2)
        assert np.array_equal(out, exp) and np.array_equal(fl, efl)
        m.set_option(B.OPT_FEXP_MODE, B.FEXP_SELF_CONTAINED)
        with pytest.raises(B.Bls381Error):
            m.set_option(B.OPT_BLOCK_SIZE, 100)
    finally:
        m.close()
    with pytest.raises(B.Bls381Error):
        B.MultiEngine([0, 99])


def test_device_pointer_api_on_caller_stream(engine, coracle):
    """The `_dev` entry points (what bench.py times): device buffers owned by the caller, launches enqueued on an
    adopted stream, no implicit synchronisation."""
    import torch
    n = 512
    g1, g2 = synthetic_batch(n, seed=77, jacobian=False, pool=32)
    dev = torch.device("cuda", 0)
    d1 = torch.from_numpy(g1.view(np.int64)).to(dev); d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    out = torch.zeros(n * 72, dtype=torch.int64, device=dev); fl = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream()
    engine.set_stream(stream.cuda_stream)
    try:
        with torch.cuda.stream(stream):
            engine.multi_pairing_dev(d1.data_ptr(), d2.data_ptr(), 1, out.data_ptr(), fl.data_ptr(), n)
        stream.synchronize()
        assert engine.last_kernel_ms() > 0
    finally:
        engine.set_stream(None)
    exp, efl = coracle.multi(g1, g2, 1, final_exp=True, threads=8)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), exp) and np.array_equal(fl.cpu().numpy(), efl)
    # Miller-only and final-exponentiation device entry points compose to the same result
    mid = torch.zeros(n * 72, dtype=torch.int64, device=dev)
    engine.multi_miller_loop_dev(d1.data_ptr(), d2.data_ptr(), 1, mid.data_ptr(), 0, n)
    out2 = torch.zeros(n * 72, dtype=torch.int64, device=dev)
    engine.final_exponentiation_dev(mid.data_ptr(), out2.data_ptr(), 0, n)
    engine.sync()
    assert torch.equal(out, out2)


def test_final_exp_scratch_across_streams_and_regrowth(engine):
    """The split final exponentiation parks norms in context-owned device scratch and every launch is asynchronous: a
    second call on ANOTHER stream, with a larger batch (the scratch is re-allocated while the first call may still be
    running), must neither race with nor corrupt the first."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    n1, n2 = 4096, 12288
    g1, g2 = synthetic_batch(n2, seed=99, jacobian=False, pool=32)
    d1 = torch.from_numpy(g1.view(np.int64)).to(dev); d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    ref = torch.zeros(n2 * 72, dtype=torch.int64, device=dev)
    engine.multi_pairing_dev(d1.data_ptr(), d2.data_ptr(), 1, ref.data_ptr(), 0, n2)
    engine.sync()
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    o1 = torch.zeros(n1 * 72, dtype=torch.int64, device=dev); o2 = torch.zeros(n2 * 72, dtype=torch.int64, device=dev)
    e2 = B.Engine(0)                                      # fresh context: its scratch starts empty and must grow twice
    try:
        e2.set_stream(s1.cuda_stream)
        e2.multi_pairing_dev(d1.data_ptr(), d2.data_ptr(), 1, o1.data_ptr(), 0, n1)
        e2.set_stream(s2.cuda_stream)
        e2.multi_pairing_dev(d1.data_ptr(), d2.data_ptr(), 1, o2.data_ptr(), 0, n2)
        s1.synchronize(); s2.synchronize()
    finally:
        e2.set_stream(None)
    assert torch.equal(o1, ref[: n1 * 72]) and torch.equal(o2, ref)


def test_witness_generator_backend(engine):
    """bls381_witness_*: the run_once hooks (fq12/fq6/fq2 inverse, fq/fq2 sqrt with sgn0 fix-up) vs the oracle."""
    from witness_cases import cases, enc_can, as_u32
    c = cases(seed=99, n=40)
    for key, fn, width in (("fq12", engine.witness_fq12_inverse, 12), ("fq6", engine.witness_fq6_inverse, 6)):
        xs, exp = c[key]
        out, fl = fn(as_u32(enc_can([v for x in xs for v in x])))
        got = dec_fps(out.view(np.uint64), IO_CANONICAL)
        assert [got[width * i:width * (i + 1)] for i in range(len(xs))] == [e[0] for e in exp], key
        assert list(fl) == [e[1] for e in exp], key
    xs, exp = c["fq2"]
    out, fl = engine.witness_fq2_inverse(as_u32(enc_can([v for x in xs for v in x])))
    got = dec_fps(out.view(np.uint64), IO_CANONICAL)
    assert [tuple(got[2 * i:2 * i + 2]) for i in range(len(xs))] == [tuple(e) for e in exp] and not fl.any()
    xs, sg, exp = c["sqrt"]
    out, fl = engine.witness_fq_sqrt(as_u32(enc_can(xs)), sg)
    assert dec_fps(out.view(np.uint64), IO_CANONICAL) == [e[0] for e in exp] and list(fl) == [e[1] for e in exp]
    xs, sg, exp = c["sqrt2"]
    out, fl = engine.witness_fq2_sqrt(as_u32(enc_can([v for x in xs for v in x])), sg)
    got = dec_fps(out.view(np.uint64), IO_CANONICAL)
    assert [tuple(got[2 * i:2 * i + 2]) for i in range(len(xs))] == [tuple(e[0]) for e in exp] and list(fl) == [e[1] for e in exp]
    # round trip at scale: x * inverse(x) == 1 for 4096 random Fq12 (MyFq12 order in, tree order for the product)
    rng = random.Random(4)
    big = [rand_fp12(rng) for _ in range(512)]
    inv, fl = engine.witness_fq12_inverse(as_u32(enc_can([v for x in big for v in O.fp12_to_myfq12(x)])))
    invs = [O.myfq12_to_fp12(dec_fps(inv.view(np.uint64)[72 * i:72 * (i + 1)], IO_CANONICAL)) for i in range(len(big))]
    assert all(O.fp12_is_one(O.fp12_mul(x, y)) for x, y in zip(big[:32], invs[:32])) and not fl.any()


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 65])
def test_tiny_and_ragged_batches(engine, coracle, n):
    """Batches that do not fill a wave (32 pairings per 64-lane wave in the lane-pair layout)."""
    g1, g2 = synthetic_batch(n, seed=500 + n, jacobian=False, pool=8)
    out, fl = engine.multi_pairing(g1, g2, 1)
    exp, efl = coracle.multi(g1, g2, 1, final_exp=True)
    assert np.array_equal(out, exp) and np.array_equal(fl, efl)
    j1, j2 = synthetic_batch(n, seed=600 + n, jacobian=True, pool=8)
    out, fl = engine.optimized_miller_loop(j1, j2)
    exp, efl = coracle.miller_ref_opt(j1, j2)
    assert np.array_equal(out, exp) and np.array_equal(fl, efl)


def test_unreduced_and_infinity_inputs(engine, coracle):
    """Field elements >= p are accepted and reduced on load; infinity in any slot of a product skips that pair."""
    vals = [O.P, O.P + 7, 2 ** 384 - 1, 2 * O.P + 5]
    a = np.array([(v >> (64 * i)) & (2 ** 64 - 1) for v in vals for i in range(6)], dtype=np.uint64)
    engine.set_option(B.OPT_IO_FORM, IO_CANONICAL)
    try:
        out = dec_fps(engine.fp_op(B.FP_MUL, a, enc_fps([1] * len(vals), IO_CANONICAL)), IO_CANONICAL)
    finally:
        engine.set_option(B.OPT_IO_FORM, IO_ARK_MONT)
    assert out == [v % O.P for v in vals]
    pairs = random_pairs(8, 321, jacobian=False)
    for k in (2, 4):
        terms = [pairs[:k], [(None, pairs[1][1])] + pairs[1:k], [(p, None) for p, q in pairs[:k]]]
        g1 = enc_fps([c for t in terms for p, q in t for c in flat_g1_aff(p)])
        g2 = enc_fps([c for t in terms for p, q in t for c in flat_g2_aff(q)])
        out, fl = engine.multi_pairing(g1, g2, k)
        exp, efl = coracle.multi(g1, g2, k, final_exp=True)
        assert np.array_equal(out, exp) and np.array_equal(fl, efl)
        got = dec_fp12s(out)
        assert got[0] == O.multi_pairing(terms[0]) and O.fp12_is_one(got[2]) and list(fl) == [0, 2, 2]


def test_g2_prepare_and_fp_div(engine):
    """bls381_g2_prepare_batch (68 coefficient triples per point) and BLS381_FP_DIV."""
    pts = [O.G2_GEN, None] + [q for _, q in random_pairs(30, 909, jacobian=False)]
    out, fl = engine.g2_prepare(enc_fps([c for q in pts for c in flat_g2_aff(q)]))
    got = dec_fps(out)
    exp = [v for q in pts for (a, b, c) in O.g2_prepare(q if q is not None else O.G2_GEN) for v in (a[0], a[1], b[0], b[1], c[0], c[1])]
    assert got == exp and list(fl) == [0, 2] + [0] * 30
    # the coefficients drive a Miller loop: ell(f) = f.mul_by_014(c, b * P.x, a * P.y) reproduces miller_loop
    P, Q = O.G1_GEN, O.G2_GEN
    f = O.FP12_ONE
    co = O.g2_prepare(Q)
    it = iter(co)
    found = False
    for bbit in range(63, -1, -1):
        bit = ((O.BLS_X >> 1) >> bbit) & 1
        if not found:
            found = bit == 1
            continue
        f = O.ell(f, next(it), P)
        if bit:
            f = O.ell(f, next(it), P)
        f = O.fp12_sqr(f)
    f = O.fp12_conj(O.ell(f, next(it), P))
    assert f == O.miller_loop(P, Q)
    rng = random.Random(3)
    xs = [rand_fp(rng) for _ in range(100)] + [5]; ys = [rand_fp(rng) for _ in range(100)] + [0]
    assert dec_fps(engine.fp_op(B.FP_DIV, enc_fps(xs), enc_fps(ys))) == [(x * pow(y, O.P - 2, O.P)) % O.P for x, y in zip(xs, ys)]


def test_cpp_host_mirror_end_to_end(tmp_path):
    """plonky2-bls12-381-pairing_amd/host/bls381_pairing.hpp: the reference's names (optimized_miller_loop,
    multi_miller_loop, MillerLoopResult::final_exponentiation) driven from C++ against the GPU library."""
    import os, subprocess
    from blsutil import ROOT
    exe = tmp_path / "host_mirror_run"
    libdir = os.path.dirname(B.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", os.path.join(ROOT, "tests", "host_mirror_run.cpp"), "-o", str(exe), "-L", libdir,
                           "-lbls381_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    (Pj, Qj), = random_pairs(1, 4711, jacobian=True)
    (Pa, Qa), = random_pairs(1, 4712, jacobian=False)
    blob = enc_fps(flat_g1_jac(Pj) + flat_g2_jac(Qj) + flat_g1_aff(Pa) + flat_g2_aff(Qa)).tobytes()
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    inp.write_bytes(blob)
    subprocess.check_call([str(exe), str(inp), str(outp)])
    out = np.frombuffer(outp.read_bytes(), dtype=np.uint64)
    got = dec_fp12s(out)
    assert got[0] == O.optimized_miller_loop(Pj, Qj, False)[0]
    assert got[1] == O.pairing(Pa, Qa)
    assert O.fp12_is_one(got[2])


def test_pow_fq_legendre_sgn0(engine):
    """bls381_fp_pow_batch (pow_fq, helpers.rs:176-195) and the LEGENDRE / IS_SQUARE / SGN0 ops."""
    from test_witness_hostsim import _pow_cases
    xs, exps = _pow_cases()
    a = enc_fps(xs)
    for e in exps:
        words = [(e >> (64 * i)) & (2 ** 64 - 1) for i in range(max(1, (e.bit_length() + 63) // 64))]
        exp = [O.pow_fq(x, words) for x in xs]
        out, fl = engine.fp_pow(a, e)
        assert list(fl) == [f for v, f in exp], hex(e)
        got = dec_fps(out)
        assert [g for g, (v, f) in zip(got, exp) if f == 0] == [v for v, f in exp if f == 0], hex(e)
    assert dec_fps(engine.fp_op(B.FP_LEGENDRE, a)) == [O.legendre(x) for x in xs]
    assert dec_fps(engine.fp_op(B.FP_IS_SQUARE, a)) == [1 if O.legendre(x) == 1 else 0 for x in xs]
    assert dec_fps(engine.fp_op(B.FP_SGN0, a)) == [1 if O.sgn0_fq(x) else 0 for x in xs]


def test_groth16_fixed_g2_path(engine, coracle):
    """bls381_groth16_pairing_batch: e(P0,Q0) e(P1,Q1) e(P2,Q2) with Q1, Q2 fixed and staged in LDS as prepared
    coefficients -- must equal the generic k = 3 product bit for bit (also across infinity cases and ragged sizes)."""
    n = 300 + 37
    p1, p2 = random_pairs(2, 1234, jacobian=False)
    Q1, Q2 = p1[1], p2[1]
    prep, pfl = engine.g2_prepare(enc_fps(flat_g2_aff(Q1) + flat_g2_aff(Q2)))
    assert not pfl.any()
    base = random_pairs(3 * 64, 4321, jacobian=False)
    proofs = []
    for i in range(n):
        P0, Q0 = base[(3 * i) % len(base)]
        P1 = base[(3 * i + 1) % len(base)][0]
        P2 = base[(3 * i + 2) % len(base)][0]
        if i == 5: P1 = None                      # infinity in a fixed slot
        if i == 6: Q0 = None                      # infinity in the variable slot
        if i == 7: P0 = P1 = P2 = None            # everything skipped -> one
        proofs.append((P0, P1, P2, Q0))
    g1 = enc_fps([c for P0, P1, P2, Q0 in proofs for P in (P0, P1, P2) for c in flat_g1_aff(P)])
    g2v = enc_fps([c for P0, P1, P2, Q0 in proofs for c in flat_g2_aff(Q0)])
    out, fl = engine.groth16_pairing(g1, g2v, prep)
    g2full = enc_fps([c for P0, P1, P2, Q0 in proofs for Q in (Q0, Q1, Q2) for c in flat_g2_aff(Q)])
    exp, efl = coracle.multi(g1, g2full, 3, final_exp=True, threads=16)
    assert np.array_equal(out, exp) and np.array_equal(fl, efl)
    assert np.array_equal(out, engine.multi_pairing(g1, g2full, 3)[0])
    assert O.fp12_is_one(dec_fp12s(out[7 * 72:8 * 72])[0]) and fl[5] == 2 and fl[6] == 2 and fl[7] == 2


def _groth16_shape_batch(pools, n, fixed=None, first=0):
    """n proofs x 3 pairs from the GPU-built pools (bls381_amd/synthetic.py indices).  The second half of the batch holds the
    first half's proofs written differently: every per-proof pair mirrored -- ((a+1) G1, (b+1) G2) -> ((b+1) G1, (a+1) G2), the
    same pairing value from different inputs -- and, in the generic shape, the three pairs rotated (a product does not depend on
    the order of its factors).  fixed = (f1, f2): pairs 1 and 2 use the pool points Q[f1], Q[f2] on EVERY proof (the verifying
    key's points of a Groth16 batch); only pair 0 is mirrored then.  Returns g1 [n, 3, 12], g2 [n, 3, 24] (uint64)."""
    from bls381_amd import synthetic
    p1, p2 = pools
    h = n // 2
    ia, ib = synthetic.element_indices(h * 3, p1.shape[0], first * 3)
    ia, ib = ia.reshape(h, 3), ib.reshape(h, 3)
    if fixed is None:
        rot = [1, 2, 0]
        g1 = np.concatenate([p1[ia], p1[ib[:, rot]]])
        g2 = np.concatenate([p2[ib], p2[ia[:, rot]]])
    else:
        g1 = np.concatenate([p1[ia], p1[np.stack([ib[:, 0], ia[:, 1], ia[:, 2]], axis=1)]])
        g2 = np.concatenate([p2[ib], p2[np.stack([ia[:, 0], ib[:, 1], ib[:, 2]], axis=1)]])
        g2[:, 1] = p2[fixed[0]]
        g2[:, 2] = p2[fixed[1]]
    return np.ascontiguousarray(g1), np.ascontiguousarray(g2)


def _spread(n, count):
    """`count` element indices spread over [0, n) with a stride coprime to the pool size, first and last element included"""
    idx = (np.arange(count, dtype=np.int64) * 4099 + 17) % n
    idx[0], idx[-1] = 0, n - 1
    return idx


@pytest.mark.parametrize("lg", [15, 18])
def test_groth16_shape_at_size_generic(engine, coracle, pools, lg):
    """BASELINE config 4 (3 pairs x 2^18 proofs; 2^15 = one GPU's share of it on 8 GPUs), generic k = 3 product
    (bls381_multi_pairing_batch: `multi_miller_loop(terms)` of src/miller_loop_native.rs:154-212 + the final exponentiation):
    (1) proof i + n/2 = proof i with its pairs mirrored and rotated -> the two halves must be equal bit for bit, no two compared
    elements computed from the same inputs; (2) 768 proofs spread over the batch against the C oracle's k = 3 product."""
    n = 1 << lg
    g1, g2 = _groth16_shape_batch(pools, n, first=4242)
    out, fl = engine.multi_pairing(g1.reshape(-1), g2.reshape(-1), 3)
    assert not fl.any()
    o = out.reshape(2, n // 2, 72)
    assert np.array_equal(o[0], o[1])
    assert len(np.unique(o[0][:, :6], axis=0)) > (n // 2) * 0.8            # distinct products, not a tiled pool
    idx = _spread(n, 768)
    exp, efl = coracle.multi(g1[idx].reshape(-1), g2[idx].reshape(-1), 3, final_exp=True, threads=16)
    assert np.array_equal(out.reshape(n, 72)[idx].reshape(-1), exp) and not efl.any()


@pytest.mark.parametrize("lg", [15, 18])
def test_groth16_shape_at_size_prepared(engine, coracle, pools, lg):
    """The same shape with the two fixed G2 points prepared (bls381_groth16_pairing_batch: `G2PreparedTarget` coefficients,
    src/miller_loop_target.rs:23-76, consumed from LDS): (1) EVERY element equal to the generic k = 3 product of the same
    inputs (another kernel: k_multi_miller<3> walks three curves, k_prepared_miller<1,2> walks one and reads 2 x 68 coefficient
    triples); (2) halves equal under mirroring of the per-proof pair; (3) 768 spread proofs against the C oracle."""
    n = 1 << lg
    f1, f2 = 17, 4000
    g1, g2 = _groth16_shape_batch(pools, n, fixed=(f1, f2), first=777)
    prep, pfl = engine.g2_prepare(np.ascontiguousarray(g2[0, 1:3]).reshape(-1))
    assert not pfl.any()
    out, fl = engine.groth16_pairing(g1.reshape(-1), np.ascontiguousarray(g2[:, 0]).reshape(-1), prep)
    assert not fl.any()
    gen, gfl = engine.multi_pairing(g1.reshape(-1), g2.reshape(-1), 3)
    assert np.array_equal(out, gen) and not gfl.any()
    o = out.reshape(2, n // 2, 72)
    assert np.array_equal(o[0], o[1])
    idx = _spread(n, 768)
    exp, _ = coracle.multi(g1[idx].reshape(-1), g2[idx].reshape(-1), 3, final_exp=True, threads=16)
    assert np.array_equal(out.reshape(n, 72)[idx].reshape(-1), exp)


@pytest.mark.parametrize("n_var,n_fixed", [(0, 1), (1, 1), (0, 2)])
def test_prepared_pairing_shapes(engine, coracle, n_var, n_fixed):
    """bls381_prepared_pairing_batch: every supported (variable, fixed) shape equals the generic product."""
    n = 130
    fixed = [q for _, q in random_pairs(n_fixed, 2222 + n_fixed, jacobian=False)]
    prep, _ = engine.g2_prepare(enc_fps([c for q in fixed for c in flat_g2_aff(q)]))
    k = n_var + n_fixed
    base = random_pairs(k * 32, 3333 + 10 * n_var + n_fixed, jacobian=False)
    elems = []
    for i in range(n):
        ps = [base[(k * i + j) % len(base)][0] for j in range(k)]
        qs = ([base[(k * i) % len(base)][1]] if n_var else []) + fixed
        if i == 3: ps[-1] = None
        elems.append((ps, qs))
    g1 = enc_fps([c for ps, qs in elems for p in ps for c in flat_g1_aff(p)])
    g2v = enc_fps([c for ps, qs in elems for c in flat_g2_aff(qs[0])]) if n_var else None
    out, fl = engine.prepared_pairing(g1, g2v, prep, n_var, n_fixed)
    g2full = enc_fps([c for ps, qs in elems for q in qs for c in flat_g2_aff(q)])
    exp, efl = coracle.multi(g1, g2full, k, final_exp=True, threads=16)
    assert np.array_equal(out, exp) and np.array_equal(fl, efl)


def test_fp2_fp6_surfaces(engine):
    """bls381_fp2_op_batch / bls381_fp6_op_batch against the oracle's tower formulas (T2, T3)."""
    rng = random.Random(66)
    n = 70
    a2 = [rand_fp2(rng) for _ in range(n)] + [(0, 0), (1, 0), (0, 1)]
    b2 = [rand_fp2(rng) for _ in range(n)] + [(5, 7), (O.P - 1, 3), (O.P - 1, O.P - 1)]
    A = enc_fps([c for v in a2 for c in v]); Bq = enc_fps([c for v in b2 for c in v])
    pair = lambda r: [(r[2 * i], r[2 * i + 1]) for i in range(len(r) // 2)]
    assert pair(dec_fps(engine.fp2_op(B.FP2_ADD, A, Bq))) == [O.fp2_add(x, y) for x, y in zip(a2, b2)]
    assert pair(dec_fps(engine.fp2_op(B.FP2_SUB, A, Bq))) == [O.fp2_sub(x, y) for x, y in zip(a2, b2)]
    assert pair(dec_fps(engine.fp2_op(B.FP2_MUL, A, Bq))) == [O.fp2_mul(x, y) for x, y in zip(a2, b2)]
    assert pair(dec_fps(engine.fp2_op(B.FP2_SQR, A))) == [O.fp2_sqr(x) for x in a2]
    assert pair(dec_fps(engine.fp2_op(B.FP2_NEG, A))) == [O.fp2_neg(x) for x in a2]
    assert pair(dec_fps(engine.fp2_op(B.FP2_CONJ, A))) == [O.fp2_conj(x) for x in a2]
    assert pair(dec_fps(engine.fp2_op(B.FP2_MUL_BY_NONRESIDUE, A))) == [O.fp2_mul_by_nonresidue(x) for x in a2]
    assert pair(dec_fps(engine.fp2_op(B.FP2_INV, A))) == [O.fp2_inv(x) or (0, 0) for x in a2]
    a6 = [tuple(rand_fp2(rng) for _ in range(3)) for _ in range(n)] + [O.FP6_ZERO, O.FP6_ONE]
    b6 = [tuple(rand_fp2(rng) for _ in range(3)) for _ in range(n)] + [O.FP6_ONE, O.FP6_ONE]
    f6 = lambda xs: enc_fps([c for x in xs for q in x for c in q])
    d6 = lambda arr: [tuple((r[6 * i + 2 * j], r[6 * i + 2 * j + 1]) for j in range(3)) for r in [dec_fps(arr)] for i in range(len(r) // 6)]
    A6, B6 = f6(a6), f6(b6)
    assert d6(engine.fp6_op(B.FP6_ADD, A6, B6)) == [O.fp6_add(x, y) for x, y in zip(a6, b6)]
    assert d6(engine.fp6_op(B.FP6_SUB, A6, B6)) == [O.fp6_sub(x, y) for x, y in zip(a6, b6)]
    assert d6(engine.fp6_op(B.FP6_MUL, A6, B6)) == [O.fp6_mul(x, y) for x, y in zip(a6, b6)]
    assert d6(engine.fp6_op(B.FP6_SQR, A6)) == [O.fp6_sqr(x) for x in a6]
    assert d6(engine.fp6_op(B.FP6_NEG, A6)) == [O.fp6_neg(x) for x in a6]
    assert d6(engine.fp6_op(B.FP6_MUL_BY_NONRESIDUE, A6)) == [O.fp6_mul_by_nonresidue(x) for x in a6]
    assert d6(engine.fp6_op(B.FP6_FROBENIUS, A6)) == [O.fp6_frobenius(x) for x in a6]
    assert d6(engine.fp6_op(B.FP6_MUL_BY_01, A6, B6)) == [O.fp6_mul_by_01(x, y[0], y[1]) for x, y in zip(a6, b6)]
    assert d6(engine.fp6_op(B.FP6_MUL_BY_1, A6, B6)) == [O.fp6_mul_by_1(x, y[0]) for x, y in zip(a6, b6)]
    assert d6(engine.fp6_op(B.FP6_INV, A6)) == [O.fp6_inv(x) or O.FP6_ZERO for x in a6]


def test_kat5_reference_fixed_inputs_on_device(engine):
    """The reference's test_fq6_arithmetic / test_fq12_arithmetic identities on their own fixed limbs, evaluated by the
    GPU tower (raw limbs fed as canonical integers, like the reference does)."""
    K = load_golden("reference_kats.json")
    engine.set_option(B.OPT_IO_FORM, IO_CANONICAL)
    try:
        raw = lambda vals: np.array([(v >> (64 * i)) & (2 ** 64 - 1) for v in vals for i in range(6)], dtype=np.uint64)
        k6 = K["kat5_fq6_inputs"]
        a, b, c = raw(k6["a"]), raw(k6["b"]), raw(k6["c"])
        for x in (a, b, c):
            assert np.array_equal(engine.fp6_op(B.FP6_SQR, x), engine.fp6_op(B.FP6_MUL, x, x))
        c2 = engine.fp6_op(B.FP6_SQR, c)
        lhs = engine.fp6_op(B.FP6_MUL, engine.fp6_op(B.FP6_ADD, a, b), c2)
        rhs = engine.fp6_op(B.FP6_ADD, engine.fp6_op(B.FP6_MUL, c2, a), engine.fp6_op(B.FP6_MUL, c2, b))
        assert np.array_equal(lhs, rhs)
        assert dec_fps(c2, IO_CANONICAL) == [v for q in O.fp6_sqr(tuple((k6["c"][2 * i] % O.P, k6["c"][2 * i + 1] % O.P) for i in range(3))) for v in q]
        k12 = K["kat5_fq12_inputs"]
        a, b, c = raw(k12["a"]), raw(k12["b"]), raw(k12["c"])
        sq = lambda x: engine.fp12_op(B.FP12_SQR, x)
        # Fq12 addition through the Fp6 surface (two halves)
        add12 = lambda x, y: engine.fp6_op(B.FP6_ADD, x, y)
        a = add12(sq(sq(a)), c); b = add12(sq(sq(b)), a); c = add12(sq(sq(c)), b)      # :904-914
        for x in (a, b, c):
            assert np.array_equal(sq(x), engine.fp12_op(B.FP12_MUL, x, x))
        c2 = sq(c)
        lhs = engine.fp12_op(B.FP12_MUL, add12(a, b), c2)
        rhs = add12(engine.fp12_op(B.FP12_MUL, c2, a), engine.fp12_op(B.FP12_MUL, c2, b))
        assert np.array_equal(lhs, rhs)
    finally:
        engine.set_option(B.OPT_IO_FORM, IO_ARK_MONT)
