"""Pins the oracle (oracle/pyref.py) against every known answer the reference's own files hold for this
path (numbers extracted by tools/extract_reference_kats.py), and against
the algebraic invariants that stand in for the Miller-loop vectors the reference lacks."""
from blsutil import O, load_golden, hx

K = load_golden("reference_kats.json")


def test_kat6_p_and_loop_constants():
    d = K["kat6_p_u32_digits"]["digits_le"]
    assert sum(x << (32 * i) for i, x in enumerate(d)) == O.P
    assert int(K["kat6_scalar_field"]["p_decimal"]) == O.P
    assert int(K["kat6_scalar_field"]["r_decimal"]) == O.R_ORDER
    assert K["kat6_bls_x"]["bls_x"] == O.BLS_X and K["kat6_bls_x"]["negative"] is True
    bits = K["kat6_pseudo_binary_encoding"]["bits"]
    assert bits == O.PSEUDO_BINARY_ENCODING and sum(bits) == 6
    assert [i for i, b in enumerate(bits) if b] == [16, 48, 57, 60, 62, 63]
    assert (O.P ** 12 - 1) % O.R_ORDER == 0 and O.P % 4 == 3


def test_kat1_kat2_fp2_add_sub():
    for name, fn in (("kat1_fp2_add", O.fp2_add), ("kat2_fp2_sub", O.fp2_sub)):
        k = K[name]
        a, b, c = tuple(k["a"]), tuple(k["b"]), tuple(k["c"])
        assert fn(a, b) == c, name


def test_kat3_generators_are_montgomery_forms():
    k = K["kat3_generators_montgomery"]
    assert O.to_mont(O.G1_GEN[0]) == k["g1_x"] and O.to_mont(O.G1_GEN[1]) == k["g1_y"]
    assert [O.to_mont(c) for c in O.G2_GEN[0]] == k["g2_x"]
    assert [O.to_mont(c) for c in O.G2_GEN[1]] == k["g2_y"]
    assert O.g1_on_curve(O.G1_GEN) and O.g2_on_curve(O.G2_GEN)
    # limb codec round trip
    assert O.fp_from_limbs(O.fp_to_limbs(O.G1_GEN[0])) == O.G1_GEN[0]


def test_kat4_frobenius_constants():
    k = K["kat4_frobenius_montgomery"]
    assert O.FROB6_C1[0] == 0 and O.to_mont(O.FROB6_C1[1]) == k["frob6_c1_c1"]
    assert O.FROB6_C2[1] == 0 and O.to_mont(O.FROB6_C2[0]) == k["frob6_c2_c0"]
    assert [O.to_mont(c) for c in O.FROB12_C1] == k["frob12_c1"]


def test_kat5_tower_identities_and_myfq12_convention():
    import random
    rng = random.Random(5)
    for _ in range(3):
        x = O.fp12_unflatten([rng.randrange(O.P) for _ in range(12)])
        a = O.fp12_unflatten([rng.randrange(O.P) for _ in range(12)])
        b = O.fp12_unflatten([rng.randrange(O.P) for _ in range(12)])
        assert O.fp12_sqr(x) == O.fp12_mul(x, x)                                    # x^2 == x*x
        c2 = O.fp12_sqr(x)
        ab = (O.fp6_add(a[0], b[0]), O.fp6_add(a[1], b[1]))
        lhs = O.fp12_mul(ab, c2)
        r1, r2 = O.fp12_mul(c2, a), O.fp12_mul(c2, b)
        assert lhs == (O.fp6_add(r1[0], r2[0]), O.fp6_add(r1[1], r2[1]))            # (a+b) c^2 == c^2 a + c^2 b
        # src/fields/helpers.rs:248-267 (test_myfq12): flat schoolbook product with w^6 = 1 + u
        assert O.myfq12_mul(O.fp12_to_myfq12(a), O.fp12_to_myfq12(b)) == O.fp12_to_myfq12(O.fp12_mul(a, b))
        assert O.fp12_frobenius(x) == O.fp12_pow(x, O.P)
        assert O.fp12_is_one(O.fp12_mul(x, O.fp12_inv(x)))


def test_mode_a_generator_matches_survey_known_answer():
    # SURVEY.md App. B.3 (derived independently by the surveyor's scratch model)
    out, fl = O.optimized_miller_loop((O.G1_GEN[0], O.G1_GEN[1], 1), (O.G2_GEN[0], O.G2_GEN[1], O.FP2_ONE))
    assert fl == 0
    flat = O.fp12_flatten(out)
    assert flat[0] == 0x0E55CD9A3D95A195EC346327D178B2FFC9F64BA2549836A96ABFDA2F36D30A7E5EC79CCD733A41654F7A4AC5734442EB
    assert flat[1] == 0x083760D368BA995315D0DD4F05CDAA088AAA6AFBBB73F38B01F17807D4AB16AFA527D7BE455E0F9D9DC471E05C804F57
    assert all(v == 0 for v in flat[2:])
    tr = []
    O.optimized_miller_loop((O.G1_GEN[0], O.G1_GEN[1], 1), (O.G2_GEN[0], O.G2_GEN[1], O.FP2_ONE), False, tr)
    assert tr.count("T") == 64 and tr.count("A") == 6 and "V" not in tr


def test_mode_a_generic_equals_fp2_shortcut():
    g = load_golden("mode_a.json")["cases"][1]
    P = tuple(hx(g["g1_jac"])); q = hx(g["g2_jac"]); Q = ((q[0], q[1]), (q[2], q[3]), (q[4], q[5]))
    a, _ = O.optimized_miller_loop(P, Q, True)
    b, _ = O.optimized_miller_loop(P, Q, False)
    assert a == b and O.fp12_flatten(a) == hx(g["out"])


def test_mode_b_invariants():
    e = O.pairing(O.G1_GEN, O.G2_GEN)
    assert not O.fp12_is_one(e) and O.fp12_is_one(O.fp12_pow(e, O.R_ORDER))
    a, b = 7, 13
    assert O.pairing(O.g1_mul(a), O.g2_mul(b)) == O.fp12_pow(e, a * b)               # bilinearity
    qn = (O.G2_GEN[0], O.fp2_neg(O.G2_GEN[1]))
    assert O.fp12_is_one(O.fp12_mul(O.pairing(O.G1_GEN, qn), e))                     # e(P,-Q) e(P,Q) = 1
    pa, qb = O.g1_mul(a), O.g2_mul(b)
    assert O.multi_miller_loop([(pa, qb), (O.G1_GEN, O.G2_GEN)]) == O.fp12_mul(O.miller_loop(pa, qb), O.miller_loop(O.G1_GEN, O.G2_GEN))
    assert len(O.g2_prepare(O.G2_GEN)) == 68                                         # src/miller_loop_target.rs:69
    assert O.fp12_is_one(O.multi_miller_loop_literal([(O.G1_GEN, O.G2_GEN)]))


def test_final_exponentiation_is_the_closed_form_power():
    """The whole easy-part / hard-part chain of src/fields_as_trees/miller_loop.rs:128-178 (Frobenius maps, five
    cyclotomic exponentiations, conjugations) equals ONE big-integer power: f ^ (3 (p^12 - 1) / r).  The factor 3 is the
    hard part's (x-1)^2 (x+p)(x^2+p^2-1) + 3 decomposition."""
    import random
    rng = random.Random(12)
    e = 3 * ((O.P ** 12 - 1) // O.R_ORDER)
    for _ in range(2):
        f = O.fp12_unflatten([rng.randrange(O.P) for _ in range(12)])
        assert O.final_exponentiation(f) == O.fp12_pow(f, e)


def test_mode_b_equals_textbook_ate_pairing_cubed():
    """Independent pin for Mode B (the reference holds no pairing vector): a textbook ate pairing -- untwisted Q, affine
    chord-and-tangent lines over Fp12, one big power (tests/textbook_ate.py; no formula shared with Alg. 26/27, the sparse
    products or the cyclotomic chain) -- cubed equals the oracle's pairing, on the generators and on other multiples;
    the Miller values themselves differ only by a factor the final exponentiation kills."""
    import textbook_ate as T
    from blsutil import point_pool
    p1, p2 = point_pool(16)
    cube = lambda a: O.fp12_mul(O.fp12_mul(a, a), a)
    for i, j in ((0, 0), (3, 1), (7, 12)):
        assert O.pairing(p1[i], p2[j]) == cube(T.textbook_ate(p1[i], p2[j])), (i, j)
    # Miller-only: reference-style f (conjugated, x < 0) times the textbook f_{|x|} is killed by the final exponentiation
    m = O.fp12_mul(O.miller_loop(p1[0], p2[0]), T.miller_function(p1[0], p2[0]))
    assert O.fp12_is_one(O.final_exponentiation(m))
    # k = 2 product
    prod = O.fp12_mul(T.textbook_ate(p1[1], p2[2]), T.textbook_ate(p1[4], p2[3]))
    assert O.multi_pairing([(p1[1], p2[2]), (p1[4], p2[3])]) == cube(prod)


def test_golden_files_reproduce():
    """The committed fixtures are what the oracle computes today (guards against silent oracle drift)."""
    for c in load_golden("mode_a.json")["cases"][:4]:
        P = tuple(hx(c["g1_jac"])); q = hx(c["g2_jac"]); Q = ((q[0], q[1]), (q[2], q[3]), (q[4], q[5]))
        out, fl = O.optimized_miller_loop(P, Q, False)
        assert O.fp12_flatten(out) == hx(c["out"]) and fl == c["flags"]
    for c in load_golden("mode_b.json")["single"][:3]:
        P = tuple(hx(c["g1"])); q = hx(c["g2"]); Q = ((q[0], q[1]), (q[2], q[3]))
        assert O.fp12_flatten(O.pairing(P, Q)) == hx(c["pairing"])


def _kat5_values(key, width):
    k = K[key]
    red = lambda v: [x % O.P for x in v]          # the reference feeds raw limbs to a canonical-limb type
    return red(k["a"]), red(k["b"]), red(k["c"]), width


def test_kat5_reference_fixed_inputs_fq6():
    """test_fq6_arithmetic (src/fields_as_trees/fq6_target_tree.rs:392-647) on its own fixed limbs:
    x^2 == x*x for a, b, c and (a+b) c^2 == c^2 a + c^2 b."""
    a, b, c, _ = _kat5_values("kat5_fq6_inputs", 6)
    f6 = lambda v: ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))
    a, b, c = f6(a), f6(b), f6(c)
    for x in (a, b, c):
        assert O.fp6_sqr(x) == O.fp6_mul(x, x)
    c2 = O.fp6_sqr(c)
    assert O.fp6_mul(O.fp6_add(a, b), c2) == O.fp6_add(O.fp6_mul(c2, a), O.fp6_mul(c2, b))


def test_kat5_reference_fixed_inputs_fq12():
    """test_fq12_arithmetic (src/fields_as_trees/fq12_target_tree.rs:448-942), including its a = a^4 + c,
    b = b^4 + a, c = c^4 + b preamble (:904-914)."""
    a, b, c, _ = _kat5_values("kat5_fq12_inputs", 12)
    a, b, c = O.fp12_unflatten(a), O.fp12_unflatten(b), O.fp12_unflatten(c)
    add12 = lambda x, y: (O.fp6_add(x[0], y[0]), O.fp6_add(x[1], y[1]))
    a = add12(O.fp12_sqr(O.fp12_sqr(a)), c)
    b = add12(O.fp12_sqr(O.fp12_sqr(b)), a)
    c = add12(O.fp12_sqr(O.fp12_sqr(c)), b)
    for x in (a, b, c):
        assert O.fp12_sqr(x) == O.fp12_mul(x, x)
    c2 = O.fp12_sqr(c)
    assert O.fp12_mul(add12(a, b), c2) == add12(O.fp12_mul(c2, a), O.fp12_mul(c2, b))


def test_decompression_and_subgroup_oracle_round_trip():
    """oracle/pyref.py decompress_point / jac_in_subgroup: (x, sgn0(y)) of a subgroup point gives the point back; the other sign
    gives its negative; an x off the curve is refused; the generators are in the subgroup, a random curve point is not."""
    from blsutil import point_pool
    p1, p2 = point_pool(6)
    for pt in p1:
        assert O.decompress_point(1, pt[0], int(O.sgn0_fq(pt[1]))) == ((pt[0], pt[1]), 0)
        assert O.decompress_point(1, pt[0], int(not O.sgn0_fq(pt[1])))[0] == (pt[0], (-pt[1]) % O.P)
    for q in p2:
        assert O.decompress_point(2, q[0], int(O.sgn0_fq2(q[1]))) == ((q[0], q[1]), 0)
    assert O.decompress_point(1, 5, 2) == (None, O.FLAG_INFINITY)
    bad = next(x for x in range(1, 50) if O.fp_sqrt_any((x ** 3 + 4) % O.P) is None)
    assert O.decompress_point(1, bad, 0) == (None, O.WFLAG_NOT_SQUARE)
    import point_cases
    for group, F, gen in ((1, O._F1, O.G1_GEN), (2, O._F2, O.G2_GEN)):
        assert O.jac_in_subgroup(F, O.jac_from_affine(F, gen))
        out = point_cases.outside_subgroup(group, 1)
        assert O.jac_on_curve(F, out) and not O.jac_in_subgroup(F, out)


def test_hard_part_exponent_identity_of_the_staged_final_exponentiation():
    """csrc/fexp_stages.hip.h walks the hard part along (x - 1)^2 (x + p) (x^2 + p^2 - 1) + 3; the reference's chain
    (src/fields_as_trees/miller_loop.rs:137-163) reaches 3 (p^4 - p^2 + 1) / r.  Same integer, hence the same field element."""
    x = -O.BLS_X
    assert (O.P ** 4 - O.P ** 2 + 1) % O.R_ORDER == 0
    assert (x - 1) ** 2 * (x + O.P) * (x * x + O.P * O.P - 1) + 3 == 3 * (O.P ** 4 - O.P ** 2 + 1) // O.R_ORDER


def test_mode_a_strength_reductions_keep_the_value():
    """What the device kernel computes for optimized_miller_loop -- `f * f` by the squaring formula, the embedded line value as an
    Fp2 scalar, `R + R` by the doubling formulas -- is the same Fq12 value as the literal restatement, on
    general-Z points, on garbage coordinates and with P or Q at infinity."""
    import random
    from blsutil import random_pairs
    rng = random.Random(31)
    cases = [(P, Q) for P, Q in random_pairs(3, 77, jacobian=True)]
    cases.append(((rng.randrange(O.P), rng.randrange(O.P), rng.randrange(O.P)),
                  tuple((rng.randrange(O.P), rng.randrange(O.P)) for _ in range(3))))          # not on any curve
    P0, Q0 = cases[0]
    cases.append(((P0[0], P0[1], 0), Q0))
    cases.append((P0, (Q0[0], Q0[1], (0, 0))))
    for P, Q in cases:
        assert O.optimized_miller_loop_strength_reduced(P, Q) == O.optimized_miller_loop(P, Q)
