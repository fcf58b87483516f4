"""Pure-Python big-integer restatement of the reference's native pairing path.

TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it, and only as
the checker.  The shipped path is the HIP library behind include/bls381_hip.h.

PARITY STATUS
  * Field/tower layer: pinned by the reference's own known answers (KAT-1..KAT-6,
    SURVEY.md section 4) -- see tests/golden/reference_kats.json and tests/test_oracle_kats.py.
  * Miller loops (Mode A `optimized_miller_loop`, Mode B optimal-ate): PARITY UNPINNED by
    the reference -- its only test of the hot path prints the result without asserting
    (src/miller_loop_native_optimized.rs:138-145) and the arkworks crates that do the
    arithmetic (ark-bls12-381 0.4.0, ark-ff 0.4.2, ark-ec 0.4.2; Cargo.toml:11-14) are
    not present in /root/reference and no Rust toolchain exists here.  This file restates
    the published algorithms of those crates (EFD dbl-2009-l / add-2007-bl Jacobian
    formulas, a = 0) at the reference's call sites; Mode B is additionally pinned by
    algebraic invariants (bilinearity, e^r = 1, non-degeneracy).

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Elements are plain Python ints in CANONICAL form; tuples build the
tower: Fp2 = (c0, c1), Fp6 = (c0, c1, c2) of Fp2, Fp12 = (c0, c1) of Fp6.
"""
from __future__ import annotations

# --------------------------------------------------------------------------- constants
# p: src/fields/bls12_381base.rs:108-113 (u32 digits of order()); KAT-6
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
# r: decimal string at src/miller_loop_native_optimized.rs:110
R_ORDER = 52435875175126190479447740508185965837690552500527637822603658699938581184513
# loop parameter: src/utils/constants.rs:1-2, src/global_constants.rs:7-8
BLS_X = 0xD201_0000_0001_0000
BLS_X_IS_NEGATIVE = True
# src/global_constants.rs:3-6 -- index i holds bit i (LSB first) of BLS_X
PSEUDO_BINARY_ENCODING = [(BLS_X >> i) & 1 for i in range(64)]

MONT_R = (1 << 384) % P          # arkworks/zkcrypto Montgomery radix R = 2^384
MONT_R2 = (MONT_R * MONT_R) % P
MONT_RINV = pow(MONT_R, -1, P)
P_INV_NEG64 = (-pow(P, -1, 1 << 64)) % (1 << 64)
P_INV_NEG32 = (-pow(P, -1, 1 << 32)) % (1 << 32)

# Standard generators (canonical).  KAT-3 checks them against the Montgomery-form limbs the
# reference hard-codes at src/curves/g1_curve_target.rs:42-59 and g2_curve_target.rs:69-112.
G1_GEN = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)
G2_GEN = (
    (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
     0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
     0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
)

# ------------------------------------------------------------------ operation counters
# Fp-level work per pairing feeds DESIGN.md's integer roofline.
class OpCount:
    def __init__(self):
        self.reset()

    def reset(self):
        self.mul = 0   # Fp multiplications (squarings counted as multiplications)
        self.add = 0   # Fp additions / subtractions / negations / doublings
        self.inv = 0   # Fp inversions

    def snapshot(self):
        return {"fp_mul": self.mul, "fp_add": self.add, "fp_inv": self.inv}


OPS = OpCount()

# ------------------------------------------------------------------------------- Fp
# Semantics: src/fields/bls12_381base.rs:174-268 (add/sub/neg/mul/div mod p, canonical results),
# inverse by Fermat :118-125.

def fp_add(a, b):
    OPS.add += 1
    return (a + b) % P


def fp_sub(a, b):
    OPS.add += 1
    return (a - b) % P


def fp_neg(a):
    OPS.add += 1
    return (-a) % P


def fp_mul(a, b):
    OPS.mul += 1
    return (a * b) % P


def fp_inv(a):
    """None for zero, like arkworks `inverse()` / Bls12_381Base::try_inverse (:118-125)."""
    if a % P == 0:
        return None
    OPS.inv += 1
    return pow(a, P - 2, P)


def to_mont(a):
    return (a * MONT_R) % P


def from_mont(a):
    return (a * MONT_RINV) % P


def fp_to_limbs(a, mont=True):
    """6 x u64 little-endian limbs; mont=True gives the arkworks in-memory form [3P]."""
    v = to_mont(a) if mont else a % P
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]


def fp_from_limbs(limbs, mont=True):
    v = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return from_mont(v) if mont else v % P


# ------------------------------------------------------------------------------- Fp2
# Fp2 = Fp[u]/(u^2+1).  Formulas: src/fields_as_trees/fq2_target_tree.rs:54-142.
FP2_ZERO = (0, 0)
FP2_ONE = (1, 0)


def fp2_add(a, b):
    return (fp_add(a[0], b[0]), fp_add(a[1], b[1]))


def fp2_sub(a, b):
    return (fp_sub(a[0], b[0]), fp_sub(a[1], b[1]))


def fp2_neg(a):
    return (fp_neg(a[0]), fp_neg(a[1]))


def fp2_dbl(a):
    return fp2_add(a, a)


def fp2_mul(a, b):
    """fq2_target_tree.rs:97-115 (value identical for schoolbook and Karatsuba)."""
    t0 = fp_mul(a[0], b[0])
    t1 = fp_mul(a[1], b[1])
    t2 = fp_mul(fp_add(a[0], a[1]), fp_add(b[0], b[1]))
    return (fp_sub(t0, t1), fp_sub(fp_sub(t2, t0), t1))


def fp2_sqr(a):
    """Complex squaring, fq2_target_tree.rs:80-91."""
    s = fp_add(a[0], a[1])
    d = fp_sub(a[0], a[1])
    c = fp_add(a[0], a[0])
    return (fp_mul(s, d), fp_mul(c, a[1]))


def fp2_mul_fp(a, k):
    """Fp2 x Fp, used by `ell` (src/miller_loop_native.rs:143-147, commented)."""
    return (fp_mul(a[0], k), fp_mul(a[1], k))


def fp2_conj(a):
    """fq2_target_tree.rs:117-122; also the Frobenius map on Fp2 (:93-95)."""
    return (a[0], fp_neg(a[1]))


def fp2_mul_by_nonresidue(a):
    """Multiply by xi = 1 + u: fq2_target_tree.rs:137-142."""
    return (fp_sub(a[0], a[1]), fp_add(a[0], a[1]))


def fp2_inv(a):
    """fq2_target_tree.rs:66-78; None when a == 0."""
    n = fp_add(fp_mul(a[0], a[0]), fp_mul(a[1], a[1]))
    ni = fp_inv(n)
    if ni is None:
        return None
    return (fp_mul(a[0], ni), fp_neg(fp_mul(a[1], ni)))


def fp2_is_zero(a):
    return a[0] % P == 0 and a[1] % P == 0


def fp2_eq(a, b):
    return (a[0] - b[0]) % P == 0 and (a[1] - b[1]) % P == 0


def fp2_from_int(k):
    return (k % P, 0)


def fp2_pow(a, e):
    r = FP2_ONE
    for bit in bin(e)[2:]:
        r = fp2_sqr(r)
        if bit == "1":
            r = fp2_mul(r, a)
    return r


# ------------------------------------------------------------------------------- Fp6
# Fp6 = Fp2[v]/(v^3 - xi).  Formulas: src/fields_as_trees/fq6_target_tree.rs:59-293.
FP6_ZERO = (FP2_ZERO, FP2_ZERO, FP2_ZERO)
FP6_ONE = (FP2_ONE, FP2_ZERO, FP2_ZERO)

# Frobenius constants in CANONICAL form (the reference file stores Montgomery-form limbs:
# fq6_target_tree.rs:139-146, :155-162, fq12_target_tree.rs:101-119; KAT-4 checks them).
XI = (1, 1)
FROB6_C1 = None   # xi^((p-1)/3)
FROB6_C2 = None   # xi^((2p-2)/3)
FROB12_C1 = None  # xi^((p-1)/6)


def _init_frobenius_constants():
    global FROB6_C1, FROB6_C2, FROB12_C1
    saved = OPS.snapshot()
    FROB6_C1 = fp2_pow(XI, (P - 1) // 3)
    FROB6_C2 = fp2_pow(XI, (2 * P - 2) // 3)
    FROB12_C1 = fp2_pow(XI, (P - 1) // 6)
    OPS.mul, OPS.add, OPS.inv = saved["fp_mul"], saved["fp_add"], saved["fp_inv"]


_init_frobenius_constants()


def fp6_add(a, b):
    return tuple(fp2_add(x, y) for x, y in zip(a, b))


def fp6_sub(a, b):
    return tuple(fp2_sub(x, y) for x, y in zip(a, b))


def fp6_neg(a):
    return tuple(fp2_neg(x) for x in a)


def fp6_mul_by_nonresidue(a):
    """Multiply by v: (a,b,c) -> (c*xi, a, b).  fq6_target_tree.rs:219-230."""
    return (fp2_mul_by_nonresidue(a[2]), a[0], a[1])


def fp6_mul(a, b):
    """Karatsuba, fq6_target_tree.rs:172-214."""
    aa = fp2_mul(a[0], b[0])
    bb = fp2_mul(a[1], b[1])
    cc = fp2_mul(a[2], b[2])
    t1 = fp2_mul(fp2_add(b[1], b[2]), fp2_add(a[1], a[2]))
    t1 = fp2_sub(fp2_sub(t1, bb), cc)
    t1 = fp2_add(fp2_mul_by_nonresidue(t1), aa)
    t3 = fp2_mul(fp2_add(b[0], b[2]), fp2_add(a[0], a[2]))
    t3 = fp2_sub(fp2_add(fp2_sub(t3, aa), bb), cc)
    t2 = fp2_mul(fp2_add(b[0], b[1]), fp2_add(a[0], a[1]))
    t2 = fp2_sub(fp2_sub(t2, aa), bb)
    t2 = fp2_add(t2, fp2_mul_by_nonresidue(cc))
    return (t1, t2, t3)


def fp6_sqr(a):
    """CH-SQR2, fq6_target_tree.rs:270-293."""
    s0 = fp2_sqr(a[0])
    ab = fp2_mul(a[0], a[1])
    s1 = fp2_add(ab, ab)
    s2 = fp2_sqr(fp2_add(fp2_sub(a[0], a[1]), a[2]))
    bc = fp2_mul(a[1], a[2])
    s3 = fp2_add(bc, bc)
    s4 = fp2_sqr(a[2])
    c0 = fp2_add(fp2_mul_by_nonresidue(s3), s0)
    c1 = fp2_add(fp2_mul_by_nonresidue(s4), s1)
    c2 = fp2_sub(fp2_sub(fp2_add(fp2_add(s1, s2), s3), s0), s4)
    return (c0, c1, c2)


def fp6_mul_by_01(s, c0, c1):
    """Sparse multiply by (c0, c1, 0): fq6_target_tree.rs:232-259."""
    aa = fp2_mul(s[0], c0)
    bb = fp2_mul(s[1], c1)
    t1 = fp2_add(fp2_mul_by_nonresidue(fp2_mul(s[2], c1)), aa)
    t2 = fp2_sub(fp2_sub(fp2_mul(fp2_add(c0, c1), fp2_add(s[0], s[1])), aa), bb)
    t3 = fp2_add(fp2_mul(s[2], c0), bb)
    return (t1, t2, t3)


def fp6_mul_by_1(s, c1):
    """Sparse multiply by (0, c1, 0): fq6_target_tree.rs:261-268."""
    return (fp2_mul_by_nonresidue(fp2_mul(s[2], c1)), fp2_mul(s[0], c1), fp2_mul(s[1], c1))


def fp6_inv(a):
    """fq6_target_tree.rs:59-89; None when a == 0."""
    c0 = fp2_sub(fp2_sqr(a[0]), fp2_mul_by_nonresidue(fp2_mul(a[1], a[2])))
    c1 = fp2_sub(fp2_mul_by_nonresidue(fp2_sqr(a[2])), fp2_mul(a[0], a[1]))
    c2 = fp2_sub(fp2_sqr(a[1]), fp2_mul(a[0], a[2]))
    tmp = fp2_add(fp2_mul(a[1], c2), fp2_mul(a[2], c1))
    tmp = fp2_add(fp2_mul_by_nonresidue(tmp), fp2_mul(a[0], c0))
    ti = fp2_inv(tmp)
    if ti is None:
        return None
    return (fp2_mul(c0, ti), fp2_mul(c1, ti), fp2_mul(c2, ti))


def fp6_frobenius(a):
    """fq6_target_tree.rs:129-169 with canonical constants."""
    return (fp2_conj(a[0]), fp2_mul(fp2_conj(a[1]), FROB6_C1), fp2_mul(fp2_conj(a[2]), FROB6_C2))


# ------------------------------------------------------------------------------- Fp12
# Fp12 = Fp6[w]/(w^2 - v).  Formulas: src/fields_as_trees/fq12_target_tree.rs:53-176.
FP12_ONE = (FP6_ONE, FP6_ZERO)
FP12_ZERO = (FP6_ZERO, FP6_ZERO)


def fp12_mul(a, b):
    """Karatsuba over Fp6, fq12_target_tree.rs:130-141."""
    aa = fp6_mul(a[0], b[0])
    bb = fp6_mul(a[1], b[1])
    o = fp6_add(b[0], b[1])
    c1 = fp6_mul(fp6_add(a[1], a[0]), o)
    c1 = fp6_sub(fp6_sub(c1, aa), bb)
    c0 = fp6_add(fp6_mul_by_nonresidue(bb), aa)
    return (c0, c1)


def fp12_sqr(a):
    """Complex squaring, fq12_target_tree.rs:143-155."""
    ab = fp6_mul(a[0], a[1])
    c0c1 = fp6_add(a[0], a[1])
    c0 = fp6_add(fp6_mul_by_nonresidue(a[1]), a[0])
    c0 = fp6_sub(fp6_mul(c0, c0c1), ab)
    c1 = fp6_add(ab, ab)
    c0 = fp6_sub(c0, fp6_mul_by_nonresidue(ab))
    return (c0, c1)


def fp12_conj(a):
    """fq12_target_tree.rs:53-58."""
    return (a[0], fp6_neg(a[1]))


def fp12_inv(a):
    """fq12_target_tree.rs:77-90; None when a == 0 (arkworks `inverse()` -> None [3P])."""
    t = fp6_sub(fp6_sqr(a[0]), fp6_mul_by_nonresidue(fp6_sqr(a[1])))
    ti = fp6_inv(t)
    if ti is None:
        return None
    return (fp6_mul(a[0], ti), fp6_mul(a[1], fp6_neg(ti)))


def fp12_frobenius(a):
    """fq12_target_tree.rs:92-128 with the canonical constant xi^((p-1)/6)."""
    c0 = fp6_frobenius(a[0])
    c1 = fp6_frobenius(a[1])
    c1 = fp6_mul(c1, (FROB12_C1, FP2_ZERO, FP2_ZERO))
    return (c0, c1)


def fp12_mul_by_014(f, c0, c1, c4):
    """fq12_target_tree.rs:157-176 / commented src/miller_loop_native.rs:118-137."""
    aa = fp6_mul_by_01(f[0], c0, c1)
    bb = fp6_mul_by_1(f[1], c4)
    o = fp2_add(c1, c4)
    r1 = fp6_mul_by_01(fp6_add(f[1], f[0]), c0, o)
    r1 = fp6_sub(fp6_sub(r1, aa), bb)
    r0 = fp6_add(fp6_mul_by_nonresidue(bb), aa)
    return (r0, r1)


def fp12_eq(a, b):
    return all(fp2_eq(x, y) for h, g in zip(a, b) for x, y in zip(h, g))


def fp12_is_one(a):
    return fp12_eq(a, FP12_ONE)


def fp12_pow(a, e):
    r = FP12_ONE
    for bit in bin(e)[2:]:
        r = fp12_sqr(r)
        if bit == "1":
            r = fp12_mul(r, a)
    return r


def fp12_from_fp2(v):
    """`Fq12::new(Fq6::new(v, 0, 0), Fq6::zero())` -- src/miller_loop_native_optimized.rs:18-36."""
    return ((v, FP2_ZERO, FP2_ZERO), FP6_ZERO)


def fp12_flatten(a):
    """arkworks in-memory coefficient order c0.c0.c0, c0.c0.c1, c0.c1.c0, ..., c1.c2.c1 (12 Fp)."""
    return [c for h in a for q in h for c in q]


def fp12_unflatten(v):
    v = list(v)
    return (((v[0], v[1]), (v[2], v[3]), (v[4], v[5])), ((v[6], v[7]), (v[8], v[9]), (v[10], v[11])))


def fp12_to_myfq12(a):
    """MyFq12 flat order, src/fields/helpers.rs:22-76: coeffs[k] = w^k u^0 part, coeffs[k+6] = w^k u^1.

    With w^2 = v:  w^0 -> c0.c0, w^1 -> c1.c0, w^2 -> c0.c1, w^3 -> c1.c1, w^4 -> c0.c2, w^5 -> c1.c2.
    """
    order = [a[0][0], a[1][0], a[0][1], a[1][1], a[0][2], a[1][2]]
    return [q[0] for q in order] + [q[1] for q in order]


def myfq12_mul(a, b):
    """Schoolbook product of two MyFq12 coefficient vectors with w^6 = 1 + u
    (src/fields/helpers.rs:90-152).  Used to pin the tower convention (test_myfq12, :248-267)."""
    # polynomial in w over Fp2 with coefficients (a[k], a[k+6])
    A = [(a[k], a[k + 6]) for k in range(6)]
    B = [(b[k], b[k + 6]) for k in range(6)]
    acc = [FP2_ZERO] * 11
    for i in range(6):
        for j in range(6):
            acc[i + j] = fp2_add(acc[i + j], fp2_mul(A[i], B[j]))
    out = list(acc[:6])
    for k in range(6, 11):
        out[k - 6] = fp2_add(out[k - 6], fp2_mul_by_nonresidue(acc[k]))
    return [q[0] for q in out] + [q[1] for q in out]


# --------------------------------------------------------------------- curves (Jacobian)
# Twist: y^2 = x^3 + 4(1+u) over Fp2; G1: y^2 = x^3 + 4 over Fp.
B_G1 = 4
B_G2 = (4, 4)


class _F1:
    add, sub, mul = staticmethod(fp_add), staticmethod(fp_sub), staticmethod(fp_mul)
    sqr = staticmethod(lambda a: fp_mul(a, a))
    dbl = staticmethod(lambda a: fp_add(a, a))
    inv = staticmethod(fp_inv)
    zero, one = 0, 1
    is_zero = staticmethod(lambda a: a % P == 0)
    eq = staticmethod(lambda a, b: (a - b) % P == 0)
    neg = staticmethod(fp_neg)


class _F2:
    add, sub, mul, sqr = staticmethod(fp2_add), staticmethod(fp2_sub), staticmethod(fp2_mul), staticmethod(fp2_sqr)
    dbl = staticmethod(fp2_dbl)
    inv = staticmethod(fp2_inv)
    zero, one = FP2_ZERO, FP2_ONE
    is_zero = staticmethod(fp2_is_zero)
    eq = staticmethod(fp2_eq)
    neg = staticmethod(fp2_neg)


def jac_double(F, pt):
    """ark-ec 0.4.2 `Projective::double_in_place`, a = 0 (EFD dbl-2009-l) [3P; SURVEY App. A.3].
    Call site: src/miller_loop_native_optimized.rs:93 (via `R + R`)."""
    X, Y, Z = pt
    if F.is_zero(Z):
        return pt
    A = F.sqr(X)
    B = F.sqr(Y)
    C = F.sqr(B)
    D = F.dbl(F.sub(F.sub(F.sqr(F.add(X, B)), A), C))
    E = F.add(F.add(A, A), A)
    Fv = F.sqr(E)
    Z3 = F.dbl(F.mul(Y, Z))
    X3 = F.sub(F.sub(Fv, D), D)
    C8 = F.dbl(F.dbl(F.dbl(C)))
    Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
    return (X3, Y3, Z3)


def jac_add(F, p1, p2):
    """ark-ec 0.4.2 `impl AddAssign<&Self> for Projective` (EFD add-2007-bl) [3P; SURVEY App. A.3].
    Call sites: src/miller_loop_native_optimized.rs:93, :98."""
    X1, Y1, Z1 = p1
    X2, Y2, Z2 = p2
    if F.is_zero(Z1):
        return p2
    if F.is_zero(Z2):
        return p1
    Z1Z1 = F.sqr(Z1)
    Z2Z2 = F.sqr(Z2)
    U1 = F.mul(X1, Z2Z2)
    U2 = F.mul(X2, Z1Z1)
    S1 = F.mul(F.mul(Y1, Z2), Z2Z2)
    S2 = F.mul(F.mul(Y2, Z1), Z1Z1)
    if F.eq(U1, U2) and F.eq(S1, S2):
        return jac_double(F, p1)
    H = F.sub(U2, U1)
    I = F.sqr(F.dbl(H))
    J = F.mul(H, I)
    r = F.dbl(F.sub(S2, S1))
    V = F.mul(U1, I)
    X3 = F.sub(F.sub(F.sqr(r), J), F.dbl(V))
    Y3 = F.sub(F.mul(r, F.sub(V, X3)), F.dbl(F.mul(S1, J)))
    Z3 = F.mul(F.sub(F.sub(F.sqr(F.add(Z1, Z2)), Z1Z1), Z2Z2), H)
    return (X3, Y3, Z3)


def jac_to_affine(F, pt):
    X, Y, Z = pt
    if F.is_zero(Z):
        return None
    zi = F.inv(Z)
    zi2 = F.sqr(zi)
    return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))


def jac_eq(F, p1, p2):
    """arkworks `Projective == Projective` [3P]: both at infinity, or neither and X1 Z2^2 == X2 Z1^2, Y1 Z2^3 == Y2 Z1^3.
    On normalised points this is the reference's `is_point_equal_to` (src/fields_as_trees/g2_curve.rs:129-149,
    g1_curve.rs:79-110)."""
    z1, z2 = F.is_zero(p1[2]), F.is_zero(p2[2])
    if z1 or z2:
        return z1 and z2
    a, b = F.sqr(p1[2]), F.sqr(p2[2])
    return F.eq(F.mul(p1[0], b), F.mul(p2[0], a)) and F.eq(F.mul(p1[1], F.mul(b, p2[2])), F.mul(p2[1], F.mul(a, p1[2])))


def jac_on_curve(F, pt):
    """Y^2 == X^3 + b Z^6 (b = 4 on G1, 4(1+u) on the twist); infinity counts as on the curve."""
    X, Y, Z = pt
    if F.is_zero(Z):
        return True
    z2 = F.sqr(Z)
    z6 = F.mul(F.sqr(z2), z2)
    b = (B_G1 * z6) % P if F is _F1 else fp2_mul(B_G2, z6)
    return F.eq(F.sqr(Y), F.add(F.mul(F.sqr(X), X), b))


def jac_normalize(F, pt):
    """The Z = 1 representative (`.into()` to affine and back, opt.rs:93,98; `From<G2AffineTarget>` g2_curve.rs:29-50);
    infinity -> arkworks' `Projective::zero()` = (1, 1, 0)."""
    a = jac_to_affine(F, pt)
    if a is None:
        return (F.one, F.one, F.zero)
    return (a[0], a[1], F.one)


def jac_from_affine(F, a):
    return (a[0], a[1], F.one)


def jac_scalar_mul(F, pt, k):
    acc = (F.one, F.one, F.zero)
    for bit in bin(k)[2:]:
        acc = jac_double(F, acc)
        if bit == "1":
            acc = jac_add(F, acc, pt)
    return acc


def jac_rerandomise(F, pt, lam):
    """Another Jacobian representative (l^2 X, l^3 Y, l Z) of the same point."""
    l2 = F.sqr(lam)
    return (F.mul(pt[0], l2), F.mul(pt[1], F.mul(l2, lam)), F.mul(pt[2], lam))


def g1_on_curve(a):
    return (a[1] * a[1] - a[0] * a[0] * a[0] - B_G1) % P == 0


def g2_on_curve(a):
    lhs = fp2_sqr(a[1])
    rhs = fp2_add(fp2_mul(fp2_sqr(a[0]), a[0]), B_G2)
    return fp2_eq(lhs, rhs)


def g1_mul(k, base=G1_GEN):
    return jac_to_affine(_F1, jac_scalar_mul(_F1, jac_from_affine(_F1, base), k))


def g2_mul(k, base=G2_GEN):
    return jac_to_affine(_F2, jac_scalar_mul(_F2, jac_from_affine(_F2, base), k))


# ======================================================================= MODE A
# src/miller_loop_native_optimized.rs

FLAG_DEN_ZERO = 1      # f_den == 0: reference panics at :102 (inverse().unwrap())
FLAG_INFINITY = 2      # an input with Z == 0 (Mode A) / explicit infinity (Mode B)


def optimized_line_function(Q1, Q2, Pt):
    """src/miller_loop_native_optimized.rs:8-78.  Returns the two Fp2 values that the reference
    wraps into the c0.c0 slot of an Fq12 (:18-36, :42-60, :64-75), plus the branch taken."""
    x1, y1, z1 = Q1
    x2, y2, z2 = Q2
    xp, yp, zp = ((Pt[0], 0), (Pt[1], 0), (Pt[2], 0))          # Fq2::new(xp, 0) etc.
    num = fp2_sub(fp2_mul(y2, z1), fp2_mul(y1, z2))             # :14
    den = fp2_sub(fp2_mul(x2, z1), fp2_mul(x1, z2))             # :15
    if not fp2_is_zero(den):                                    # :16 chord
        branch = "A"
    elif fp2_is_zero(num):                                      # :38 tangent
        num = fp2_mul(fp2_mul(fp2_from_int(3), x1), x1)         # :39
        den = fp2_mul(fp2_mul(fp2_from_int(2), y1), z1)         # :40
        branch = "T"
    else:                                                       # :62 vertical
        n = fp2_sub(fp2_mul(xp, z1), fp2_mul(x1, zp))           # :66
        d = fp2_mul(z1, zp)                                     # :73
        return n, d, "V"
    n = fp2_sub(fp2_mul(num, fp2_sub(fp2_mul(xp, z1), fp2_mul(x1, zp))),
                fp2_mul(den, fp2_sub(fp2_mul(yp, z1), fp2_mul(y1, zp))))   # :20-23 / :44-47
    d = fp2_mul(fp2_mul(den, zp), z1)                                      # :31 / :55
    return n, d, branch


def optimized_miller_loop(Pt, Q, generic_fp12=True, trace=None):
    """src/miller_loop_native_optimized.rs:81-127.

    Pt = (xp, yp, zp) Fp ints, Q = (x, y, z) Fp2 -- arkworks Jacobian coordinates used raw.
    Returns (Fp12, flags).  With generic_fp12=True every product is a full Fq12 multiply like the
    reference executes; with False the same values are computed in the Fp2 subfield --
    the outputs are bit-identical.
    """
    R = Q
    flags = 0
    if generic_fp12:
        mul, one, emb = fp12_mul, FP12_ONE, fp12_from_fp2
    else:
        mul, one, emb = fp2_mul, FP2_ONE, (lambda v: v)
    f_num, f_den = one, one                                          # :83-84
    for v in PSEUDO_BINARY_ENCODING:                                 # :89 (index 0 first = LSB)
        n, d, br = optimized_line_function(R, R, Pt)                 # :90
        if trace is not None:
            trace.append(br)
        f_num = mul(mul(f_num, f_num), emb(n))                       # :91
        f_den = mul(mul(f_den, f_den), emb(d))                       # :92
        R = jac_add(_F2, R, R)                                       # :93
        if v == 1:
            n, d, br = optimized_line_function(R, Q, Pt)             # :95
            if trace is not None:
                trace.append(br)
            f_num = mul(f_num, emb(n))                               # :96
            f_den = mul(f_den, emb(d))                               # :97
            R = jac_add(_F2, R, Q)                                   # :98
    if generic_fp12:
        di = fp12_inv(f_den)
    else:
        di = fp2_inv(f_den)
    if di is None:                                                   # :102 panics in the reference
        return FP12_ZERO, FLAG_DEN_ZERO
    f = mul(f_num, di)                                               # :102
    f = mul(f, f)                                                    # :116-121 exactly one squaring
    if not generic_fp12:
        f = fp12_from_fp2(f)
    return f, flags


def optimized_miller_loop_strength_reduced(Pt, Q):
    """The same function with the three strength reductions the device kernel applies to the reference's generic `*` / `+`
    (none can change a field value): `f * f` by the Fq12 squaring formula, `f * embed(v)` as six Fp2 products,
    `R + R` straight to the doubling formulas.  Used to check value equality with optimized_miller_loop and to count the Fp
    multiplications the kernel's formulation needs (`python oracle/pyref.py`)."""
    def scale(f, v):
        (a0, a1, a2), (b0, b1, b2) = f
        return ((fp2_mul(a0, v), fp2_mul(a1, v), fp2_mul(a2, v)), (fp2_mul(b0, v), fp2_mul(b1, v), fp2_mul(b2, v)))
    R = Q
    f_num, f_den = FP12_ONE, FP12_ONE
    for v in PSEUDO_BINARY_ENCODING:
        n, d, _ = optimized_line_function(R, R, Pt)
        f_num = scale(fp12_sqr(f_num), n)
        f_den = scale(fp12_sqr(f_den), d)
        R = jac_double(_F2, R)
        if v == 1:
            n, d, _ = optimized_line_function(R, Q, Pt)
            f_num = scale(f_num, n)
            f_den = scale(f_den, d)
            R = jac_add(_F2, R, Q)
    di = fp12_inv(f_den)
    if di is None:
        return FP12_ZERO, FLAG_DEN_ZERO
    f = fp12_mul(f_num, di)
    return fp12_mul(f, f), 0


# ======================================================================= MODE B
# src/miller_loop_native.rs (+ src/fields_as_trees/miller_loop.rs for the final exponentiation)

def point_doubling_and_line_evaluation(r):
    """src/miller_loop_native.rs:27-55 (Alg. 26 of ePrint 2010/354).  r = [X, Y, Z] (list, in place)."""
    X, Y, Z = r
    tmp0 = fp2_sqr(X)
    tmp1 = fp2_sqr(Y)
    tmp2 = fp2_sqr(tmp1)
    tmp3 = fp2_sub(fp2_sub(fp2_sqr(fp2_add(tmp1, X)), tmp0), tmp2)
    tmp3 = fp2_add(tmp3, tmp3)
    tmp4 = fp2_add(fp2_add(tmp0, tmp0), tmp0)
    tmp6 = fp2_add(X, tmp4)
    tmp5 = fp2_sqr(tmp4)
    zsquared = fp2_sqr(Z)
    X3 = fp2_sub(fp2_sub(tmp5, tmp3), tmp3)
    Z3 = fp2_sub(fp2_sub(fp2_sqr(fp2_add(Z, Y)), tmp1), zsquared)
    Y3 = fp2_mul(fp2_sub(tmp3, X3), tmp4)
    tmp2 = fp2_dbl(fp2_dbl(fp2_dbl(tmp2)))
    Y3 = fp2_sub(Y3, tmp2)
    tmp3 = fp2_neg(fp2_dbl(fp2_mul(tmp4, zsquared)))                 # :44-46 (negation)
    tmp6 = fp2_sub(fp2_sub(fp2_sqr(tmp6), tmp0), tmp5)
    tmp1 = fp2_dbl(fp2_dbl(tmp1))
    tmp6 = fp2_sub(tmp6, tmp1)
    tmp0 = fp2_dbl(fp2_mul(Z3, zsquared))
    r[0], r[1], r[2] = X3, Y3, Z3
    return (tmp0, tmp3, tmp6)


def point_addition_and_line_evaluation(r, q):
    """src/miller_loop_native.rs:58-87 (Alg. 27).  q = (xq, yq) affine."""
    X, Y, Z = r
    xq, yq = q
    zsquared = fp2_sqr(Z)
    ysquared = fp2_sqr(yq)
    t0 = fp2_mul(zsquared, xq)
    t1 = fp2_mul(fp2_sub(fp2_sub(fp2_sqr(fp2_add(yq, Z)), ysquared), zsquared), zsquared)
    t2 = fp2_sub(t0, X)
    t3 = fp2_sqr(t2)
    t4 = fp2_dbl(fp2_dbl(t3))
    t5 = fp2_mul(t4, t2)
    t6 = fp2_sub(fp2_sub(t1, Y), Y)
    t9 = fp2_mul(t6, xq)
    t7 = fp2_mul(t4, X)
    X3 = fp2_sub(fp2_sub(fp2_sub(fp2_sqr(t6), t5), t7), t7)
    Z3 = fp2_sub(fp2_sub(fp2_sqr(fp2_add(Z, t2)), zsquared), t3)
    t10 = fp2_add(yq, Z3)
    t8 = fp2_mul(fp2_sub(t7, X3), t6)
    t0 = fp2_dbl(fp2_mul(Y, t5))
    Y3 = fp2_sub(t8, t0)
    t10 = fp2_sub(fp2_sqr(t10), ysquared)
    ztsquared = fp2_sqr(Z3)
    t10 = fp2_sub(t10, ztsquared)
    t9 = fp2_sub(fp2_dbl(t9), t10)
    t10 = fp2_dbl(Z3)
    t6 = fp2_neg(t6)                                                  # :83 (negation)
    t1 = fp2_dbl(t6)
    r[0], r[1], r[2] = X3, Y3, Z3
    return (t10, t1, t9)


def ell(f, coeffs, p_aff):
    """Commented src/miller_loop_native.rs:139-152: c0 = coeffs.0 * p.y, c1 = coeffs.1 * p.x,
    f.mul_by_014(coeffs.2, c1, c0)."""
    c0 = fp2_mul_fp(coeffs[0], p_aff[1])
    c1 = fp2_mul_fp(coeffs[1], p_aff[0])
    return fp12_mul_by_014(f, coeffs[2], c1, c0)


def g2_prepare(q_aff):
    """The 68 line-coefficient triples (G2Prepared; src/miller_loop_target.rs:23-76 states the shape,
    src/miller_loop_native.rs:89-116 the schedule)."""
    coeffs = []
    r = [q_aff[0], q_aff[1], FP2_ONE]
    found_one = False
    for b in range(63, -1, -1):
        bit = ((BLS_X >> 1) >> b) & 1
        if not found_one:
            found_one = bit == 1
            continue
        coeffs.append(point_doubling_and_line_evaluation(r))
        if bit:
            coeffs.append(point_addition_and_line_evaluation(r, q_aff))
    coeffs.append(point_doubling_and_line_evaluation(r))
    assert len(coeffs) == 68                                          # src/miller_loop_target.rs:69
    return coeffs


def multi_miller_loop_literal(terms):
    """What src/miller_loop_native.rs:154-212 computes TODAY: the driver's line steps are empty
    (:163-188), so the loop squares/conjugates one -> Fq12::one() for every input."""
    f = FP12_ONE
    found_one = False
    for b in range(63, -1, -1):
        bit = ((BLS_X >> 1) >> b) & 1
        if not found_one:
            found_one = bit == 1
            continue
        f = fp12_sqr(f)
    if BLS_X_IS_NEGATIVE:
        f = fp12_conj(f)
    return f


def multi_miller_loop(terms):
    """Intended behaviour of src/miller_loop_native.rs:89-116 + :154-212 with `ell` live:
    terms = [(P_affine | None, Q_affine | None), ...]; None = point at infinity (skipped, like the
    commented `either_identity` select at :167-170).  Schedule per iteration: doubling step for
    every term, addition step if the bit is set, then ONE shared squaring (:100-106)."""
    live = [(p, q) for p, q in terms if p is not None and q is not None]
    rs = [[q[0], q[1], FP2_ONE] for _, q in live]
    f = FP12_ONE
    found_one = False
    for b in range(63, -1, -1):
        bit = ((BLS_X >> 1) >> b) & 1
        if not found_one:
            found_one = bit == 1
            continue
        for (p, q), r in zip(live, rs):
            f = ell(f, point_doubling_and_line_evaluation(r), p)
        if bit:
            for (p, q), r in zip(live, rs):
                f = ell(f, point_addition_and_line_evaluation(r, q), p)
        f = fp12_sqr(f)
    for (p, q), r in zip(live, rs):
        f = ell(f, point_doubling_and_line_evaluation(r), p)
    if BLS_X_IS_NEGATIVE:
        f = fp12_conj(f)
    return f


def miller_loop(p_aff, q_aff):
    return multi_miller_loop([(p_aff, q_aff)])


def fq4_square(a, b):
    """src/fields_as_trees/miller_loop.rs:29-44."""
    t0 = fp2_sqr(a)
    t1 = fp2_sqr(b)
    c0 = fp2_add(fp2_mul_by_nonresidue(t1), t0)
    t2 = fp2_sqr(fp2_add(a, b))
    c1 = fp2_sub(fp2_sub(t2, t0), t1)
    return c0, c1


def cyclotomic_square(f):
    """src/fields_as_trees/miller_loop.rs:46-104."""
    z0, z4, z3 = f[0]
    z2, z1, z5 = f[1]
    t0, t1 = fq4_square(z0, z1)
    z0 = fp2_add(fp2_dbl(fp2_sub(t0, z0)), t0)
    z1 = fp2_add(fp2_dbl(fp2_add(t1, z1)), t1)
    t0, t1 = fq4_square(z2, z3)
    t2, t3 = fq4_square(z4, z5)
    z4 = fp2_add(fp2_dbl(fp2_sub(t0, z4)), t0)
    z5 = fp2_add(fp2_dbl(fp2_add(t1, z5)), t1)
    t0 = fp2_mul_by_nonresidue(t3)
    z2 = fp2_add(fp2_dbl(fp2_add(t0, z2)), t0)
    z3 = fp2_add(fp2_dbl(fp2_sub(t2, z3)), t2)
    return ((z0, z4, z3), (z2, z1, z5))


def cyclotomic_exp(f):
    """src/fields_as_trees/miller_loop.rs:106-126 with the dropped product at :121 kept
    (`tmp = tmp * f`), as SURVEY FE2 requires.  Returns conj(f^|x|)."""
    tmp = FP12_ONE
    found_one = False
    for b in range(63, -1, -1):
        bit = (BLS_X >> b) & 1
        if found_one:
            tmp = cyclotomic_square(tmp)
        else:
            found_one = bit == 1
        if bit:
            tmp = fp12_mul(tmp, f)
    return fp12_conj(tmp)


def final_exponentiation(f):
    """src/fields_as_trees/miller_loop.rs:166-178 (easy part) + f_conversion :128-164 (hard part).
    Returns None when f == 0 (the circuit's `inv` has no witness then)."""
    t0 = f
    for _ in range(6):
        t0 = fp12_frobenius(t0)                                       # :168-174 (== conjugate)
    t1 = fp12_inv(f)                                                  # :175
    if t1 is None:
        return None
    t2 = fp12_mul(t0, t1)                                             # :133
    t1 = t2
    t2 = fp12_frobenius(fp12_frobenius(t2))                           # :135
    t2 = fp12_mul(t2, t1)                                             # :136
    t1 = fp12_conj(cyclotomic_square(t2))                             # :137
    t3 = cyclotomic_exp(t2)
    t4 = cyclotomic_square(t3)
    t5 = fp12_mul(t1, t3)
    t1 = cyclotomic_exp(t5)
    t0 = cyclotomic_exp(t1)
    t6 = cyclotomic_exp(t0)
    t6 = fp12_mul(t6, t4)
    t4 = cyclotomic_exp(t6)
    t5 = fp12_conj(t5)
    t4 = fp12_mul(t4, fp12_mul(t5, t2))
    t5 = fp12_conj(t2)
    t1 = fp12_mul(t1, t2)
    t1 = fp12_frobenius(fp12_frobenius(fp12_frobenius(t1)))
    t6 = fp12_mul(t6, t5)
    t6 = fp12_frobenius(t6)
    t3 = fp12_mul(t3, t0)
    t3 = fp12_frobenius(fp12_frobenius(t3))
    t3 = fp12_mul(t3, t1)
    t3 = fp12_mul(t3, t6)
    return fp12_mul(t3, t4)


def pairing(p_aff, q_aff):
    return final_exponentiation(miller_loop(p_aff, q_aff))


def multi_pairing(terms):
    return final_exponentiation(multi_miller_loop(terms))


# ================================================================ witness-generator hooks
# The native arithmetic plonky2 runs while filling a witness.

WFLAG_NO_INVERSE, WFLAG_NOT_SQUARE, WFLAG_SIGN_UNSATISFIABLE = 4, 8, 16


def sgn0_fq(x):
    """src/fields/helpers.rs:159-167: lowest bit of the canonical integer (false for zero)."""
    return (x % P) & 1 == 1


def sgn0_fq2(x):
    """src/fields/helpers.rs:169-174."""
    return sgn0_fq(x[0]) or (x[0] % P == 0 and sgn0_fq(x[1]))


def get_naf(exp_words):
    """src/fields/helpers.rs:197-239, literal restatement (exp as little-endian u64 words)."""
    exp = list(exp_words)
    ln = len(exp)
    naf = []
    for idx in range(ln):
        e = exp[idx]
        for _ in range(64):
            if e & 1:
                z = 2 - (e % 4)
                e //= 2
                if z == -1:
                    e += 1
                naf.append(z)
            else:
                naf.append(0)
                e //= 2
        if e != 0:
            assert e == 1
            j = idx + 1
            while j < len(exp) and exp[j] == 2 ** 64 - 1:
                exp[j] = 0
                j += 1
            if j < len(exp):
                exp[j] += 1
            else:
                exp.append(1)
    if len(exp) != ln:
        naf.append(1)
    return naf


def pow_fq(a, exp_words):
    """src/fields/helpers.rs:176-195 -> (value, flags).  Division by a for -1 digits: a == 0 panics there."""
    a %= P
    res, started = a, False
    for z in reversed(get_naf(exp_words)):
        if started:
            res = res * res % P
        if z != 0:
            if started:
                if z == 1:
                    res = res * a % P
                else:
                    if a == 0:
                        return 0, WFLAG_NO_INVERSE
                    res = res * pow(a, P - 2, P) % P
            else:
                assert z == 1
                started = True
    return res, 0


def legendre(a):
    """FqTarget::legendre, src/fields/fq_target.rs:271-274: a^((p-1)/2)."""
    return pow(a % P, (P - 1) // 2, P)


def fp_sqrt_any(x):
    """Some square root of x or None (p = 3 mod 4: x^((p+1)/4))."""
    s = pow(x, (P + 1) // 4, P)
    return s if (s * s - x) % P == 0 else None


def fp2_sqrt_any(x):
    """Some square root of x in Fp2 or None, by brute algebra independent of the device algorithm:
    solve (c0 + c1 u)^2 = a + b u  <=>  c0^2 - c1^2 = a, 2 c0 c1 = b."""
    a, b = x[0] % P, x[1] % P
    if b == 0:
        s = fp_sqrt_any(a)
        if s is not None:
            return (s, 0)
        t = fp_sqrt_any((-a) % P)
        return None if t is None else (0, t)
    n = (a * a + b * b) % P
    sn = fp_sqrt_any(n)
    if sn is None:
        return None
    inv2 = pow(2, P - 2, P)
    for d in (((a + sn) * inv2) % P, ((a - sn) * inv2) % P):
        c0 = fp_sqrt_any(d)
        if c0 is not None and c0 != 0:
            c1 = (b * pow(2 * c0, P - 2, P)) % P
            if ((c0 * c0 - c1 * c1 - a) % P, (2 * c0 * c1 - b) % P) == (0, 0):
                return (c0, c1)
    return None


def witness_fq_sqrt(x, sgn):
    """FqSqrtGenerator::run_once, src/fields/fq_target.rs:330-343 -> (root, flags)."""
    s = fp_sqrt_any(x % P)
    if s is None:
        return 0, WFLAG_NOT_SQUARE                       # sqrt().unwrap() panics
    if sgn0_fq(s) != bool(sgn & 1):
        s = (-s) % P
    if sgn0_fq(s) != bool(sgn & 1):
        return s, WFLAG_SIGN_UNSATISFIABLE               # assert_eq! at :340 fails (only for the root 0)
    return s, 0


def witness_fq2_sqrt(x, sgn):
    """Fq2SqrtGenerator::run_once, src/fields/fq2_target.rs:385-410 -> (root, flags)."""
    s = fp2_sqrt_any(x)
    if s is None:
        return (0, 0), WFLAG_NOT_SQUARE
    if sgn0_fq2(s) != bool(sgn & 1):
        s = ((-s[0]) % P, (-s[1]) % P)
    if sgn0_fq2(s) != bool(sgn & 1):
        return s, WFLAG_SIGN_UNSATISFIABLE
    return s, 0


def jac_in_subgroup(F, pt):
    """On the curve and [r] pt == infinity (arkworks `is_in_correct_subgroup_assuming_on_curve` after `is_on_curve`
    [third party]; r: the decimal at src/miller_loop_native_optimized.rs:110)."""
    if not jac_on_curve(F, pt):
        return False
    return F.is_zero(jac_scalar_mul(F, pt, R_ORDER)[2])


def decompress_point(group, x, sgn):
    """Affine point with abscissa x and sgn0(y) == sgn & 1, from the reference's pieces: curve equation
    (src/curves/g{1,2}_curve_target.rs), FqSqrtGenerator / Fq2SqrtGenerator roots and sgn0 (src/fields/helpers.rs:159-174).
    Returns (point | None, flags); sgn & 2 = infinity."""
    if sgn & 2:
        return None, FLAG_INFINITY
    if group == 1:
        rhs = (pow(x, 3, P) + 4) % P
        y, fl = witness_fq_sqrt(rhs, sgn)
    else:
        rhs = fp2_add(fp2_mul(fp2_sqr(x), x), (4, 4))
        y, fl = witness_fq2_sqrt(rhs, sgn)
    if fl & WFLAG_NOT_SQUARE:
        return None, fl
    return (x, y), fl


def witness_fq2_inverse(x):
    """Fq2InverseGenerator::run_once, src/fields/fq2_target.rs:331-352: None -> zero."""
    r = fp2_inv((x[0] % P, x[1] % P))
    return (0, 0) if r is None else r


def fp6_to_myfq6(a):
    """src/fields/my_fq6.rs:12-27: coeffs = [c00, c10, c20, c01, c11, c21]."""
    return [a[0][0], a[1][0], a[2][0], a[0][1], a[1][1], a[2][1]]


def myfq6_to_fp6(c):
    return ((c[0], c[3]), (c[1], c[4]), (c[2], c[5]))


def myfq12_to_fp12(c):
    """src/fields/helpers.rs:45-76."""
    q = lambda k: (c[k], c[k + 6])
    return ((q(0), q(2), q(4)), (q(1), q(3), q(5)))


def witness_fq6_inverse(coeffs):
    """Fq6InverseGenerator::run_once, src/fields/fq6_target.rs:395-418 on MyFq6 coefficients -> (coeffs, flags)."""
    r = fp6_inv(myfq6_to_fp6([c % P for c in coeffs]))
    if r is None:
        return [0] * 6, WFLAG_NO_INVERSE                 # inverse().unwrap() panics
    return fp6_to_myfq6(r), 0


def witness_fq12_inverse(coeffs):
    """Fq12InverseGenerator::run_once, src/fields/fq12_target.rs:351-374 on MyFq12 coefficients."""
    r = fp12_inv(myfq12_to_fp12([c % P for c in coeffs]))
    if r is None:
        return [0] * 12, WFLAG_NO_INVERSE
    return fp12_to_myfq12(r), 0


# ----------------------------------------------------------------- synthetic batches
SEED = 0x424C533132333831      # ASCII "BLS12381" 


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def prng_fp(seed, index, lane):
    """Counter-based field element: 6 splitmix64 words reduced mod p (never 0 in practice)."""
    v = 0
    for k in range(6):
        v |= splitmix64(seed ^ ((index * 16 + lane) * 8 + k) * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF) << (64 * k)
    v %= P
    return v if v else 1


if __name__ == "__main__":
    import json
    import time

    OPS.reset()
    t = time.time()
    out, fl = optimized_miller_loop((G1_GEN[0], G1_GEN[1], 1), (G2_GEN[0], G2_GEN[1], FP2_ONE))
    a_ops = OPS.snapshot()
    print("mode A generator c0.c0 =", hex(out[0][0][0]), hex(out[0][0][1]), "flags", fl, "%.2fs" % (time.time() - t))
    OPS.reset()
    ml = miller_loop(G1_GEN, G2_GEN)
    m_ops = OPS.snapshot()
    OPS.reset()
    e = final_exponentiation(ml)
    f_ops = OPS.snapshot()
    OPS.reset()
    out2, fl2 = optimized_miller_loop_strength_reduced((G1_GEN[0], G1_GEN[1], 1), (G2_GEN[0], G2_GEN[1], FP2_ONE))
    a2_ops = OPS.snapshot()
    assert out2 == out and fl2 == fl
    # the k = 3 product (BASELINE config 4): three pairs through one shared loop, then ONE final exponentiation
    OPS.reset()
    g1s = [g1_mul(k) for k in (1, 2, 3)]
    g2s = [g2_mul(k) for k in (4, 5, 6)]
    OPS.reset()
    ml3 = multi_miller_loop(list(zip(g1s, g2s)))
    m3_ops = OPS.snapshot()
    print(json.dumps({"modeA_generic": a_ops, "modeB_miller": m_ops, "modeB_finalexp": f_ops, "modeB_miller_k3": m3_ops,
                      "modeA_strength_reduced": a2_ops}))
