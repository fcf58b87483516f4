"""TEST INFRASTRUCTURE.  A textbook ate pairing on BLS12-381, written independently of the formulas the reference (and
therefore oracle/ and the kernels) uses: the G2 point is untwisted into E(Fp12), the Miller function f_{|x|, Q}(P) is
accumulated with plain affine chord-and-tangent lines over Fp12 (slope = 3x^2/2y or dy/dx, one Fp12 inversion per step),
and the final exponentiation is ONE big-integer power (p^12 - 1)/r.  No Jacobian coordinates, no sparse products, no
cyclotomic tricks, no Frobenius maps.  It shares only the tower's multiplication / inversion with oracle/pyref.py, and
those are pinned by the reference's own vectors (tests/test_oracle_kats.py).

Used to pin Mode B (the reference holds no pairing vector): the reference's final exponentiation raises to
3 (p^12 - 1)/r (its hard part is the (x-1)^2 (x+p)(x^2+p^2-1) + 3 chain), so

    pairing_reference(P, Q) == textbook_ate(P, Q) ** 3 .
"""
from blsutil import O

ZERO2, ONE2 = (0, 0), (1, 0)
FP12_ONE = ((ONE2, ZERO2, ZERO2), (ZERO2, ZERO2, ZERO2))
W = ((ZERO2, ZERO2, ZERO2), (ONE2, ZERO2, ZERO2))           # w, with w^2 = v, v^3 = xi = 1 + u, so w^6 = xi


def _from_fp(a):
    return (((a % O.P, 0), ZERO2, ZERO2), (ZERO2, ZERO2, ZERO2))


def _from_fp2(a):
    return ((a, ZERO2, ZERO2), (ZERO2, ZERO2, ZERO2))


def _sub(a, b):
    return tuple(tuple(O.fp2_sub(x, y) for x, y in zip(h, k)) for h, k in zip(a, b))


def _add(a, b):
    return tuple(tuple(O.fp2_add(x, y) for x, y in zip(h, k)) for h, k in zip(a, b))


def untwist(q_aff):
    """E'(Fp2): y^2 = x^3 + 4 xi  ->  E(Fp12): y^2 = x^3 + 4,  (x', y') -> (x' / w^2, y' / w^3)."""
    w2 = O.fp12_mul(W, W)
    w3 = O.fp12_mul(w2, W)
    x = O.fp12_mul(_from_fp2(q_aff[0]), O.fp12_inv(w2))
    y = O.fp12_mul(_from_fp2(q_aff[1]), O.fp12_inv(w3))
    # the image really is on E: y^2 = x^3 + 4
    assert O.fp12_eq(O.fp12_mul(y, y), _add(O.fp12_mul(O.fp12_mul(x, x), x), _from_fp(4)))
    return x, y


def miller_function(p_aff, q_aff, loop=None):
    """f_{|x|, psi(Q)}(P), MSB-first double-and-add, vertical lines omitted (their values lie in Fp6)."""
    loop = O.BLS_X if loop is None else loop
    xq, yq = untwist(q_aff)
    xp, yp = _from_fp(p_aff[0]), _from_fp(p_aff[1])
    xt, yt = xq, yq
    f = FP12_ONE

    def line_and_step(xt, yt, lam):
        # l(P) = (yP - yT) - lam (xP - xT);   T' = (lam^2 - xT - x2, lam (xT - x3) - yT)
        return _sub(_sub(yp, yt), O.fp12_mul(lam, _sub(xp, xt)))

    for bit in bin(loop)[3:]:
        three_x2 = O.fp12_mul(_from_fp(3), O.fp12_mul(xt, xt))
        lam = O.fp12_mul(three_x2, O.fp12_inv(_add(yt, yt)))
        f = O.fp12_mul(O.fp12_mul(f, f), line_and_step(xt, yt, lam))
        x3 = _sub(O.fp12_mul(lam, lam), _add(xt, xt))
        yt = _sub(O.fp12_mul(lam, _sub(xt, x3)), yt)
        xt = x3
        if bit == "1":
            lam = O.fp12_mul(_sub(yq, yt), O.fp12_inv(_sub(xq, xt)))
            f = O.fp12_mul(f, line_and_step(xt, yt, lam))
            x3 = _sub(_sub(O.fp12_mul(lam, lam), xt), xq)
            yt = _sub(O.fp12_mul(lam, _sub(xt, x3)), yt)
            xt = x3
    return f


def textbook_ate(p_aff, q_aff):
    """a(P, Q) = f_{x, psi(Q)}(P) ^ ((p^12 - 1) / r) with x < 0: f_{x} = 1 / f_{|x|} up to factors the exponent kills."""
    f = O.fp12_inv(miller_function(p_aff, q_aff))
    return O.fp12_pow(f, (O.P ** 12 - 1) // O.R_ORDER)
