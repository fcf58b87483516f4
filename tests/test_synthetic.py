"""bls381_amd.synthetic (the bench's input generator): host-side pieces that need no GPU -- index hashing,
generator encodings against the oracle's constants, the PRNG -- and the pool construction driven through a stand-in for the
engine's point kernels (the oracle's Jacobian arithmetic), so the doubling scheme itself is checked on CPU."""
import numpy as np

import pyref as O
from blsutil import enc_fps, dec_fps

import bls381_amd as B
from bls381_amd import synthetic as S


def test_generators_and_modulus_match_the_oracle():
    assert S.P_MOD == O.P
    assert (S.G1_X, S.G1_Y) == tuple(O.G1_GEN)
    assert (S.G2_X, S.G2_Y) == (tuple(O.G2_GEN[0]), tuple(O.G2_GEN[1]))
    for v in (0, 1, 5, O.P - 1, S.G1_X):
        assert S.mont_limbs(v) == list(O.fp_to_limbs(v, mont=True))


def test_element_indices_follow_the_survey_formula():
    ia, ib = S.element_indices(10000, 4096, first=7)
    for t in (0, 1, 4095, 4096, 9999):
        i = 7 + t
        assert ia[t] == i % 4096 and ib[t] == ((i * 2654435761) >> 12) % 4096
    # a 2^20 batch really holds 2^20 distinct (P, Q) index pairs
    ia, ib = S.element_indices(1 << 20, 4096)
    assert len(np.unique(ia.astype(np.int64) * 4096 + ib)) == 1 << 20


def test_splitmix64_reference_values():
    # splitmix64 of the seeds 0 and 1 (state advanced once), well-known outputs
    got = S.splitmix64(np.array([0, 1], dtype=np.uint64) - np.uint64(0))
    assert int(got[0]) == 0xE220A8397B1DCDAF and int(got[1]) == 0x910A2DEC89025CC1
    r = S.random_fp(1000, S.SEED, 1).reshape(1000, 6)
    assert (r[:, 5] >> np.uint64(60) == 0).all() and len(np.unique(r[:, 0])) == 1000


class _OracleEngine:
    """stand-in for Engine.point_op built on the oracle's group law: lets build_pools run without a GPU"""

    def get_option(self, opt):
        return B.IO_ARK_MONT

    def point_op(self, group, op, a, b=None):
        F = O._F1 if group == 1 else O._F2
        w = 18 if group == 1 else 36
        n = a.size // w

        def pt(arr, i):
            f = dec_fps(arr[i * w:(i + 1) * w])
            return (f[0], f[1], f[2]) if group == 1 else ((f[0], f[1]), (f[2], f[3]), (f[4], f[5]))

        out = []
        for i in range(n):
            r = O.jac_add(F, pt(a, i), pt(b, i)) if op == B.POINT_ADD else O.jac_normalize(F, pt(a, i))
            out += list(r) if group == 1 else [c for co in r for c in co]
        return enc_fps(out), np.zeros(n, dtype=np.uint8)


def test_pool_doubling_scheme_yields_consecutive_multiples():
    p1, p2 = S.build_pools(_OracleEngine(), 8)
    for k in range(8):
        assert dec_fps(p1[k]) == list(O.g1_mul(k + 1))
        q = O.g2_mul(k + 1)
        assert dec_fps(p2[k]) == [q[0][0], q[0][1], q[1][0], q[1][1]]
    g1, g2 = S.mirrored_affine_batch((p1, p2), 16)
    a = g1.reshape(16, 12); b = g2.reshape(16, 24)
    ia, ib = S.element_indices(8, 8)
    assert np.array_equal(a[:8], p1[ia]) and np.array_equal(a[8:], p1[ib]) and np.array_equal(b[:8], p2[ib]) and np.array_equal(b[8:], p2[ia])
