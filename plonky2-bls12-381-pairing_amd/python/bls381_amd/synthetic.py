"""Synthetic (G1, G2) batches as SURVEY.md 8(d) specifies them, built with the engine's own point kernels (no oracle):

* pools of K subgroup points  P_k = (k+1) G1,  Q_k = (k+1) G2  (K = 4096 by default), grown by doubling the pool with one
  batched Jacobian addition per step (`bls381_g{1,2}_op_batch`, POINT_ADD) and normalised once (POINT_NORMALIZE);
* element i of the batch uses  P_{i mod K}  and  Q_{(i * 2654435761 >> 12) mod K}  -- K^2 distinct pairings are possible,
  a 2^20 batch holds 2^20 distinct ones;
* Mode A (`optimized_miller_loop`, general-Z inputs like the reference's `G*Projective::rand`, opt.rs:141-142): every
  element is re-randomised to the Jacobian representative (l^2 X, l^3 Y, l Z) with l from splitmix64(seed, i).

Element encoding = the ABI's: 6 x u64 little-endian limbs per Fp, arkworks Montgomery form (R = 2^384).
"""
from __future__ import annotations

import numpy as np

from . import POINT_ADD, POINT_NORMALIZE, FP_MUL, FP_SQR, FP2_MUL, FP2_SQR, OPT_IO_FORM, IO_ARK_MONT

SEED = 0x424C533132333831          # ASCII "BLS12381" 
POOL = 4096
HASH_MUL = 2654435761
P_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
# the standard generators (canonical integers); src/curves/g1_curve_target.rs:42-59 and g2_curve_target.rs:69-112 hold the
# same numbers as Montgomery limbs (tests/test_oracle_kats.py, KAT-3)
G1_X = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
G1_Y = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
G2_X = (0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
        0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e)
G2_Y = (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
        0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)


def mont_limbs(v: int) -> list[int]:
    """canonical integer -> arkworks in-memory limbs of v * 2^384 mod p"""
    m = (v << 384) % P_MOD
    return [(m >> (64 * i)) & (2 ** 64 - 1) for i in range(6)]


def _jac(coords: list[int]) -> np.ndarray:
    return np.array([w for c in coords for w in mont_limbs(c)], dtype=np.uint64)


def build_pools(eng, size: int = POOL):
    """(P_k), (Q_k) for k < size as AFFINE arrays [size, 12] / [size, 24] (uint64), computed on the GPU."""
    if eng.get_option(OPT_IO_FORM) != IO_ARK_MONT:
        raise ValueError("build_pools: engine must use the arkworks Montgomery encoding")
    pools = []
    for group, gen, w in ((1, _jac([G1_X, G1_Y, 1]), 18), (2, _jac([*G2_X, *G2_Y, 1, 0]), 36)):
        pool = gen.reshape(1, w)
        while pool.shape[0] < size:
            m = pool.shape[0]                                   # pool[m-1] = m * G, so pool[i] + pool[m-1] = (i + 1 + m) G
            shift = np.ascontiguousarray(np.broadcast_to(pool[m - 1], (m, w))).reshape(-1)
            more, _ = eng.point_op(group, POINT_ADD, pool.reshape(-1), shift)
            pool = np.concatenate([pool, more.reshape(m, w)])
        pool = pool[:size]
        norm, fl = eng.point_op(group, POINT_NORMALIZE, np.ascontiguousarray(pool).reshape(-1))
        if fl.any():
            raise RuntimeError("build_pools: a pool point came out at infinity")
        pools.append(np.ascontiguousarray(norm.reshape(size, w)[:, : 2 * w // 3]))
    return pools[0], pools[1]


def element_indices(n: int, size: int = POOL, first: int = 0):
    """pool indices (into P, into Q) of batch elements first .. first + n - 1"""
    i = np.arange(first, first + n, dtype=np.uint64)
    return (i % np.uint64(size)).astype(np.int64), (((i * np.uint64(HASH_MUL)) >> np.uint64(12)) % np.uint64(size)).astype(np.int64)


def affine_batch(pools, n: int, k: int = 1, first: int = 0):
    """n units of k (P, Q) pairs in ABI layout (n*k*12 / n*k*24 u64); pair j of unit i is element (first + i) * k + j."""
    p1, p2 = pools
    ia, ib = element_indices(n * k, p1.shape[0], first * k)
    return p1[ia].reshape(-1), p2[ib].reshape(-1)


def mirrored_affine_batch(pools, n: int, first: int = 0):
    """A batch whose second half pairs the same multiples the other way round: element i + n/2 is (P_b, Q_a) where element i
    is (P_a, Q_b).  e((a+1) G1, (b+1) G2) = e((b+1) G1, (a+1) G2), so out[i] == out[i + n/2] is a full-size check that needs
    no oracle (n even; 2^20 elements hold 2^19 distinct pairings)."""
    assert n % 2 == 0
    p1, p2 = pools
    ia, ib = element_indices(n // 2, p1.shape[0], first)
    return np.concatenate([p1[ia], p1[ib]]).reshape(-1), np.concatenate([p2[ib], p2[ia]]).reshape(-1)


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def random_fp(n: int, seed: int, stream: int) -> np.ndarray:
    """n field elements as 6 x u64 patterns below 2^380 (the engine reduces them; every value mod p is hit about equally)"""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(6 * 16) + np.uint64((seed + 0x1000003 * stream) & (2 ** 64 - 1))
        limbs = np.stack([splitmix64(idx + np.uint64(j)) for j in range(6)], axis=1)
    limbs[:, 5] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    limbs[:, 0] |= np.uint64(1)                                   # never the zero pattern
    return limbs.reshape(-1)


def jacobian_batch(eng, pools, n: int, first: int = 0, seed: int = SEED):
    """Mode-A inputs: element i of affine_batch re-randomised to (l^2 X, l^3 Y, l) with its own l (Fp for G1, Fp2 for G2).
    Returns (g1 n*18, g2 n*36 u64)."""
    a1, a2 = affine_batch(pools, n, 1, first)
    a1 = a1.reshape(n, 2, 6); a2 = a2.reshape(n, 2, 12)
    one = np.array(mont_limbs(1), dtype=np.uint64)
    lam = random_fp(n, seed + first, 1)
    l1 = eng.fp_op(FP_MUL, lam, np.tile(one, n))                  # reduced representative of l
    l2 = eng.fp_op(FP_SQR, l1)
    l3 = eng.fp_op(FP_MUL, l2, l1)
    g1 = np.stack([eng.fp_op(FP_MUL, np.ascontiguousarray(a1[:, 0]).reshape(-1), l2).reshape(n, 6),
                   eng.fp_op(FP_MUL, np.ascontiguousarray(a1[:, 1]).reshape(-1), l3).reshape(n, 6), l1.reshape(n, 6)], axis=1)
    one2 = np.concatenate([one, np.zeros(6, dtype=np.uint64)])
    mu = np.stack([random_fp(n, seed + first, 2).reshape(n, 6), random_fp(n, seed + first, 3).reshape(n, 6)], axis=1).reshape(-1)
    m1 = eng.fp2_op(FP2_MUL, mu, np.tile(one2, n))
    m2 = eng.fp2_op(FP2_SQR, m1)
    m3 = eng.fp2_op(FP2_MUL, m2, m1)
    g2 = np.stack([eng.fp2_op(FP2_MUL, np.ascontiguousarray(a2[:, 0]).reshape(-1), m2).reshape(n, 12),
                   eng.fp2_op(FP2_MUL, np.ascontiguousarray(a2[:, 1]).reshape(-1), m3).reshape(n, 12), m1.reshape(n, 12)], axis=1)
    return np.ascontiguousarray(g1).reshape(-1), np.ascontiguousarray(g2).reshape(-1)
