"""One-off randomized stress: larger random batches through the main entry points against the C oracle
(Mode A both variants, Miller, pairing, k = 2/3 products, prepared shapes).  Developer tool, run on the GPU box."""
import os, sys, time, random
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from blsutil import *
import bls381_amd as B

eng = B.Engine(0); orc = COracle()
rng = random.Random(int(time.time()))
seed = rng.randrange(1 << 30)
print("seed", seed, flush=True)
N = 6000
# distinct random elements: pool points with fresh random Jacobian representatives / random pool picks
pairs = random_pairs(N, seed, jacobian=True)
g1 = enc_fps([c for P, Q in pairs for c in flat_g1_jac(P)]); g2 = enc_fps([c for P, Q in pairs for c in flat_g2_jac(Q)])
exp, efl = orc.miller_ref_opt(g1, g2, threads=16)
for sc in (0, 1):
    eng.set_option(B.OPT_REF_OPT_FP2_SHORTCUT, sc)
    out, fl = eng.optimized_miller_loop(g1, g2)
    assert np.array_equal(out, exp) and np.array_equal(fl, efl), ("mode A", sc)
eng.set_option(B.OPT_REF_OPT_FP2_SHORTCUT, 0)
print("mode A ok", flush=True)
# garbage (not-on-curve) inputs: pure arithmetic parity incl. rare branches
garb1 = np.frombuffer(np.random.default_rng(seed).bytes(N * 144), dtype=np.uint64).copy(); garb1[5::6] &= np.uint64(0x0fffffffffffffff)
garb2 = np.frombuffer(np.random.default_rng(seed + 1).bytes(N * 288), dtype=np.uint64).copy(); garb2[5::6] &= np.uint64(0x0fffffffffffffff)
exp, efl = orc.miller_ref_opt(garb1, garb2, threads=16)
out, fl = eng.optimized_miller_loop(garb1, garb2)
assert np.array_equal(out, exp) and np.array_equal(fl, efl), "mode A garbage"
print("mode A garbage ok", flush=True)
for k in (1, 2, 3):
    n = N // k
    a = np.frombuffer(np.random.default_rng(seed + 10 + k).bytes(n * k * 96), dtype=np.uint64).copy(); a[5::6] &= np.uint64(0x0fffffffffffffff)
    b = np.frombuffer(np.random.default_rng(seed + 20 + k).bytes(n * k * 192), dtype=np.uint64).copy(); b[5::6] &= np.uint64(0x0fffffffffffffff)
    exp, efl = orc.multi(a, b, k, final_exp=True, threads=16)
    # off the curve the homogeneous Miller steps and Alg. 26/27 are different functions (their lines are proportional only
    # through the curve equation): arithmetic parity on garbage is a statement about the Alg. 26/27 path
    eng.set_option(B.OPT_PAIRING_HOMOGENEOUS_MILLER, 0)
    out, fl = eng.multi_pairing(a, b, k)
    eng.set_option(B.OPT_PAIRING_HOMOGENEOUS_MILLER, 1)
    assert np.array_equal(out, exp) and np.array_equal(fl, efl), ("pairing garbage", k)
    expm, _ = orc.multi(a, b, k, final_exp=False, threads=16)
    outm, _ = eng.multi_miller_loop(a, b, k)
    assert np.array_equal(outm, expm), ("miller garbage", k)
print("mode B garbage ok", flush=True)
ap = random_pairs(N, seed + 3, jacobian=False)
a = enc_fps([c for P, Q in ap for c in flat_g1_aff(P)]); b = enc_fps([c for P, Q in ap for c in flat_g2_aff(Q)])
exp, efl = orc.multi(a, b, 1, final_exp=True, threads=16)
out, fl = eng.multi_pairing(a, b, 1)
assert np.array_equal(out, exp)
print("mode B curve points ok", flush=True)
x = np.frombuffer(np.random.default_rng(seed + 5).bytes(N * 576), dtype=np.uint64).copy(); x[5::6] &= np.uint64(0x0fffffffffffffff)
expx = orc.final_exp(x, threads=16)
for mode in (B.FEXP_SELF_CONTAINED, B.FEXP_BATCHED_INVERSION, B.FEXP_STAGED, B.FEXP_AUTO):     # arbitrary (not unitary) Fq12 values
    eng.set_option(B.OPT_FEXP_MODE, mode)
    assert np.array_equal(eng.final_exponentiation(x), expx), ("final exponentiation", mode)
eng.set_option(B.OPT_FEXP_MODE, B.FEXP_STAGED)
out, fl = eng.multi_pairing(a, b, 1)
assert np.array_equal(out, exp), "pairing, staged final exponentiation"
eng.set_option(B.OPT_FEXP_MODE, B.FEXP_AUTO)
for op in (0, 1, 2, 3, 4):
    assert np.array_equal(eng.fp12_op(op, x, x[::-1].copy() if False else x), orc.fp12_op(op, x, x)), op
print("fp12 ok; ALL OK", flush=True)
