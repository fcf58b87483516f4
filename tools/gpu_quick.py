"""Quick GPU sanity + timing of the HIP library against the oracle (developer tool)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from blsutil import *
import bls381_amd as B
import numpy as np

eng = B.Engine(0)
rng = random.Random(5)
xs = [rand_fp(rng) for _ in range(64)]; ys = [rand_fp(rng) for _ in range(64)]
a = enc_fps(xs); b = enc_fps(ys)
assert dec_fps(eng.fp_op(B.FP_MUL, a, b)) == [(x * y) % O.P for x, y in zip(xs, ys)]
assert dec_fps(eng.fp_op(B.FP_INV, a)) == [pow(x, O.P - 2, O.P) for x in xs]
print("fp ok", flush=True)
x12 = [rand_fp12(rng) for _ in range(4)]; y12 = [rand_fp12(rng) for _ in range(4)]
X = enc_fp12s(x12); Y = enc_fp12s(y12)
assert dec_fp12s(eng.fp12_op(B.FP12_MUL, X, Y)) == [O.fp12_mul(x, y) for x, y in zip(x12, y12)]
assert dec_fp12s(eng.fp12_op(B.FP12_INV, X)) == [O.fp12_inv(x) for x in x12]
assert dec_fp12s(eng.final_exponentiation(X)) == [O.final_exponentiation(x) for x in x12]
print("fp12 ok", flush=True)
pairs = random_pairs(8, 11, True)
g1 = enc_fps([c for P, Q in pairs for c in flat_g1_jac(P)]); g2 = enc_fps([c for P, Q in pairs for c in flat_g2_jac(Q)])
exp = [O.optimized_miller_loop(P, Q, False)[0] for P, Q in pairs]
for sc in (0, 1):
    eng.set_option(B.OPT_REF_OPT_FP2_SHORTCUT, sc)
    out, fl = eng.optimized_miller_loop(g1, g2)
    print("mode A shortcut=%d" % sc, dec_fp12s(out) == exp, fl, flush=True)
ap = random_pairs(6, 12, False)
g1 = enc_fps([c for P, Q in ap for c in flat_g1_aff(P)]); g2 = enc_fps([c for P, Q in ap for c in flat_g2_aff(Q)])
out, fl = eng.multi_pairing(g1, g2, 1)
print("mode B pairing", dec_fp12s(out) == [O.pairing(P, Q) for P, Q in ap], fl, flush=True)
out, fl = eng.multi_pairing(g1, g2, 3)
print("multi k=3", dec_fp12s(out) == [O.multi_pairing(ap[0:3]), O.multi_pairing(ap[3:6])], fl, flush=True)
# timing
for n in (1 << 12, 1 << 14, 1 << 16):
    reps = (n + 5) // 6
    G1 = np.tile(g1, reps)[: n * 12]; G2 = np.tile(g2, reps)[: n * 24]
    for name, fn in (("miller", eng.multi_miller_loop), ("pairing", eng.multi_pairing)):
        t = time.time(); out, fl = fn(G1, G2, 1); dt = time.time() - t
        print("%s n=%d wall %.3fs kernel %.3f ms -> %.1f k/s" % (name, n, dt, eng.last_kernel_ms(), n / eng.last_kernel_ms()), flush=True)
    for sc in (1, 0):
        eng.set_option(B.OPT_REF_OPT_FP2_SHORTCUT, sc)
        pj = np.tile(enc_fps([c for P, Q in pairs for c in flat_g1_jac(P)]), (n + 7) // 8)[: n * 18]
        qj = np.tile(enc_fps([c for P, Q in pairs for c in flat_g2_jac(Q)]), (n + 7) // 8)[: n * 36]
        out, fl = eng.optimized_miller_loop(pj, qj)
        print("ref_opt shortcut=%d n=%d kernel %.3f ms -> %.1f k/s" % (sc, n, eng.last_kernel_ms(), n / eng.last_kernel_ms()), flush=True)
