"""Developer tool: throughput of the widened surfaces as a HOST caller sees them -- the host-pointer
entry points stage through PCIe, so these figures include both copies.  Inputs are random canonical residues (timing is
data independent apart from the square-root sign fix-up)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from blsutil import synthetic_batch
import bls381_amd as B

eng = B.Engine(0)
rs = np.random.default_rng(7)


def residues(count):
    """count x 12 u32 digits, each value < 2^380 < p"""
    a = rs.integers(0, 1 << 32, size=(count, 12), dtype=np.uint64).astype(np.uint32)
    a[:, 11] &= 0x0fffffff
    return a.reshape(-1)


def best(fn, reps=3):
    t = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t = min(t, time.perf_counter() - t0)
    return t


def report(name, n, t, bytes_io):
    print("%-44s n=%-8d %8.2f ms  %9.1f k/s  (%.0f MB over PCIe)" % (name, n, t * 1e3, n / t / 1e3, bytes_io / 1e6), flush=True)


n = 1 << 16
x = residues(n * 12); report("witness fq12 inverse (fq12_target.rs:351)", n, best(lambda: eng.witness_fq12_inverse(x)), 2 * n * 576)
x = residues(n * 6); report("witness fq6 inverse (fq6_target.rs:395)", n, best(lambda: eng.witness_fq6_inverse(x)), 2 * n * 288)
x = residues(n * 2); report("witness fq2 inverse (fq2_target.rs:331)", n, best(lambda: eng.witness_fq2_inverse(x)), 2 * n * 96)
sg = (rs.integers(0, 2, size=n)).astype(np.uint8)
x = residues(n); report("witness fq sqrt (fq_target.rs:330)", n, best(lambda: eng.witness_fq_sqrt(x, sg)), 2 * n * 48)
x = residues(n * 2); report("witness fq2 sqrt (fq2_target.rs:385)", n, best(lambda: eng.witness_fq2_sqrt(x, sg)), 2 * n * 96)
g1, g2 = synthetic_batch(1 << 14, jacobian=False, pool=64)
report("g2 prepare, 68 triples/point (miller_loop_target.rs:23)", 1 << 14, best(lambda: eng.g2_prepare(g2)), (1 << 14) * (192 + 68 * 288))
j1, j2 = synthetic_batch(n, jacobian=True, pool=64)
report("G2 Jacobian add (opt.rs:93,98)", n, best(lambda: eng.point_op(2, B.POINT_ADD, j2, j2[::-1].copy())), 3 * n * 288)
report("G2 normalise", n, best(lambda: eng.point_op(2, B.POINT_NORMALIZE, j2)), 2 * n * 288)
report("G1 Jacobian add", n, best(lambda: eng.point_op(1, B.POINT_ADD, j1, j1[::-1].copy())), 3 * n * 144)
a = np.frombuffer(rs.bytes(n * 576), dtype=np.uint64).copy(); a[5::6] &= np.uint64(0x0fffffffffffffff)
report("fq12 mul", n, best(lambda: eng.fp12_op(B.FP12_MUL, a, a)), 3 * n * 576)
report("fq12 inverse", n, best(lambda: eng.fp12_op(B.FP12_INV, a)), 2 * n * 576)
