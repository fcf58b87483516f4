#!/usr/bin/env python3
"""Decodes tools/kbench/kb_dbg_hom2 output (raw 28-bit limbs, Montgomery R = 2^392) and compares every variant's
intermediates with a big-integer evaluation of addition_step_hom.  usage: dbg_hom2.py gpurun_out/dbg_hom2.txt"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle"))
import pyref as O
P = O.P
RINV = pow(1 << 392, -1, P)
rows = {}
for line in open(sys.argv[1]):
    t = line.split()
    if len(t) != 17 or not t[0].isdigit():
        continue
    slot, v, lane = int(t[0]), int(t[1]), int(t[2])
    val = sum(int(x, 16) << (28 * i) for i, x in enumerate(t[3:]))
    rows[(slot, v, lane)] = val * RINV % P
def fp2(slot, v, pair=0):
    return (rows[(slot, v, 2 * pair)], rows[(slot, v, 2 * pair + 1)])
M, S, sub, add, dbl = O.fp2_mul, O.fp2_sqr, O.fp2_sub, O.fp2_add, O.fp2_dbl
names = ["theta", "lambda", "c", "d", "e", "f", "g", "h", "j", "o.x", "o.y", "o.z", "R.x", "R.y", "R.z"]
for pair in (0, 1):
    qx, qy = fp2(7, 0, pair), fp2(7, 1, pair)
    x, y, z = fp2(7, 12, pair), fp2(7, 13, pair), fp2(7, 14, pair)
    theta = sub(y, M(qy, z)); lam = sub(x, M(qx, z)); c = S(theta); d = S(lam); e = M(lam, d); f = M(z, c); g = M(x, d)
    h = sub(add(e, f), dbl(g)); j = sub(M(theta, qx), M(lam, qy))
    o = (M(lam, h), sub(M(theta, sub(g, h)), M(e, y)), M(z, e))
    exp = [theta, lam, c, d, e, f, g, h, j, o[0], o[1], o[2], o[0], o[1], o[2]]
    for slot, nm in [(0, "inline"), (1, "call by reference"), (2, "call by value"), (3, "call, templated body"), (4, "inline + dump"), (5, "call + dump")]:
        got = [fp2(slot, v, pair) for v in range(15)]
        rng = range(15) if slot >= 4 else range(12, 15)
        bad = [names[v] for v in rng if got[v] != exp[v]]
        print("pair %d  %-22s %s" % (pair, nm, "OK" if not bad else "WRONG: " + " ".join(bad)))
