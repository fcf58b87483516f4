#!/usr/bin/env python3
"""Does running the Miller kernel of one half batch concurrently with the final exponentiation of another pay?
Two contexts on one device, each on its own stream: Miller loops of 2^19 pairs + final exponentiation of 2^19 values, one after
the other vs at the same time.   python tools/overlap_probe.py [log2 half-batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plonky2-bls12-381-pairing_amd", "python"))
import numpy as np, torch
import bls381_amd as B
from bls381_amd import synthetic
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 19
n = 1 << lg
dev = torch.device("cuda", 0)
e1, e2 = B.Engine(0), B.Engine(0)
pools = synthetic.build_pools(e1, 4096)
g1, g2 = synthetic.affine_batch(pools, n, 1)
d1 = torch.from_numpy(g1.view(np.int64)).to(dev); d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
mil = torch.zeros(n * 72, dtype=torch.int64, device=dev); mil2 = torch.zeros(n * 72, dtype=torch.int64, device=dev)
out = torch.zeros(n * 72, dtype=torch.int64, device=dev)
e2.set_option(B.OPT_PAIRING_HOMOGENEOUS_MILLER, 1)
e1.multi_miller_loop_dev(d1.data_ptr(), d2.data_ptr(), 1, mil.data_ptr(), 0, n); e1.sync()
for pipes in (1, 2):
    e2.set_option(B.OPT_FEXP_PIPELINES, pipes)
    e2.final_exponentiation_dev(mil.data_ptr(), out.data_ptr(), 0, n); e2.sync()      # warm (scratch allocation)
    res = {}
    for mode in ("serial", "overlap", "serial", "overlap"):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e1.multi_miller_loop_dev(d1.data_ptr(), d2.data_ptr(), 1, mil2.data_ptr(), 0, n)
        if mode == "serial":
            e1.sync()
        e2.final_exponentiation_dev(mil.data_ptr(), out.data_ptr(), 0, n)
        e1.sync(); e2.sync()
        res.setdefault(mode, []).append((time.perf_counter() - t0) * 1e3)
    print("n=2^%d fexp pipelines=%d  Miller then FE: %.2f ms   Miller || FE: %.2f ms" % (lg, pipes, min(res["serial"]), min(res["overlap"])))
