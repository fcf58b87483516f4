#!/usr/bin/env python3
"""Staged final exponentiation alone at 1, 2 and 3 part-batch pipelines over batch sizes (where should the default split start?):
    python tools/time_fexp_pipelines.py [log2 sizes ...]"""
import os, sys
sys.path.insert(0, "plonky2-bls12-381-pairing_amd/python")
import numpy as np, torch
import bls381_amd as B
from bls381_amd import synthetic
dev = torch.device("cuda", 0); eng = B.Engine(0)
pools = synthetic.build_pools(eng, 4096)
for lg in [int(a) for a in sys.argv[1:]] or (13, 14, 15, 16, 17, 18):
    n = 1 << lg
    g1, g2 = synthetic.affine_batch(pools, n, 1)
    d1 = torch.from_numpy(g1.view(np.int64)).to(dev); d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    mil = torch.zeros(n * 72, dtype=torch.int64, device=dev); out = torch.zeros(n * 72, dtype=torch.int64, device=dev)
    eng.multi_miller_loop_dev(d1.data_ptr(), d2.data_ptr(), 1, mil.data_ptr(), 0, n); eng.sync()
    line = "n=2^%d" % lg
    eng.set_option(B.OPT_FEXP_MODE, B.FEXP_STAGED)
    for pipes in (1, 2, 3):
        eng.set_option(B.OPT_FEXP_PIPELINES, pipes)
        best = 1e9
        for r in range(4):
            eng.final_exponentiation_dev(mil.data_ptr(), out.data_ptr(), 0, n); t = eng.last_kernel_ms()
            if r: best = min(best, t)
        line += "  pipes=%d %.3f ms" % (pipes, best)
    print(line)
