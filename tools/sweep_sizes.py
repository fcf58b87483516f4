#!/usr/bin/env python3
"""Device time of a full-pairing call (bls381_multi_pairing_batch_dev, k = 1) over batch sizes: where does the engine reach its rate?
    python tools/sweep_sizes.py [log2 sizes ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plonky2-bls12-381-pairing_amd", "python"))
import numpy as np, torch
import bls381_amd as B
from bls381_amd import synthetic
sizes = [int(a) for a in sys.argv[1:]] or list(range(8, 21))
dev = torch.device("cuda", 0); eng = B.Engine(0)
pools = synthetic.build_pools(eng, 4096)
for lg in sizes:
    n = 1 << lg
    g1, g2 = synthetic.affine_batch(pools, n, 1)
    d1 = torch.from_numpy(g1.view(np.int64)).to(dev); d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    out = torch.zeros(n * 72, dtype=torch.int64, device=dev)
    best = 1e9
    for r in range(5):
        eng.multi_pairing_dev(d1.data_ptr(), d2.data_ptr(), 1, out.data_ptr(), 0, n); t = eng.last_kernel_ms()
        if r: best = min(best, t)
    print("n=2^%-2d  %8.3f ms  %7.3f M pairings/s  (Miller %.3f ms, final exponentiation %.3f ms: mode %d, %d pipeline(s))" % (
        lg, best, n / best / 1e3, eng.last_phase_ms(0), eng.last_phase_ms(1), eng.get_option(B.INFO_LAST_FEXP_MODE), eng.get_option(B.INFO_LAST_FEXP_PIPELINES)))
