#!/usr/bin/env python3
"""Device time of pairing products of k pairs (bls381_multi_pairing_batch_dev / bls381_multi_miller_batch_dev) over k:
    python tools/time_multi_k.py [log2 n] [k ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plonky2-bls12-381-pairing_amd", "python"))
import numpy as np, torch
import bls381_amd as B
from bls381_amd import synthetic
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ks = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 5, 6, 8]
n = 1 << lg
dev = torch.device("cuda", 0); eng = B.Engine(0)
pools = synthetic.build_pools(eng, 4096)
for k in ks:
    g1, g2 = synthetic.affine_batch(pools, n, k)
    d1 = torch.from_numpy(g1.view(np.int64)).to(dev); d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    out = torch.zeros(n * 72, dtype=torch.int64, device=dev)
    tm, tp = [], []
    for r in range(4):
        eng.multi_miller_loop_dev(d1.data_ptr(), d2.data_ptr(), k, out.data_ptr(), 0, n); t = eng.last_kernel_ms()
        if r: tm.append(t)
        eng.multi_pairing_dev(d1.data_ptr(), d2.data_ptr(), k, out.data_ptr(), 0, n); t = eng.last_kernel_ms()
        if r: tp.append(t)
    print("n=2^%d k=%d  Miller loops %.2f ms (%.2f per pair)   pairing product %.2f ms" % (lg, k, min(tm), min(tm) / k, min(tp)))
