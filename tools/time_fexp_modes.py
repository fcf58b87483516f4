#!/usr/bin/env python3
"""Final exponentiation alone, batched-inversion form vs staged form, over batch sizes (where should BLS381_FEXP_AUTO switch?):
    python tools/time_fexp_modes.py [log2 sizes ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plonky2-bls12-381-pairing_amd", "python"))
import torch
import bls381_amd as B
from bls381_amd import synthetic

sizes = [int(a) for a in sys.argv[1:]] or [15, 16, 17, 18, 19, 20]
dev = torch.device("cuda", 0)
eng = B.Engine(0)
pools = synthetic.build_pools(eng, 4096)
for lg in sizes:
    n = 1 << lg
    g1, g2 = synthetic.affine_batch(pools, n, 1)
    d1 = torch.from_numpy(g1.view(np.int64)).to(dev); d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    mil = torch.zeros(n * 72, dtype=torch.int64, device=dev)
    out = torch.zeros(n * 72, dtype=torch.int64, device=dev)
    eng.multi_miller_loop_dev(d1.data_ptr(), d2.data_ptr(), 1, mil.data_ptr(), 0, n); eng.sync()
    line = "n=2^%d " % lg
    ref = None
    for mode, name in ((B.FEXP_BATCHED_INVERSION, "batched-inversion"), (B.FEXP_STAGED, "staged")):
        eng.set_option(B.OPT_FEXP_MODE, mode)
        best = 1e9
        for r in range(4):
            eng.final_exponentiation_dev(mil.data_ptr(), out.data_ptr(), 0, n); t = eng.last_kernel_ms()
            if r: best = min(best, t)
        if ref is None: ref = out.clone()
        else: assert torch.equal(ref, out)
        line += " %s %.3f ms" % (name, best)
    print(line)
    eng.set_option(B.OPT_FEXP_MODE, B.FEXP_AUTO)
