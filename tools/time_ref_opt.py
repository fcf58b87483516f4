"""Developer tool: Mode A (optimized_miller_loop) throughput, generic Fq12 accumulators vs Fp2 shortcut."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
from blsutil import synthetic_batch
import bls381_amd as B
eng = B.Engine(0)
dev = torch.device("cuda", 0)
for lg in (16, 20):
    n = 1 << lg
    g1, g2 = synthetic_batch(n, jacobian=True, pool=64)
    d1 = torch.from_numpy(g1.view(np.int64)).to(dev); d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    out = torch.empty(n * 72, dtype=torch.int64, device=dev); fl = torch.empty(n, dtype=torch.uint8, device=dev)
    for sc in (0, 1):
        eng.set_option(B.OPT_REF_OPT_FP2_SHORTCUT, sc)
        best = 1e9
        for _ in range(3):
            eng.optimized_miller_loop_dev(d1.data_ptr(), d2.data_ptr(), out.data_ptr(), fl.data_ptr(), n); eng.sync()
            best = min(best, eng.last_kernel_ms())
        print("n=2^%d shortcut=%d: %.2f ms -> %.2f M/s" % (lg, sc, best, n / best / 1e3), flush=True)
