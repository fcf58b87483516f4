"""Developer tool: throughput of the fixed-Q pairing shape (n_var = 0, n_fixed = 1) at 2^20 elements."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
from blsutil import synthetic_batch
import bls381_amd as B
eng = B.Engine(0)
n = 1 << 20
g1, g2 = synthetic_batch(n, jacobian=False, pool=64)
prep, _ = eng.g2_prepare(g2[:24])
dev = torch.device("cuda", 0)
d1 = torch.from_numpy(g1.view(np.int64)).to(dev); dp = torch.from_numpy(prep.view(np.int64)).to(dev)
out = torch.empty(n * 72, dtype=torch.int64, device=dev); fl = torch.empty(n, dtype=torch.uint8, device=dev)
for _ in range(3):
    eng.prepared_pairing_dev(d1.data_ptr(), 0, dp.data_ptr(), 0, 1, out.data_ptr(), fl.data_ptr(), n); eng.sync()
    print("fixed-Q pairings: %.1f ms -> %.2f M pairings/s" % (eng.last_kernel_ms(), n / eng.last_kernel_ms() / 1e3))
