#!/usr/bin/env python3
"""A/B of two builds of libbls381_hip.so on ONE box (box-to-box variation is ~3 %, larger than most single changes):
    python tools/ab_libs.py tools/ab/lib_old.so plonky2-bls12-381-pairing_amd/lib/libbls381_hip.so [log2 n]
alternates the two libraries over the same resident batch and prints, per library, the device time of a full pairing
call, of the Miller loops alone and of the final exponentiation alone (bls381_last_kernel_ms), and whether outputs agree."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plonky2-bls12-381-pairing_amd", "python"))
import torch
import bls381_amd as B
from bls381_amd import synthetic

paths = sys.argv[1:3]
lg = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n = 1 << lg
dev = torch.device("cuda", 0)
engines = [B.Engine(0, lib_path=os.path.abspath(p)) for p in paths]
pools = synthetic.build_pools(engines[0], 4096)
g1, g2 = synthetic.affine_batch(pools, n, 1)
d1 = torch.from_numpy(g1.view(np.int64)).to(dev); d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
outs = [torch.zeros(n * 72, dtype=torch.int64, device=dev) for _ in engines]
mil = torch.zeros(n * 72, dtype=torch.int64, device=dev)
res = {i: {"pairing": [], "miller": [], "fexp": []} for i in range(len(engines))}
for rep in range(4):
    for i, e in enumerate(engines):
        e.multi_pairing_dev(d1.data_ptr(), d2.data_ptr(), 1, outs[i].data_ptr(), 0, n); t = e.last_kernel_ms()
        if rep: res[i]["pairing"].append(t)
        e.multi_miller_loop_dev(d1.data_ptr(), d2.data_ptr(), 1, mil.data_ptr(), 0, n); t = e.last_kernel_ms()
        if rep: res[i]["miller"].append(t)
        e.final_exponentiation_dev(mil.data_ptr(), mil.data_ptr(), 0, n); t = e.last_kernel_ms()
        if rep: res[i]["fexp"].append(t)
for i, p in enumerate(paths):
    print("%-60s " % p + "  ".join("%s %.2f ms" % (k, min(v)) for k, v in res[i].items()))
print("outputs equal:", bool(torch.equal(outs[0], outs[1])))
