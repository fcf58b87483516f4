"""N > 1 path on CPU: world_size-2 gloo processes shard a batch by the product's own range logic, compute
their range (here with the C oracle standing in for the GPU, which this container lacks) and gather; the
result must equal the single-process result.  Exercises bls381_amd.sharding exactly as bench.py uses it."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from blsutil import ROOT, synthetic_batch, COracle

from bls381_amd.sharding import shard_range, shard_inputs, gather_outputs, scatter_batch, gather_results, sharded_pairing


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 1 << 20, (1 << 20) + 3):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port, n, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g1, g2 = synthetic_batch(n, seed=31337, jacobian=False, k=k, pool=8)
        a, b, lo, hi = shard_inputs(g1, g2, 12 * k, 24 * k, rank, world)
        out, fl = COracle().multi(np.ascontiguousarray(a), np.ascontiguousarray(b), k, final_exp=True)
        t = torch.from_numpy(out.view(np.int64))
        full = gather_outputs(t, n, 72, dst=0)
        dist.barrier()
        if rank == 0:
            q.put(full.numpy().view(np.uint64).copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,k", [(13, 1), (6, 3)])
def test_two_rank_gloo_matches_single_process(n, k):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, k, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g1, g2 = synthetic_batch(n, seed=31337, jacobian=False, k=k, pool=8)
    exp, _ = COracle().multi(g1, g2, k, final_exp=True)
    assert np.array_equal(got, exp)


def _exchange_worker(rank, world, port, n, k, root, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g1 = g2 = None
        if rank == root:                                   # the batch exists on the root only
            a, b = synthetic_batch(n, seed=4711, jacobian=False, k=k, pool=8)
            g1 = torch.from_numpy(a.view(np.int64)); g2 = torch.from_numpy(b.view(np.int64))

        def run_local(l1, l2, n_local):                    # the C oracle stands in for the GPU launch (no GPU here)
            assert l1.numel() == n_local * 12 * k and l2.numel() == n_local * 24 * k
            if n_local == 0:
                return torch.empty(0, dtype=torch.int64), torch.empty(0, dtype=torch.uint8)
            out, fl = COracle().multi(np.ascontiguousarray(l1.numpy().view(np.uint64)), np.ascontiguousarray(l2.numpy().view(np.uint64)),
                                      k, final_exp=True)
            return torch.from_numpy(out.view(np.int64)), torch.from_numpy(fl)

        out, fl = sharded_pairing(run_local, g1, g2, k, n, root=root)
        dist.barrier()
        if rank == root:
            q.put((out.numpy().view(np.uint64).copy(), fl.numpy().copy()))
        else:
            assert out is None and fl is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,k,root,world", [(13, 1, 0, 2), (5, 3, 1, 2), (2, 1, 0, 3)])
def test_scatter_compute_gather_from_one_rank(n, k, root, world):
    """scatter_batch -> local launch -> gather_results with the batch living on one rank: ragged ranges (13 over 2), a
    non-zero root, and a world larger than the batch (an empty range on one rank)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, n, k, root, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, gfl = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = synthetic_batch(n, seed=4711, jacobian=False, k=k, pool=8)
    exp, efl = COracle().multi(a, b, k, final_exp=True)
    assert np.array_equal(got, exp) and np.array_equal(gfl, efl)
