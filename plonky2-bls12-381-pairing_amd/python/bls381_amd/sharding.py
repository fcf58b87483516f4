"""Multi-GPU layer: pairings are independent (no cross-element state in optimized_miller_loop,
src/miller_loop_native_optimized.rs:81-127), so a batch shards into contiguous ranges, one per rank, with
NO collective on the data path.  One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" in the CPU tests).

For callers whose batch lives on ONE rank ("one scatter of inputs from root and one gather of outputs"):
`scatter_batch` / `gather_results` move the contiguous ranges with grouped point-to-point sends (RCCL send/recv, one
message per peer and array).  xGMI is a full mesh of direct links, so the root talks to its seven peers over seven
different links at once; ranges may differ in size by one element, which RCCL's fixed-size scatter/gather cannot
express.  `sharded_pairing` strings scatter -> local launch -> gather together.  None of this is inside bench.py's
timed region (the inputs there are resident per rank); `bench.py --exchange` reports its cost separately.
"""
from __future__ import annotations

import numpy as np


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [lo, hi) of `n` elements owned by `rank` (sizes differ by at most one)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return n * rank // world, n * (rank + 1) // world


def shard_inputs(g1: np.ndarray, g2: np.ndarray, w1: int, w2: int, rank: int, world: int):
    """Slice ABI-layout input arrays (w1 / w2 u64 words per element) down to this rank's range."""
    n = g1.size // w1
    assert g1.size == n * w1 and g2.size == n * w2, "ragged input"
    lo, hi = shard_range(n, rank, world)
    return g1[lo * w1:hi * w1], g2[lo * w2:hi * w2], lo, hi


def gather_outputs(local_out, n_total: int, words: int = 72, dst: int = 0):
    """Collect per-rank result shards (torch tensors of int64/uint8, on the backend's device) on rank `dst`.
    Returns the concatenated tensor on dst, None elsewhere.  Uses all_gather on padded shards so it works on
    every backend (gloo has no gather for CUDA tensors; RCCL has no variable-size gather)."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    per = [shard_range(n_total, r, world) for r in range(world)]
    max_n = max(hi - lo for lo, hi in per)
    pad = torch.zeros(max_n * words, dtype=local_out.dtype, device=local_out.device)
    pad[: local_out.numel()] = local_out.reshape(-1)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: (hi - lo) * words] for r, (lo, hi) in enumerate(per)])


def _p2p(ops):
    import torch.distributed as dist
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def scatter_batch(arrays, words, n_total: int, src: int = 0, device=None, dtype=None):
    """Hand every rank its contiguous range of a batch that lives on rank `src`.

    arrays: on `src` a list of 1-D torch tensors in ABI layout (array j has words[j] entries per element, n_total
    elements); ignored (may be None) elsewhere.  Returns this rank's slices as a list of tensors (views on `src`,
    freshly received buffers elsewhere).  `device` / `dtype` describe the receive buffers on non-root ranks
    (defaults: the backend's natural device -- current CUDA device for nccl, CPU for gloo -- and int64)."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    per = [shard_range(n_total, r, world) for r in range(world)]
    lo, hi = per[rank]
    if rank == src:
        for a, w in zip(arrays, words):
            if a.numel() != n_total * w:
                raise ValueError("scatter_batch: array does not hold n_total elements")
        ops = [dist.P2POp(dist.isend, a[l * w:h * w], r)
               for r, (l, h) in enumerate(per) if r != src and h > l for a, w in zip(arrays, words)]
        _p2p(ops)
        return [a[lo * w:hi * w] for a, w in zip(arrays, words)]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    bufs = [torch.empty((hi - lo) * w, dtype=dtype or torch.int64, device=device) for w in words]
    if hi > lo:
        _p2p([dist.P2POp(dist.irecv, b, src) for b in bufs])
    return bufs


def gather_results(local, words, n_total: int, dst: int = 0):
    """Inverse of scatter_batch: collect every rank's result ranges (list of 1-D tensors, words[j] entries per element)
    on rank `dst` in batch order.  Returns the list of full tensors on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    per = [shard_range(n_total, r, world) for r in range(world)]
    lo, hi = per[rank]
    for t, w in zip(local, words):
        if t.numel() != (hi - lo) * w:
            raise ValueError("gather_results: local tensor does not match this rank's range")
    if rank != dst:
        if hi > lo:
            _p2p([dist.P2POp(dist.isend, t, dst) for t in local])
        return None
    full = [torch.empty(n_total * w, dtype=t.dtype, device=t.device) for t, w in zip(local, words)]
    for f, t, w in zip(full, local, words):
        f[lo * w:hi * w] = t
    ops = [dist.P2POp(dist.irecv, f[l * w:h * w], r)
           for r, (l, h) in enumerate(per) if r != dst and h > l for f, w in zip(full, words)]
    _p2p(ops)
    return full


def sharded_pairing(run_local, g1, g2, k: int, n_total: int, root: int = 0, device=None):
    """A batch of n_total k-pair products that lives on rank `root`: scatter the (G1, G2) ranges, run the local launch,
    gather GT values and flags back.  `run_local(g1_local, g2_local, n_local) -> (out int64[n_local*72], flags uint8[n_local])`
    is the per-rank hot path (bls381_multi_pairing_batch_dev behind Engine.multi_pairing_dev on a GPU rank).
    Returns (out, flags) on root, (None, None) elsewhere."""
    import torch.distributed as dist

    a, b = scatter_batch([g1, g2] if dist.get_rank() == root else None, [12 * k, 24 * k], n_total, src=root, device=device)
    lo, hi = shard_range(n_total, dist.get_rank(), dist.get_world_size())
    out, flags = run_local(a, b, hi - lo)
    res = gather_results([out, flags], [72, 1], n_total, dst=root)
    return (res[0], res[1]) if res is not None else (None, None)
