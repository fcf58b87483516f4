"""Multi-GPU layer: pairings are independent (no cross-element state in optimized_miller_loop,
src/miller_loop_native_optimized.rs:81-127), so a batch shards into contiguous ranges, one per rank, with
NO collective on the data path.  One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" in the CPU tests).  A gather of the results to one rank is offered for callers that want the
whole batch in one place; it is not part of the timed path.
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
