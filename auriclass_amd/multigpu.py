"""Sharded sketching across the GPUs of one node: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests), each rank sketches its own
record-aligned shard, then ONE exchange step merges the partial results:

  1. all-reduce(MIN) of the ranks' admission thresholds  -> common limit T_min
  2. every rank exports all (hash, count) it saw with hash <= T_min (no multiplicity filter,
     so counts stay summable), all-gather of the sizes, all-gather of the padded slabs
  3. every rank merges: sum counts per hash, keep count >= m, first s ascending.

Exact for any m: each rank's threshold never drops below the global s-th qualifying hash
(local counts are lower bounds of global counts), so below T_min every rank has complete
counts.  The payload is a few thousand 12-byte entries per rank: latency-bound, not link-bound.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np
import torch
import torch.distributed as dist

U64_MAX = (1 << 64) - 1


def _to_i64(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.astype(np.uint64).view(np.int64).copy())


def exchange_and_merge(local_threshold: int, export: Callable[[int], Tuple[np.ndarray, np.ndarray]], s: int, min_mult: int,
                       merge: Callable[[np.ndarray, np.ndarray, int, int], Tuple[np.ndarray, np.ndarray]],
                       device: torch.device) -> Tuple[np.ndarray, np.ndarray]:
    """The exchange step. `export(limit)` -> (hashes, counts) of this rank; `merge` = engine.merge_partials.
    Works on any initialised process group; tensors live on `device` (cuda for nccl, cpu for gloo)."""
    world = dist.get_world_size()
    # 1. common limit: min over ranks of a u64, done as two non-negative int64 halves
    hi_lo = torch.tensor([local_threshold >> 32], dtype=torch.int64, device=device)
    dist.all_reduce(hi_lo, op=dist.ReduceOp.MIN)
    hi = int(hi_lo.item())
    lo_t = torch.tensor([(local_threshold & 0xFFFFFFFF) if (local_threshold >> 32) == hi else 0xFFFFFFFF],
                        dtype=torch.int64, device=device)
    dist.all_reduce(lo_t, op=dist.ReduceOp.MIN)
    t_min = (hi << 32) | int(lo_t.item())
    # 2. slabs
    hashes, counts = export(t_min)
    n_local = torch.tensor([len(hashes)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(x.item()) for x in sizes]
    pad = max(max(sizes), 1)
    slab = torch.zeros(2 * pad, dtype=torch.int64, device=device)
    if len(hashes):
        slab[:len(hashes)] = _to_i64(hashes).to(device)
        slab[pad:pad + len(counts)] = torch.from_numpy(counts.astype(np.int64)).to(device)
    slabs = [torch.empty_like(slab) for _ in range(world)]
    dist.all_gather(slabs, slab)
    # 3. merge
    all_h, all_c = [], []
    for r in range(world):
        v = slabs[r].cpu().numpy()
        all_h.append(v[:sizes[r]].view(np.uint64))
        all_c.append(v[pad:pad + sizes[r]].astype(np.uint32))
    return merge(np.concatenate(all_h), np.concatenate(all_c), s, min_mult)


def shard_bounds(n_records: int, world: int, rank: int) -> Tuple[int, int]:
    """Record range [lo, hi) of `rank` (records never split, k-mers never span shards)."""
    base, extra = divmod(n_records, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
