"""Sharded sketching across the GPUs of one node: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests), each rank sketches its own
record-aligned shard, then ONE collective merges the partial results:

  1. every rank exports all (hash, count) it saw with hash <= its own admission threshold T_r
     (no multiplicity filter, so counts stay summable)
  2. ONE all-gather of a fixed-size slab per rank: [n, T_r, hashes.., counts..]
  3. every rank takes T_min = min_r T_r, drops what lies above it, and merges: sum counts per hash,
     keep count >= m, first s ascending.

Exact for any m: each rank's threshold never drops below the global s-th qualifying hash
(local counts are lower bounds of global counts), so below T_min every rank's list is complete
and its counts are exact.  The payload is a few thousand 12-byte entries per rank: latency-bound,
not link-bound, which is why it is a single collective.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import numpy as np
import torch
import torch.distributed as dist

U64_MAX = (1 << 64) - 1


def _to_i64(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.astype(np.uint64).view(np.int64).copy())


def _gather(t: torch.Tensor) -> List[torch.Tensor]:
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return out


def exchange_and_merge(local_threshold: int, export: Callable[[int], Tuple[np.ndarray, np.ndarray]], s: int, min_mult: int,
                       merge: Callable[[np.ndarray, np.ndarray, int, int], Tuple[np.ndarray, np.ndarray]],
                       device: torch.device) -> Tuple[np.ndarray, np.ndarray]:
    """The exchange step. `export(limit)` -> (hashes, counts) of this rank; `merge` = engine.merge_partials.
    Works on any initialised process group; tensors live on `device` (cuda for nccl, cpu for gloo).
    One collective: an all-gather of one fixed-size slab per rank ([n, threshold, hashes.., counts..],
    4*s + 4096 entries; a second, exact-size round only if a rank holds more)."""
    hashes, counts = export(local_threshold)
    cap = 4 * s + 4096
    n = len(hashes)

    def slab(capacity: int) -> torch.Tensor:
        buf = np.zeros(2 + 2 * capacity, dtype=np.int64)
        buf[0] = n
        buf[1] = np.uint64(local_threshold).astype(np.uint64).view(np.int64)   # u64 travels as int64 bits
        m = min(n, capacity)
        buf[2:2 + m] = hashes[:m].view(np.int64)
        buf[2 + capacity:2 + capacity + m] = counts[:m]
        return torch.from_numpy(buf).to(device)

    got = [x.cpu().numpy() for x in _gather(slab(cap))]
    sizes = [int(g[0]) for g in got]
    if max(sizes) > cap:   # rare: a rank saw more distinct hashes below its threshold than the fixed slab holds
        cap = max(sizes)
        got = [x.cpu().numpy() for x in _gather(slab(cap))]
    t_min = min(int(g[1:2].view(np.uint64)[0]) for g in got)
    all_h, all_c = [], []
    for r, g in enumerate(got):
        h = g[2:2 + sizes[r]].view(np.uint64)
        c = g[2 + cap:2 + cap + sizes[r]].astype(np.uint32)
        keep = h <= np.uint64(t_min)
        all_h.append(h[keep])
        all_c.append(c[keep])
    return merge(np.concatenate(all_h), np.concatenate(all_c), s, min_mult)


def shard_bounds(n_records: int, world: int, rank: int) -> Tuple[int, int]:
    """Record range [lo, hi) of `rank` (records never split, k-mers never span shards)."""
    base, extra = divmod(n_records, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
