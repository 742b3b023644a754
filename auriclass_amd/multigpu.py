"""Sharded sketching across the GPUs of one node: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests), each rank sketches its own
record-aligned shard, then ONE collective merges the partial results:

  1. every rank exports all (hash, count) it saw with hash <= its own admission threshold T_r
     (no multiplicity filter, so counts stay summable)
  2. sizes first (an all-gather of 64-byte headers: n_r, T_r, flags), then ONE all-gather of slabs sized
     from the data: [hashes[max n_r] | counts]
  3. every rank takes T_min = min_r T_r, drops what lies above it, and merges ON THE DEVICE: the other
     ranks' entries are added to its own candidate table (sum of counts per hash), the ordinary extraction
     keeps count >= m, first s ascending (exchange_and_merge_device / mhx_sketcher_merge_slabs; the host
     form exchange_and_merge / mhx_merge_shard_partials serves the CPU tests of the decision logic).
  4. the exactness rule, on gathered data only, so that every rank reaches the same verdict:
     >= s merged entries qualify below T_min, or no rank has ever rejected a hash (T_min is still
     the largest hash value) -> done; otherwise :class:`InexactShardedSketch` -- never a short sketch.
     `sharded_sketch` answers that by sketching every shard again with a 16x admission budget.

Why 4 is needed: below T_min every rank's list is complete and its counts exact (a threshold only
ever falls).  A threshold lowered by the tighten pass sits above s locally qualifying hashes, which
qualify globally too, so the union has its s entries below T_min.  With a multiplicity filter
(m > 1; 3 is AuriClass's FASTQ default, /root/reference/auriclass/args.py:128-134) a shard's threshold
is at first a host-imposed CAP that follows the bytes seen (mhx_engine.cpp, push_device); a shallow
shard can end with that cap below the global s-th solid hash -- plenty of solid k-mers exist, but above
T_min, where the other ranks' lists are incomplete.  The payload is ~1.1 s entries of 12 bytes per rank for
m = 1 and ~10-14 s entries for m = 3 on error-bearing reads (every singleton below T_r travels, counts
must stay summable): 13 KB ... 8 MB per rank, still latency- rather than link-bound on xGMI.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import engine

U64_MAX = (1 << 64) - 1
DEVICE_FLAG_MASK = 0xFF   # word [2] of a slab: device flags below, MHX_SLAB_* host-state bits above
SLAB_BOUNDED = 0x100
SLAB_ESTABLISHED = 0x200


class InexactShardedSketch(engine.EngineError):
    """The gathered partials cannot decide the sketch of the union (raised on every rank alike)."""


def _to_i64(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.astype(np.uint64).view(np.int64).copy())


def _gather(t: torch.Tensor) -> List[np.ndarray]:
    """All-gather of equally sized 1-D tensors; returns one host array per rank (ONE device-to-host copy)."""
    world = dist.get_world_size()
    flat = torch.empty(world * t.numel(), dtype=t.dtype, device=t.device)
    try:
        dist.all_gather_into_tensor(flat, t)
    except (RuntimeError, NotImplementedError, AttributeError):   # backend without the flat form
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        flat = torch.cat(parts)
    host = flat.cpu().numpy()
    return [host[r * t.numel():(r + 1) * t.numel()] for r in range(world)]


def _merge(parts_h: List[np.ndarray], parts_c: List[np.ndarray], thresholds: List[int], k: int, s: int, min_mult: int
           ) -> Tuple[np.ndarray, np.ndarray]:
    try:
        return engine.merge_shard_partials(parts_h, parts_c, thresholds, k, s, min_mult)
    except engine.EngineError as e:
        if e.code == engine.MHX_E_CAPACITY:
            raise InexactShardedSketch(e.code, e.message) from None
        raise


def exchange_and_merge(local_threshold: int, export: Callable[[int], Tuple[np.ndarray, np.ndarray]], k: int, s: int,
                       min_mult: int, device: torch.device) -> Tuple[np.ndarray, np.ndarray]:
    """The exchange step. `export(limit)` -> (hashes, counts) of this rank.
    Works on any initialised process group; tensors live on `device` (cuda for nccl, cpu for gloo).
    One collective: an all-gather of one fixed-size slab per rank ([n, threshold, hashes.., counts..],
    4*s + 4096 entries; a second, exact-size round only if a rank holds more).
    Raises InexactShardedSketch on every rank when the partials do not determine the union's sketch."""
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

    got = _gather(slab(cap))
    sizes = [int(g[0]) for g in got]
    if max(sizes) > cap:   # rare: a rank saw more distinct hashes below its threshold than the fixed slab holds
        cap = max(sizes)
        got = _gather(slab(cap))
    thresholds = [int(g[1:2].view(np.uint64)[0]) for g in got]
    all_h = [g[2:2 + sizes[r]].view(np.uint64) for r, g in enumerate(got)]
    all_c = [g[2 + cap:2 + cap + sizes[r]].astype(np.uint32) for r, g in enumerate(got)]
    return _merge(all_h, all_c, thresholds, k, s, min_mult)


# timings of the last exchange_and_merge_device call on this rank (ms) and its payload: bench.py reports them
last_exchange: dict = {}
_buffers: dict = {}


def _buffer(tag: str, numel: int, device: torch.device) -> torch.Tensor:
    """Send / receive buffers are kept between calls (a fresh 66 MB host tensor costs more than the collective)."""
    key = (tag, str(device))
    t = _buffers.get(key)
    if t is None or t.numel() < numel:
        t = torch.empty(numel + numel // 4, dtype=torch.int64, device=device)
        _buffers[key] = t
    return t[:numel]


def _all_gather_flat(out: torch.Tensor, t: torch.Tensor) -> None:
    try:
        dist.all_gather_into_tensor(out, t)
    except (RuntimeError, NotImplementedError, AttributeError):   # backend without the flat form
        world = dist.get_world_size()
        parts = [out[r * t.numel():(r + 1) * t.numel()] for r in range(world)]
        dist.all_gather(parts, t)


def exchange_and_merge_device(sk, device: torch.device) -> Tuple[np.ndarray, np.ndarray]:
    """The exchange as SURVEY.md 8(e) lays it out, with the merge where the data is:
      1. `sk.export_begin()` compacts the shard's partial result on the GPU; the ranks all-gather the 64-byte headers
         (n_r, T_r, flags, count of the hash value 2^64-1, table occupancy) -- sizes first;
      2. ONE all-gather of slabs sized from the data (max_r n_r entries, rounded up to 1024): RCCL moves them from HBM
         to HBM over xGMI (`device` cuda), gloo through host memory (`device` cpu, the rehearsal / test form);
      3. `sk.merge_slabs` adds the other ranks' entries <= T_min to this rank's candidate table with device atomics and
         runs the ordinary extraction: nothing is sorted or merged on the host, whatever m and s are.
    Same exactness rule as finish(), decided from gathered data only, so every rank raises InexactShardedSketch
    together.  The sketcher's table holds the union afterwards: reset() it before the next push."""
    import os
    import time

    world, rank = dist.get_world_size(), dist.get_rank()
    on_device = device.type == "cuda"
    sk.sync()   # the shard's own sketch kernels (the export would wait for them anyway): not part of the exchange's time
    if on_device and os.environ.get("MHX_EXCHANGE_SIZES_FIRST") != "1":
        return _exchange_one_collective(sk, device, world, rank)
    t0 = time.perf_counter()
    hdr = sk.export_begin()
    t1 = time.perf_counter()
    mine = torch.from_numpy(hdr.view(np.int64)).to(device)
    all_hdr_t = torch.empty(world * 8, dtype=torch.int64, device=device)
    _all_gather_flat(all_hdr_t, mine)
    all_hdr = all_hdr_t.cpu().numpy().view(np.uint64).reshape(world, 8)
    t2 = time.perf_counter()
    _raise_on_device_flags(all_hdr)
    max_n = int(all_hdr[:, 0].max())
    cap = max(1024, (max_n + 1023) // 1024 * 1024)
    words = cap + cap // 2
    send = _buffer("send", words, device)
    sk.export_pack(send.data_ptr(), cap)
    t3 = time.perf_counter()
    recv = _buffer("recv", world * words, device)
    _all_gather_flat(recv, send)
    if on_device:
        torch.cuda.current_stream(device).synchronize()   # the merge runs on the engine's own stream
    t4 = time.perf_counter()
    try:
        result = sk.merge_slabs(recv.data_ptr(), on_device, world, cap, all_hdr, rank)
    except engine.EngineError as e:
        if e.code == engine.MHX_E_CAPACITY:
            raise InexactShardedSketch(e.code, e.message) from None
        raise
    t5 = time.perf_counter()
    last_exchange.update(export_ms=(t1 - t0) * 1e3, sizes_ms=(t2 - t1) * 1e3, pack_ms=(t3 - t2) * 1e3, gather_ms=(t4 - t3) * 1e3,
                         merge_ms=(t5 - t4) * 1e3, total_ms=(t5 - t0) * 1e3, entries_per_rank=[int(x) for x in all_hdr[:, 0]],
                         slab_bytes=words * 8, collectives=2)
    return result


def _raise_on_device_flags(all_hdr: np.ndarray) -> None:
    flags = 0
    for r in range(all_hdr.shape[0]):
        flags |= int(all_hdr[r, 2]) & DEVICE_FLAG_MASK & ~0x8
    if flags:
        raise RuntimeError(f"device flags {flags:#x} raised during sketching (table full / malformed FASTQ)")


_cap_guess: dict = {}   # (s, m, world) -> slab capacity that held the last exchange's largest shard, with room


def _exchange_one_collective(sk, device: torch.device, world: int, rank: int) -> Tuple[np.ndarray, np.ndarray]:
    """Device-resident slabs (RCCL): the sizes ride in front of the slabs.  `sk.export_into` compacts the shard's partial
    result STRAIGHT into the send buffer [header8 | hashes[cap] | counts[cap]] (no separate compaction buffer, no pack
    step), ONE all-gather moves the slabs, `sk.merge_gathered` reads the gathered headers back and merges on the device.
    `cap` is a guess: what held the last exchange of this (s, m, world), 4 s + 4096 the first time; it is derived from
    gathered data only, so that every rank arrives at the same number.  If some rank holds more than that, every rank learns so from the same gathered headers and the exchange is
    repeated once with room for the largest shard -- the explicit sizes-first round only where it is needed.  Saves a
    collective, two host round trips and two copies per exchange (0.07 ms of 0.18 on one rank)."""
    import time

    key = (sk.s, sk.m, world)
    cap = _cap_guess.get(key, (4 * sk.s + 4096 + 1023) // 1024 * 1024)
    t0 = time.perf_counter()
    t_export = t_gather = 0.0
    for attempt in range(3):
        words = 8 + cap + cap // 2
        send = _buffer("send1", words, device)
        ta = time.perf_counter()
        hdr = sk.export_into(send.data_ptr(), cap)
        tb = time.perf_counter()
        recv = _buffer("recv1", world * words, device)
        _all_gather_flat(recv, send)
        torch.cuda.current_stream(device).synchronize()   # the merge runs on the engine's own stream
        tc = time.perf_counter()
        t_export += tb - ta
        t_gather += tc - tb
        try:
            hashes, counts, need = sk.merge_gathered(recv.data_ptr(), world, cap, rank)
        except engine.EngineError as e:
            if e.code == engine.MHX_E_CAPACITY:
                raise InexactShardedSketch(e.code, e.message) from None
            raise
        if not need:
            _cap_guess[key] = cap   # (never from this rank's own size: every rank must arrive at the same capacity)
            t1 = time.perf_counter()
            last_exchange.update(export_ms=t_export * 1e3, sizes_ms=0.0, pack_ms=0.0, gather_ms=t_gather * 1e3,
                                 merge_ms=(t1 - tc) * 1e3, total_ms=(t1 - t0) * 1e3, entries_per_rank=None, slab_bytes=words * 8,
                                 collectives=attempt + 1)
            return hashes, counts
        cap = max(1024, (need * 5 // 4 + 1023) // 1024 * 1024)   # the same number on every rank: same headers
        _cap_guess[key] = cap
    raise RuntimeError("sharded exchange: slab capacity kept growing")


def sharded_sketch(push: Callable[[object], None], k: int, s: int, min_mult: int, expected_bytes: int, device: torch.device,
                   first: Optional[object] = None, max_budget_scale: int = 4096,
                   sketcher_factory: Optional[Callable[[int], object]] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Sketch of the union of all ranks' shards, exact for any m, on every rank.
    `push(sk)` feeds this rank's shard into the engine.Sketcher it is given (it may be called again:
    every retry sketches the shard afresh).  `first`: a sketcher that already holds the shard (budget 1).
    `device`: where the collective's tensors live (cuda -> RCCL with device-resident slabs, cpu -> gloo).
    When the exchange reports InexactShardedSketch -- on all ranks at once -- every rank repeats with 16x
    the admission budget and candidate table (what mhx_sketch_files does by itself on one GPU); past
    `max_budget_scale` the exception is passed on.  `sketcher_factory(budget_scale)` replaces engine.Sketcher
    (the CPU tests of the decision logic use it)."""
    scale, sk, own = 1, first, False
    while True:
        if sk is None:
            sk = (sketcher_factory(scale) if sketcher_factory else
                  engine.Sketcher(k, s, min_mult, expected_bytes=expected_bytes, budget_scale=scale))
            own = True
            push(sk)
        try:
            if hasattr(sk, "export_begin"):   # the engine's sketcher: partials merged on the GPU (RCCL or gloo transport)
                return exchange_and_merge_device(sk, device)
            return exchange_and_merge(sk.threshold(), sk.export, k, s, min_mult, device)
        except InexactShardedSketch:
            if scale * 16 > max_budget_scale:
                raise
            scale *= 16
        finally:
            if own:
                sk.close()
            sk, own = None, False


def shard_bounds(n_records: int, world: int, rank: int) -> Tuple[int, int]:
    """Record range [lo, hi) of `rank` (records never split, k-mers never span shards)."""
    base, extra = divmod(n_records, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sharded_dist_batch(q: np.ndarray, q_len: np.ndarray, r: np.ndarray, r_len: np.ndarray, k: int, s: int,
                       device: torch.device, gather: bool = True,
                       compute: Optional[Callable[..., Tuple[np.ndarray, np.ndarray, np.ndarray]]] = None
                       ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, Tuple[int, int]]:
    """Batched distances over the GPUs of a node (`mash dist` is independent per (query, reference) pair,
    /root/reference/auriclass/classes.py:92-104): rank i compares query rows shard_bounds(nq, world, i) with ALL
    references (replicated: 24 x 400 KB at AuriClass's defaults) on its own GPU -- no collective on the data path.
    Returns (common, denom, dist, (lo, hi)).  gather=False: the three arrays hold this rank's rows [lo, hi) only.
    gather=True: ONE all-gather of the results (16 bytes per pair, rows padded to the largest shard) and every rank
    returns all [nq, nr] rows.  `compute` (default engine.dist_batch, the HIP path) is a hook for the CPU tests of this
    sharding logic, which have no GPU to compute on."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    nq, nr = int(q.shape[0]), int(r.shape[0])
    lo, hi = shard_bounds(nq, world, rank)
    fn = compute or engine.dist_batch
    if hi > lo:
        common, denom, dd = fn(q[lo:hi], q_len[lo:hi], r, r_len, k, s)
    else:   # more ranks than queries
        common, denom, dd = (np.zeros((0, nr), np.uint32), np.zeros((0, nr), np.uint32), np.zeros((0, nr), np.float64))
    if not gather or world == 1:
        return common, denom, dd, (lo, hi)
    per = -(-nq // world)   # rows of the largest shard
    # one int64 payload per pair and plane: plane 0 = (common << 32) | denom, plane 1 = the bits of the f64 distance
    send = torch.zeros((per, nr, 2), dtype=torch.int64)
    send[: hi - lo, :, 0] = torch.from_numpy((common.astype(np.int64) << 32) | denom.astype(np.int64))
    send[: hi - lo, :, 1] = torch.from_numpy(np.ascontiguousarray(dd, dtype=np.float64).view(np.int64))
    send = send.to(device)
    out = torch.empty((world, per, nr, 2), dtype=torch.int64, device=device)
    if device.type == "cuda":
        dist.all_gather_into_tensor(out.view(-1), send.view(-1))
    else:
        dist.all_gather(list(out.unbind(0)), send)
    out = out.cpu().numpy()
    all_common = np.zeros((nq, nr), np.uint32)
    all_denom = np.zeros((nq, nr), np.uint32)
    all_dist = np.zeros((nq, nr), np.float64)
    for i in range(world):
        a, b = shard_bounds(nq, world, i)
        pk = out[i, : b - a, :, 0]
        all_common[a:b] = (pk >> 32).astype(np.uint32)
        all_denom[a:b] = (pk & 0xFFFFFFFF).astype(np.uint32)
        all_dist[a:b] = np.ascontiguousarray(out[i, : b - a, :, 1]).view(np.float64)
    return all_common, all_denom, all_dist, (lo, hi)


def fastq_record_cuts(data, world: int) -> List[int]:
    """Byte offsets [c_0 = 0, c_1, ..., c_world = len] that cut a 4-line FASTQ held in `data` (bytes, memoryview
    or a uint8 numpy array) into `world` record-aligned shards of roughly equal size: cut r is the first record
    start at or after r * len / world.  A record start is a line that begins with '@', whose
    successor line does not begin with '@' or '+' ... and whose line after that begins with '+' -- the shape no
    quality line can imitate over two lines (a quality line may begin with '@', but then the line after it is a
    header, which begins with '@' as well, not '+')."""
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    n = len(buf)
    cuts = [0]
    for r in range(1, world):
        pos = r * n // world
        cut = n
        # scan line starts from pos on; a handful of lines suffices
        i = pos
        if i > 0:
            nl = np.flatnonzero(buf[i - 1:min(n, i + (1 << 20))] == 10)
            i = n if len(nl) == 0 else i - 1 + int(nl[0]) + 1
        while i < n:
            window = buf[i:min(n, i + (1 << 20))]
            nls = np.flatnonzero(window == 10)
            if len(nls) < 2:
                break
            l1, l2 = i + int(nls[0]) + 1, i + int(nls[1]) + 1       # starts of the next two lines
            if buf[i] == 64 and l2 < n and buf[l2] == 43 and (l1 >= n or buf[l1] != 64):
                cut = i
                break
            i = l1
        cuts.append(max(cut, cuts[-1]))
    cuts.append(n)
    return cuts


# ------------------------------------------------------------------------------------------------------------------
# One sample, several GPUs, at file level: what /root/reference/auriclass/classes.py:576-596 hands to `mash sketch -r`
# (the FASTQ files of ONE sample) spread over the ranks of the process group.
def _first_counted_header(path, k: int) -> Tuple[str, str]:
    """(name, comment) of the first record of `path` whose sequence holds >= k bytes (the record that names mash's
    reference); the very first record if the first MiB has none.  Reads the head of the file only."""
    import zlib

    with open(path, "rb") as fh:
        head = fh.read(1 << 20)
    if head[:2] == b"\x1f\x8b":
        head = zlib.decompressobj(31).decompress(head, 1 << 20)
    lines = head.split(b"\n")
    pick = None
    for i in range(0, len(lines) - 1, 4):
        if not lines[i].startswith(b"@"):
            break
        if pick is None:
            pick = lines[i]
        if len(lines[i + 1].rstrip(b"\r")) >= k:
            pick = lines[i]
            break
    if pick is None:
        return "", ""
    text = pick[1:].rstrip(b"\r").decode("utf-8", "surrogateescape")
    name, _, comment = text.partition(" ")
    if "\t" in name:   # kseq splits the name at the first blank of either kind
        name, _, rest = text.partition("\t")
        comment = rest
    return name, comment


def sketch_fastq_files(paths, k: int, s: int, min_mult: int, out_msh, device: torch.device) -> Tuple[str, float]:
    """`mash sketch -r -m M -o OUT -k K -s S paths...` for ONE sample over all ranks of the initialised process group
    (one process per GPU): every uncompressed file is cut into `world` record-aligned byte ranges and rank r takes
    range r of each (fastq_record_cuts only looks at the bytes around the cut points); a gzip file can only be read
    from its start, so file i goes whole to rank i % world (the usual pair of .fq.gz keeps two GPUs busy, and inflating is
    what bounds such inputs anyway).  Each rank sketches what it took on its GPU, ONE exchange merges the partial results
    (sharded_sketch), rank 0 writes the .msh -- the same bytes `engine.sketch_files(..., reads=True)` writes on one GPU.
    Returns (stderr text, estimated genome size) on every rank.  Strict 4-line FASTQ only (what the device parser
    takes); anything else: use the single-GPU call."""
    world, rank = dist.get_world_size(), dist.get_rank()
    pieces = []
    for i, p in enumerate(paths):
        with open(p, "rb") as fh:
            magic = fh.read(2)
        if magic == b"\x1f\x8b":
            if i % world == rank:
                with open(p, "rb") as fh:
                    pieces.append(np.frombuffer(engine.gunzip(fh.read(), threads=8), dtype=np.uint8))
                    if not engine.fastq_tail_complete(pieces[-1][-65536:].tobytes()):
                        raise engine.EngineError(engine.MHX_E_FORMAT, f"truncated quality string in the last FASTQ record of {p}")
        else:
            mm = np.memmap(p, dtype=np.uint8, mode="r")
            if not engine.fastq_tail_complete(bytes(mm[max(0, len(mm) - 65536):])):   # every rank looks: all raise together
                raise engine.EngineError(engine.MHX_E_FORMAT, f"truncated quality string in the last FASTQ record of {p}")
            cuts = fastq_record_cuts(mm, world)
            if cuts[rank + 1] > cuts[rank]:
                pieces.append(mm[cuts[rank]:cuts[rank + 1]])
    nbytes = int(sum(len(x) for x in pieces))
    records = []

    def push(sk):
        for piece in pieces:
            sk.push_host(np.ascontiguousarray(piece), engine.FMT_FASTQ4)
        sk.sync()
        records.append(sk.record_count())

    hashes, counts = sharded_sketch(push, k, s, min_mult, nbytes, device)
    total = torch.tensor([records[-1] if records else 0], dtype=torch.int64, device=device)
    dist.all_reduce(total, op=dist.ReduceOp.SUM)
    n_records = int(total.item())
    if n_records == 0:   # no record holds k bases (the device may still have hashed windows of k BYTES: CRLF reads of k - 1 bases)
        raise engine.NoRecordsError(engine.MHX_E_NO_RECORDS, f'ERROR: Did not find fasta records in "{paths[0]}".')
    set_size = mult = 0.0
    if len(hashes):
        set_size = (2.0 ** (64 if k > 16 else 32)) * len(hashes) / float(hashes[-1])
        mult = float(counts.astype(np.uint64).sum()) / len(hashes)
    name, comment = _first_counted_header(paths[0], k)
    text = f"{name} {comment}"
    if n_records > 1:
        text = f"[{n_records} seqs] {text} [...]"
    if rank == 0:
        engine.msh_write(out_msh, k, s, [str(paths[0])], [text], [int(set_size)], [hashes])
    dist.barrier()
    stderr = "Estimated genome size: %g\nEstimated coverage:    %g\nWriting to %s...\n" % (set_size, mult, out_msh)
    return stderr, set_size
