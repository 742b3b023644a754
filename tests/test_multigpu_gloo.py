"""The cross-rank exchange (auriclass_amd.multigpu) on CPU: world_size 2 (and 3) with the gloo
backend.  Each rank's partial result is produced by the oracle's definition-level hashing of
its own record shard; the exchange code and the merge (libmhx's host-side
mhx_merge_partials) are the real ones."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from auriclass_amd import engine, multigpu, synth

U64_MAX = (1 << 64) - 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _local_partial(records, k, s, m):
    """(all distinct hashes with counts, local admission threshold) of one shard."""
    from oracle import mash_oracle as mo
    import ctypes

    parts = []
    for r in records:
        if len(r) >= k:
            o = np.zeros(len(r), dtype=np.uint64)
            b = ctypes.create_string_buffer(r, len(r))
            n = mo.lib().mo_all_window_hashes(b, len(r), k, o.ctypes.data)
            parts.append(o[:n])
    allh = np.concatenate(parts) if parts else np.zeros(0, np.uint64)
    vals, cnts = np.unique(allh, return_counts=True)
    solid = vals[cnts >= m]
    thr = int(solid[s - 1]) if len(solid) >= s else U64_MAX
    return vals, cnts.astype(np.uint32), thr


def _worker(rank, world, port, k, s, m, n_reads, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    genome = synth.make_genome(30_000, seed=9)
    fq = synth.make_fastq(genome, n_reads, 100, seed=10, device="cpu").numpy().tobytes()
    recs = [fq[i:i + synth.record_bytes(100)].split(b"\n")[1] for i in range(0, len(fq), synth.record_bytes(100))]
    lo, hi = multigpu.shard_bounds(len(recs), world, rank)
    vals, cnts, thr = _local_partial(recs[lo:hi], k, s, m)

    def export(limit):
        keep = vals <= np.uint64(limit)
        return vals[keep], cnts[keep]

    got_h, got_c = multigpu.exchange_and_merge(thr, export, s, m, engine.merge_partials, torch.device("cpu"))
    np.save(os.path.join(out_dir, f"h{rank}.npy"), got_h)
    np.save(os.path.join(out_dir, f"c{rank}.npy"), got_c)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,k,s,m", [(2, 21, 500, 1), (2, 21, 500, 3), (3, 16, 200, 2)])
def test_exchange_and_merge_equals_single_sketch(tmp_path, world, k, s, m):
    from oracle import mash_oracle as mo

    engine.build()
    n_reads = 3000
    port = _free_port()
    mp.spawn(_worker, args=(world, port, k, s, m, n_reads, str(tmp_path)), nprocs=world, join=True)
    genome = synth.make_genome(30_000, seed=9)
    fq = synth.make_fastq(genome, n_reads, 100, seed=10, device="cpu").numpy().tobytes()
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(fq)
    want, _ = ref.finish()
    assert len(want) == s
    for r in range(world):
        got = np.load(tmp_path / f"h{r}.npy")
        assert np.array_equal(got, want), f"rank {r}"
        cnt = np.load(tmp_path / f"c{r}.npy")
        assert cnt.min() >= m


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 100):
        for w in (1, 2, 3, 8):
            spans = [multigpu.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def test_fastq_record_cuts_land_on_record_starts():
    """Byte-range sharding of a real-looking FASTQ: ragged reads, quality lines that begin with '@' or '+',
    CRLF; every cut must be the start of a record and the shards must tile the file."""
    rng = np.random.default_rng(5)
    quals = np.frombuffer(b"@+!#IJ5<?ACGT", np.uint8)
    for nl in (b"\n", b"\r\n"):
        recs, starts, off = [], set(), 0
        for i in range(3000):
            L = int(rng.integers(1, 300))
            seq = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L))
            q = bytes(rng.choice(quals, size=L))
            if i % 7 == 0:
                q = b"@" + q[1:]
            if i % 11 == 0:
                q = b"+" + q[1:]
            rec = b"@read%d some text" % i + nl + seq + nl + b"+" + nl + q + nl
            starts.add(off)
            off += len(rec)
            recs.append(rec)
        data = b"".join(recs)
        for world in (1, 2, 3, 8, 64):
            cuts = multigpu.fastq_record_cuts(data, world)
            assert cuts[0] == 0 and cuts[-1] == len(data) and len(cuts) == world + 1
            assert all(a <= b for a, b in zip(cuts, cuts[1:]))
            assert all(c in starts or c == len(data) for c in cuts)
            sizes = [b - a for a, b in zip(cuts, cuts[1:])]
            assert max(sizes) <= len(data) // world + 2000
