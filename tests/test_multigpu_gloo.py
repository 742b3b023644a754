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

    got_h, got_c = multigpu.exchange_and_merge(thr, export, k, s, m, torch.device("cpu"))
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


@pytest.mark.parametrize("world,n_reads,m", [(3, 2, 1), (3, 2, 2), (2, 1, 1)])
def test_exchange_with_an_empty_shard_and_fewer_kmers_than_the_sketch_holds(tmp_path, world, n_reads, m):
    """Fewer records than ranks: some rank holds nothing (its partial is empty, its threshold still the largest hash
    value); the union has fewer than s k-mers, which is exact as long as no rank has ever rejected a hash."""
    from oracle import mash_oracle as mo

    engine.build()
    k, s = 21, 500
    mp.spawn(_worker, args=(world, _free_port(), k, s, m, n_reads, str(tmp_path)), nprocs=world, join=True)
    genome = synth.make_genome(30_000, seed=9)
    fq = synth.make_fastq(genome, n_reads, 100, seed=10, device="cpu").numpy().tobytes()
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(fq)
    want, _ = ref.finish()
    assert len(want) < s
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"h{r}.npy"), want), f"rank {r}"


class _FakeSketcher:
    """Stands where engine.Sketcher stands in multigpu.sharded_sketch: the shard's exact (hash, count) table with a
    CAPPED admission threshold, the cap widening with the budget scale as the engine's does."""

    def __init__(self, vals, cnts, cap):
        self.vals, self.cnts, self.cap = vals, cnts, cap

    def threshold(self):
        return self.cap

    def export(self, limit):
        keep = self.vals <= np.uint64(min(limit, self.cap))
        return self.vals[keep], self.cnts[keep]

    def close(self):
        pass


def _capped_worker(rank, world, port, k, s, m, n_reads, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    genome = synth.make_genome(30_000, seed=9)
    fq = synth.make_fastq(genome, n_reads, 100, seed=10, device="cpu").numpy().tobytes()
    recs = [fq[i:i + synth.record_bytes(100)].split(b"\n")[1] for i in range(0, len(fq), synth.record_bytes(100))]
    lo, hi = multigpu.shard_bounds(len(recs), world, rank)
    vals, cnts, _ = _local_partial(recs[lo:hi], k, s, m)
    # rank 0's cap lies where far fewer than s globally solid hashes exist below it; budget x16 lifts it, x256 removes it
    base_cap = U64_MAX // 400 if rank == 0 else U64_MAX // 3
    made = []

    def factory(scale):
        made.append(scale)
        return _FakeSketcher(vals, cnts, U64_MAX if scale >= 256 else min(U64_MAX, base_cap * scale))

    # the bare exchange must refuse on every rank
    f = factory(1)
    try:
        multigpu.exchange_and_merge(f.threshold(), f.export, k, s, m, torch.device("cpu"))
        verdict = 0
    except multigpu.InexactShardedSketch:
        verdict = 1
    got_h, got_c = multigpu.sharded_sketch(lambda sk: None, k, s, m, 0, torch.device("cpu"), sketcher_factory=factory)
    np.save(os.path.join(out_dir, f"h{rank}.npy"), got_h)
    np.save(os.path.join(out_dir, f"v{rank}.npy"), np.array([verdict] + made))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_capped_shard_is_refused_on_every_rank_and_the_retry_is_exact(tmp_path, world):
    """Decision logic of the sharded path without a GPU: one rank's threshold is a cap below the union's s-th solid
    hash.  Every rank must raise InexactShardedSketch from the bare exchange (same gathered data, same verdict) and
    multigpu.sharded_sketch must widen the budget on all ranks in lockstep until the result is the oracle's."""
    from oracle import mash_oracle as mo

    engine.build()
    k, s, m, n_reads = 21, 500, 3, 3000
    mp.spawn(_capped_worker, args=(world, _free_port(), k, s, m, n_reads, str(tmp_path)), nprocs=world, join=True)
    genome = synth.make_genome(30_000, seed=9)
    fq = synth.make_fastq(genome, n_reads, 100, seed=10, device="cpu").numpy().tobytes()
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(fq)
    want, _ = ref.finish()
    assert len(want) == s
    scales = set()
    for r in range(world):
        v = np.load(tmp_path / f"v{r}.npy")
        assert v[0] == 1, f"rank {r} accepted partials that do not determine the sketch"
        scales.add(tuple(v[1:].tolist()))
        assert np.array_equal(np.load(tmp_path / f"h{r}.npy"), want), f"rank {r}"
    assert len(scales) == 1 and len(next(iter(scales))) >= 3, scales   # bare + >= 2 attempts, identical on all ranks


def test_merge_shard_partials_rule():
    """mhx_merge_shard_partials: entries above T_min are dropped; short results are accepted only when no shard
    has ever rejected a hash (T_min at the largest value of this hash width)."""
    engine.build()
    a = np.array([5, 10, 20, 30], np.uint64)
    b = np.array([10, 15, 30, 40], np.uint64)
    ones = np.ones(4, np.uint32)
    # complete shards: short sketch is fine
    h, c = engine.merge_shard_partials([a, b], [ones, ones], [U64_MAX, U64_MAX], 21, 10, 2)
    assert h.tolist() == [10, 30] and c.tolist() == [2, 2]
    # T_min = 25: only 10 qualifies below it, s = 1 is satisfied, s = 2 is not
    h, _ = engine.merge_shard_partials([a[:3], b], [ones[:3], ones], [25, U64_MAX], 21, 1, 2)
    assert h.tolist() == [10]
    with pytest.raises(engine.EngineError) as e:
        engine.merge_shard_partials([a[:3], b], [ones[:3], ones], [25, U64_MAX], 21, 2, 2)
    assert e.value.code == engine.MHX_E_CAPACITY
    # 32-bit hashes (k <= 16): the untouched threshold is 2^32 - 1
    h, _ = engine.merge_shard_partials([a, b], [ones, ones], [0xFFFFFFFF, 0xFFFFFFFF], 16, 10, 2)
    assert h.tolist() == [10, 30]
    with pytest.raises(engine.EngineError):
        engine.merge_shard_partials([a, b], [ones, ones], [0xFFFFFFFE, 0xFFFFFFFF], 16, 10, 2)


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


def test_first_counted_header_of_a_file(tmp_path):
    """multigpu.sketch_fastq_files names the reference after the first record mash would count (sequence of >= k bytes),
    read from the head of the first file, plain or gzipped; blanks and tabs split name and comment as in kseq."""
    import gzip

    recs = (b"@short extra words\nACGT\n+\nIIII\n" +
            b"@read7\tlane=3 x\n" + b"ACGTACGTACGTACGTACGTACGTACGT\n+\n" + b"I" * 28 + b"\n" +
            b"@later one\n" + b"C" * 40 + b"\n+\n" + b"I" * 40 + b"\n")
    plain = tmp_path / "a.fq"
    plain.write_bytes(recs)
    gz = tmp_path / "a.fq.gz"
    gz.write_bytes(gzip.compress(recs))
    for p in (plain, gz):
        assert multigpu._first_counted_header(p, 21) == ("read7", "lane=3 x")
        assert multigpu._first_counted_header(p, 4) == ("short", "extra words")
        assert multigpu._first_counted_header(p, 35) == ("later", "one")
        assert multigpu._first_counted_header(p, 64) == ("short", "extra words")   # none is long enough: the very first header


def _dist_inputs(nq, nr, s, seed=5):
    """nr reference lists and nq queries (ascending unique u64 rows, ragged lengths, one empty query)."""
    rng = np.random.default_rng(seed)
    base = np.unique(rng.integers(0, 1 << 63, size=s, dtype=np.uint64))
    R = np.zeros((nr, s), np.uint64)
    r_len = np.zeros(nr, np.uint32)
    for j in range(nr):
        v = np.unique(np.where(rng.random(len(base)) < 0.05 * (j + 1), rng.integers(0, 1 << 63, size=len(base), dtype=np.uint64), base))
        R[j, :len(v)], r_len[j] = v, len(v)
    Q = np.zeros((nq, s), np.uint64)
    q_len = np.zeros(nq, np.uint32)
    for i in range(nq):
        src = R[i % nr, :r_len[i % nr]]
        v = np.unique(np.where(rng.random(len(src)) < i / (2.0 * nq), rng.integers(0, 1 << 63, size=len(src), dtype=np.uint64), src))
        if i == 3:
            v = v[:0]
        elif i % 4 == 1:
            v = v[: len(v) // 2]
        Q[i, :len(v)], q_len[i] = v, len(v)
    return Q, q_len, R, r_len


def _oracle_dist(q, q_len, r, r_len, k, s):
    from oracle import mash_oracle as mo

    nq, nr = q.shape[0], r.shape[0]
    common, denom, dd = np.zeros((nq, nr), np.uint32), np.zeros((nq, nr), np.uint32), np.zeros((nq, nr), np.float64)
    for i in range(nq):
        for j in range(nr):
            common[i, j], denom[i, j], dd[i, j] = mo.compare(r[j, :r_len[j]], q[i, :q_len[i]], s, k)
    return common, denom, dd


def _dist_worker(rank, world, port, nq, nr, s, k, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Q, q_len, R, r_len = _dist_inputs(nq, nr, s)
    c, d, x, (lo, hi) = multigpu.sharded_dist_batch(Q, q_len, R, r_len, k, s, torch.device("cpu"), compute=_oracle_dist)
    own = multigpu.sharded_dist_batch(Q, q_len, R, r_len, k, s, torch.device("cpu"), gather=False, compute=_oracle_dist)
    assert own[0].shape == (hi - lo, nr) and np.array_equal(own[0], c[lo:hi]) and np.array_equal(own[2], x[lo:hi])
    np.savez(os.path.join(out_dir, f"d{rank}.npz"), c=c, d=d, x=x, lo=lo, hi=hi)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nq", [(2, 11), (3, 7), (3, 2)])
def test_sharded_distances_are_the_unsharded_table(tmp_path, world, nq):
    """distances shard over the query rows with no exchange on the data path; the gathered table on every
    rank is the unsharded one (ragged shard sizes, an empty query, more ranks than queries).  The comparison itself is
    the oracle's here (no GPU on this side); tests/test_gpu_multirank.py runs the same call on the HIP path."""
    engine.build()
    nr, s, k = 5, 300, 21
    mp.spawn(_dist_worker, args=(world, _free_port(), nq, nr, s, k, str(tmp_path)), nprocs=world, join=True)
    Q, q_len, R, r_len = _dist_inputs(nq, nr, s)
    wc, wd, wx = _oracle_dist(Q, q_len, R, r_len, k, s)
    covered = np.zeros(nq, bool)
    for r in range(world):
        z = np.load(tmp_path / f"d{r}.npz")
        assert np.array_equal(z["c"], wc) and np.array_equal(z["d"], wd) and np.array_equal(z["x"], wx), f"rank {r}"
        covered[int(z["lo"]):int(z["hi"])] = True
    assert covered.all()
