"""Sharded sketching end to end on the GPU engine: 2 and 3 ranks (one process each, sharing the
box's one MI355X), every rank sketches its own record shard with the HIP kernels, then the real
exchange (auriclass_amd.multigpu, gloo here; RCCL in bench.py --gpus N) and the merge.  The
result on every rank must equal the oracle's sketch of the whole input, for m = 1 and m > 1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from auriclass_amd import multigpu, synth

pytestmark = pytest.mark.gpu

READ_LEN = 150


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _input(n_reads, sub_rate=0.005):
    genome = synth.make_genome(200_000, seed=21)
    return synth.make_fastq(genome, n_reads, READ_LEN, seed=22, sub_rate=sub_rate, device="cpu").numpy()


def _seqs(fq_bytes):
    """sequence lines of a 4-line FASTQ (for the definition-level oracle)"""
    return fq_bytes.split(b"\n")[1::4]


def _exchange(sk, form, k, s, m):
    """`device`: sizes first, data-sized slabs, merge on the GPU (gloo only carries the bytes); `host`: the
    callback form with the host merge (what the CPU tests of the decision logic run)."""
    if form == "device-cuda":   # the buffers the collective moves live in HBM, as under RCCL (gloo stages them through the host)
        return multigpu.exchange_and_merge_device(sk, torch.device("cuda", 0))
    if form.startswith("device"):
        return multigpu.exchange_and_merge_device(sk, torch.device("cpu"))
    return multigpu.exchange_and_merge(sk.threshold(), sk.export, k, s, m, torch.device("cpu"))


def _worker(rank, world, port, k, s, m, n_reads, out_dir, form="device", sub_rate=0.005):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if form == "device-table":   # the fallback of the binned merge: the other ranks' entries go into this rank's candidate table
        os.environ["MHX_MERGE_TABLE"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from auriclass_amd import engine

    engine.init(0)
    fq = _input(n_reads, sub_rate)
    rb = synth.record_bytes(READ_LEN)
    lo, hi = multigpu.shard_bounds(n_reads, world, rank)
    shard = torch.from_numpy(fq[lo * rb:hi * rb]).to("cuda:0")
    torch.cuda.synchronize()
    sk = engine.Sketcher(k, s, m, expected_bytes=shard.numel())
    sk.push_device(shard.data_ptr(), shard.numel(), engine.FMT_FASTQ4)
    sk.sync()
    assert sk.record_count() == hi - lo
    got_h, got_c = _exchange(sk, form, k, s, m)
    np.save(os.path.join(out_dir, f"h{rank}.npy"), got_h)
    np.save(os.path.join(out_dir, f"c{rank}.npy"), got_c)
    if form.startswith("device"):
        np.save(os.path.join(out_dir, f"x{rank}.npy"), np.array([multigpu.last_exchange.get("collectives", 0)]))
        if multigpu.last_exchange.get("entries_per_rank") is not None:   # (the one-collective form does not bring them to the host)
            np.save(os.path.join(out_dir, f"n{rank}.npy"), np.array(multigpu.last_exchange["entries_per_rank"]))
        # the table now holds the union: pushing without a reset must be refused, after a reset the sketcher is as new
        with pytest.raises(engine.EngineError):
            sk.push_device(shard.data_ptr(), shard.numel(), engine.FMT_FASTQ4)
        sk.reset()
        sk.push_device(shard.data_ptr(), shard.numel(), engine.FMT_FASTQ4)
        again_h, again_c = _exchange(sk, form, k, s, m)
        assert np.array_equal(again_h, got_h) and np.array_equal(again_c, got_c)
    sk.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("form", ["device", "device-cuda", "device-table", "host"])
@pytest.mark.parametrize("world,k,s,m", [(2, 21, 1000, 1), (2, 21, 1000, 3), (3, 27, 5000, 2), (3, 16, 3000, 2)])
def test_sharded_gpu_sketch_plus_exchange_equals_the_oracle(tmp_path, world, k, s, m, form):
    from oracle import mash_oracle as mo

    n_reads = 60_000
    port = _free_port()
    mp.spawn(_worker, args=(world, port, k, s, m, n_reads, str(tmp_path), form), nprocs=world, join=True)
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(_input(n_reads).tobytes())
    want, _ = ref.finish()
    assert len(want) == s
    # multiplicities: the engine's are exact; the definition-level oracle counts every window (mash's heap under-counts
    # repeats of its current maximum)
    bf_h, bf_c = mo.bruteforce_sketch(_seqs(_input(n_reads).tobytes()), k, s, m)
    assert np.array_equal(bf_h, want)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"h{r}.npy"), want), f"rank {r}"
        assert np.array_equal(np.load(tmp_path / f"c{r}.npy"), bf_c), f"rank {r}: summed multiplicities"


def test_one_collective_exchange_repeats_with_room_for_the_largest_shard(tmp_path):
    """Device-resident slabs: the first guess of the slab capacity (4 s + 4096) is too small for these shards (3 % substitutions,
    m = 3: ~30 error k-mers per solid one), every rank learns so from the gathered headers, and the exchange is repeated once."""
    from oracle import mash_oracle as mo

    world, k, s, m, n_reads, sub_rate = 2, 21, 1000, 3, 60_000, 0.03
    mp.spawn(_worker, args=(world, _free_port(), k, s, m, n_reads, str(tmp_path), "device-cuda", sub_rate), nprocs=world, join=True)
    bf_h, bf_c = mo.bruteforce_sketch(_seqs(_input(n_reads, sub_rate).tobytes()), k, s, m)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"h{r}.npy"), bf_h)
        assert np.array_equal(np.load(tmp_path / f"c{r}.npy"), bf_c)
        assert int(np.load(tmp_path / f"x{r}.npy")[0]) == 2   # collectives of the first exchange: guess, then room for all


def test_exchange_whose_shards_exceed_a_fixed_slab(tmp_path):
    """Error-bearing reads with a multiplicity filter: a shard exports every singleton below its threshold, far more
    than the 4*s + 4096 entries of the fixed slab that rounds 1-2 gathered first (with AuriClass's defaults that
    fixed slab ALWAYS overflowed).  The slabs are sized from the exchanged headers instead."""
    from oracle import mash_oracle as mo

    world, k, s, m, n_reads, sub_rate = 2, 21, 1000, 3, 60_000, 0.03   # 3 % substitutions: ~30 error k-mers per solid one
    mp.spawn(_worker, args=(world, _free_port(), k, s, m, n_reads, str(tmp_path), "device", sub_rate), nprocs=world, join=True)
    bf_h, bf_c = mo.bruteforce_sketch(_seqs(_input(n_reads, sub_rate).tobytes()), k, s, m)
    assert len(bf_h) == s
    for r in range(world):
        assert max(np.load(tmp_path / f"n{r}.npy")) > 4 * s + 4096
        assert np.array_equal(np.load(tmp_path / f"h{r}.npy"), bf_h)
        assert np.array_equal(np.load(tmp_path / f"c{r}.npy"), bf_c)


def _nccl_worker(rank, world, port, k, s, m, n_reads, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from auriclass_amd import engine

    engine.init(0)
    fq = torch.from_numpy(_input(n_reads)).to("cuda:0")
    torch.cuda.synchronize()
    sk = engine.Sketcher(k, s, m, expected_bytes=fq.numel())
    sk.push_device(fq.data_ptr(), fq.numel(), engine.FMT_FASTQ4)
    ref_h, ref_c = sk.finish()
    got_h, got_c = multigpu.exchange_and_merge_device(sk, torch.device("cuda", 0))
    np.save(os.path.join(out_dir, "slab_h.npy"), got_h)
    np.save(os.path.join(out_dir, "slab_c.npy"), got_c)
    np.save(os.path.join(out_dir, "fin_h.npy"), ref_h)
    np.save(os.path.join(out_dir, "fin_c.npy"), ref_c)
    # a tiny shard never tightens its threshold (T stays at 2^64-1, a short sketch is the exact answer)
    sk2 = engine.Sketcher(k, s, m, expected_bytes=0)
    rb = synth.record_bytes(READ_LEN)
    sk2.push_device(fq.data_ptr(), 5 * rb, engine.FMT_FASTQ4)
    want_h, _ = sk2.finish()
    tiny_h, _ = multigpu.exchange_and_merge_device(sk2, torch.device("cuda", 0))
    assert np.array_equal(tiny_h, want_h) and len(tiny_h) < s and (m > 1 or len(tiny_h) > 0)
    sk.close()
    sk2.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("k,s,m", [(21, 1000, 1), (21, 1000, 3)])
def test_device_slab_exchange_over_rccl_equals_finish(tmp_path, k, s, m):
    """The RCCL form of the exchange (partials gathered on the device, one copy to the host) on the
    one GPU of the box: a single-rank process group, so the all-gather runs through RCCL itself."""
    from oracle import mash_oracle as mo

    n_reads = 60_000
    mp.spawn(_nccl_worker, args=(1, _free_port(), k, s, m, n_reads, str(tmp_path)), nprocs=1, join=True)
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(_input(n_reads).tobytes())
    want, want_c = ref.finish()
    assert np.array_equal(np.load(tmp_path / "slab_h.npy"), want)
    assert np.array_equal(np.load(tmp_path / "fin_h.npy"), want)
    assert np.array_equal(np.load(tmp_path / "slab_c.npy"), np.load(tmp_path / "fin_c.npy"))


def _ragged_input():
    rng = np.random.default_rng(33)
    genome = synth.make_genome(150_000, seed=34)
    quals = np.frombuffer(b"@+!#IJ5<?ACGT", np.uint8)
    recs = []
    for i in range(40_000):
        L = int(rng.integers(20, 251))
        p0 = int(rng.integers(0, len(genome) - L))
        recs.append(b"@r%d" % i + b"\n" + genome[p0:p0 + L].tobytes() + b"\n+\n" + bytes(rng.choice(quals, size=L)) + b"\n")
    return b"".join(recs)


def _byte_range_worker(rank, world, port, k, s, m, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from auriclass_amd import engine

    engine.init(0)
    data = np.frombuffer(_ragged_input(), dtype=np.uint8)
    cuts = multigpu.fastq_record_cuts(data, world)
    shard = torch.from_numpy(data[cuts[rank]:cuts[rank + 1]].copy()).to("cuda:0")
    torch.cuda.synchronize()
    sk = engine.Sketcher(k, s, m, expected_bytes=shard.numel())
    sk.push_device(shard.data_ptr(), shard.numel(), engine.FMT_FASTQ4)
    got_h, _ = multigpu.exchange_and_merge_device(sk, torch.device("cpu"))
    np.save(os.path.join(out_dir, f"b{rank}.npy"), got_h)
    sk.close()
    dist.destroy_process_group()


def test_byte_range_shards_of_a_ragged_fastq(tmp_path):
    """SURVEY.md 8(e) partitioning: every rank takes the byte range [r*B/R, (r+1)*B/R) of ONE ragged FASTQ,
    snapped to record starts (multigpu.fastq_record_cuts), sketches it on the GPU and joins the exchange."""
    from oracle import mash_oracle as mo

    k, s, m, world = 21, 2000, 2, 3
    mp.spawn(_byte_range_worker, args=(world, _free_port(), k, s, m, str(tmp_path)), nprocs=world, join=True)
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(_ragged_input())
    want, _ = ref.finish()
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"b{r}.npy"), want), f"rank {r}"


# ---- the exactness rule of the sharded path (m > 1 protocol) -------------------------------
# Inputs on which a shard's admission threshold ends as a host-imposed cap below the global s-th solid hash:
# the plain exchange must refuse them on EVERY rank (never a short sketch), and multigpu.sharded_sketch must
# come back with the oracle's sketch on every rank after sketching the shards again with a wider budget.
HARD_CASES = {
    # name: (genome bp, reads, k, s, m)
    "low_coverage_m3": (60_000_000, 120_000, 21, 20_000, 3),   # coverage 0.3: the input of the single-GPU retry test
    "m8_at_40x": (300_000, 80_000, 21, 2_000, 8),              # m >= 8 is the regime DESIGN 3.2 sends to the retry
    "fewer_than_s_solid": (2_000_000, 7_000, 21, 50_000, 3),   # 0.5x coverage: 16 512 solid k-mers in total, s = 50 000
}


def _hard_input(name):
    glen, n_reads, *_ = HARD_CASES[name]
    genome = synth.make_genome(glen, seed=77)
    return synth.make_fastq(genome, n_reads, READ_LEN, seed=78, device="cpu").numpy()


def _hard_worker(rank, world, port, name, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from auriclass_amd import engine

    engine.init(0)
    _, n_reads, k, s, m = HARD_CASES[name]
    fq = _hard_input(name)
    rb = synth.record_bytes(READ_LEN)
    lo, hi = multigpu.shard_bounds(n_reads, world, rank)
    shard = torch.from_numpy(fq[lo * rb:hi * rb]).to("cuda:0")
    torch.cuda.synchronize()
    attempts = []

    def push(sk):
        attempts.append(1)
        sk.push_device(shard.data_ptr(), shard.numel(), engine.FMT_FASTQ4)

    # 1. the bare exchange on budget-1 sketchers: exact or refused, on every rank alike
    sk = engine.Sketcher(k, s, m, expected_bytes=shard.numel())
    push(sk)
    try:
        h, _ = multigpu.exchange_and_merge_device(sk, torch.device("cpu"))
        np.save(os.path.join(out_dir, f"plain{rank}.npy"), h)
    except multigpu.InexactShardedSketch:
        np.save(os.path.join(out_dir, f"plain{rank}.npy"), np.zeros(1, np.float64))   # marker: refused
    sk.close()
    # 2. the retrying driver
    got_h, got_c = multigpu.sharded_sketch(push, k, s, m, shard.numel(), torch.device("cpu"))
    np.save(os.path.join(out_dir, f"h{rank}.npy"), got_h)
    np.save(os.path.join(out_dir, f"c{rank}.npy"), got_c)
    np.save(os.path.join(out_dir, f"a{rank}.npy"), np.array([len(attempts)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", sorted(HARD_CASES))
def test_sharded_sketch_is_exact_or_refused_never_short(tmp_path, name, world):
    from oracle import mash_oracle as mo

    _, n_reads, k, s, m = HARD_CASES[name]
    mp.spawn(_hard_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(_hard_input(name).tobytes())
    want, want_c = ref.finish()
    if name == "fewer_than_s_solid":
        assert 0 < len(want) < s
    else:
        assert len(want) == s
    bf_h, bf_c = mo.bruteforce_sketch(_seqs(_hard_input(name).tobytes()), k, s, m)   # exact multiplicities
    assert np.array_equal(bf_h, want)
    refused = []
    for r in range(world):
        plain = np.load(tmp_path / f"plain{r}.npy")
        refused.append(plain.dtype == np.float64)
        if not refused[-1]:
            assert np.array_equal(plain, want), f"rank {r}: the bare exchange returned a sketch that is not the oracle's"
        assert np.array_equal(np.load(tmp_path / f"h{r}.npy"), want), f"rank {r}"
        # counts: the engine's are exact multiplicities (the oracle's heap reproduces mash's order-dependent
        # under-count), so they are compared with the definition-level count of every window
        assert np.array_equal(np.load(tmp_path / f"c{r}.npy"), bf_c), f"rank {r}: summed multiplicities"
    assert len(set(refused)) == 1, "ranks disagree on the verdict"
    attempts = {int(np.load(tmp_path / f"a{r}.npy")[0]) for r in range(world)}
    assert len(attempts) == 1, "ranks retried a different number of times"
    if refused[0]:
        assert attempts.pop() >= 3   # 1 (bare) + >= 2 inside sharded_sketch


def _nccl_hard_worker(rank, world, port, name, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from auriclass_amd import engine

    engine.init(0)
    _, n_reads, k, s, m = HARD_CASES[name]
    fq = torch.from_numpy(_hard_input(name)).to("cuda:0")
    torch.cuda.synchronize()
    attempts = []

    def push(sk):
        attempts.append(1)
        sk.push_device(fq.data_ptr(), fq.numel(), engine.FMT_FASTQ4)

    got_h, got_c = multigpu.sharded_sketch(push, k, s, m, fq.numel(), torch.device("cuda", 0))
    np.save(os.path.join(out_dir, "h.npy"), got_h)
    np.save(os.path.join(out_dir, "a.npy"), np.array([len(attempts)]))
    dist.destroy_process_group()


def test_device_slab_exchange_retries_a_capped_shard(tmp_path):
    """The slab form (RCCL, 1-rank group on the box's one GPU) carries the same rule: the low-coverage input is
    refused at budget 1 and comes back exact from the retry."""
    from oracle import mash_oracle as mo

    name = "low_coverage_m3"
    _, n_reads, k, s, m = HARD_CASES[name]
    mp.spawn(_nccl_hard_worker, args=(1, _free_port(), name, str(tmp_path)), nprocs=1, join=True)
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(_hard_input(name).tobytes())
    want, _ = ref.finish()
    assert np.array_equal(np.load(tmp_path / "h.npy"), want)
    assert int(np.load(tmp_path / "a.npy")[0]) >= 2


# ---- one sample over several GPUs at file level (what classes.py:576-596 hands to `mash sketch -r`) ----------------
def _files_worker(rank, world, port, paths, k, s, m, out_msh, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from auriclass_amd import engine

    engine.init(0)
    text, size = multigpu.sketch_fastq_files(paths, k, s, m, out_msh, torch.device("cpu"))
    with open(os.path.join(out_dir, f"stderr{rank}.txt"), "w") as fh:
        fh.write(text)
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["plain", "gz_pair"])
def test_sharded_file_level_sketch_writes_the_single_gpu_msh(tmp_path, kind):
    """multigpu.sketch_fastq_files over 2 ranks (byte-range shards of plain files, whole files of a .gz pair) writes the
    bytes engine.sketch_files(..., reads=True) writes on one GPU, and reports the same genome size."""
    import gzip

    from auriclass_amd import engine

    k, s, m, world = 27, 5000, 3, 2
    data = _ragged_input()
    half = data.index(b"\n@r20000\n") + 1
    if kind == "plain":
        paths = [str(tmp_path / "a.fq"), str(tmp_path / "b.fq")]
        (tmp_path / "a.fq").write_bytes(data[:half])
        (tmp_path / "b.fq").write_bytes(data[half:])
    else:
        paths = [str(tmp_path / "a.fq.gz"), str(tmp_path / "b.fq.gz")]
        (tmp_path / "a.fq.gz").write_bytes(gzip.compress(data[:half], 1))
        (tmp_path / "b.fq.gz").write_bytes(gzip.compress(data[half:], 1))
    sharded = tmp_path / "sharded.msh"
    mp.spawn(_files_worker, args=(world, _free_port(), paths, k, s, m, str(sharded), str(tmp_path)), nprocs=world, join=True)
    engine.init(0)
    single = tmp_path / "single.msh"
    text, size = engine.sketch_files(paths, k, s, single, reads=True, min_mult=m)
    assert sharded.read_bytes() == single.read_bytes()
    want_lines = [ln for ln in text.splitlines() if ln.startswith("Estimated")]
    for r in range(world):
        got = (tmp_path / f"stderr{r}.txt").read_text()
        assert [ln for ln in got.splitlines() if ln.startswith("Estimated")] == want_lines


def _dist_worker(rank, world, port, nq, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.test_multigpu_gloo import _dist_inputs

    nr, s, k = 24, 4000, 27
    Q, q_len, R, r_len = _dist_inputs(nq, nr, s, seed=8)
    dev = torch.device("cpu")   # gloo carries the small result gather; the comparison runs on the GPU
    c, d, x, _ = multigpu.sharded_dist_batch(Q, q_len, R, r_len, k, s, dev)
    np.savez(os.path.join(out_dir, f"d{rank}.npz"), c=c, d=d, x=x)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nq", [(2, 300), (3, 130)])
def test_sharded_distances_on_the_gpu_equal_the_oracle(tmp_path, world, nq):
    """The query rows of a distance batch over 2 and 3 ranks (sharing the one GPU): each rank's rows through the HIP
    kernels (the one-query-per-lane form at 150 rows, the wave form at 43), the gathered table equal to the oracle's."""
    from tests.test_multigpu_gloo import _dist_inputs, _oracle_dist

    mp.spawn(_dist_worker, args=(world, _free_port(), nq, str(tmp_path)), nprocs=world, join=True)
    Q, q_len, R, r_len = _dist_inputs(nq, 24, 4000, seed=8)
    wc, wd, wx = _oracle_dist(Q, q_len, R, r_len, 27, 4000)
    for r in range(world):
        z = np.load(tmp_path / f"d{r}.npz")
        assert np.array_equal(z["c"], wc) and np.array_equal(z["d"], wd), f"rank {r}"
        assert np.allclose(z["x"], wx, rtol=1e-12, atol=0), f"rank {r}"   # device log vs libm: a few ulp


def _sparse_worker(rank, world, port, k, s, m, n_reads, form, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from auriclass_amd import engine

    engine.init(0)
    fq = _input(n_reads)
    rb = synth.record_bytes(READ_LEN)
    lo, hi = multigpu.shard_bounds(n_reads, world, rank)
    sk = engine.Sketcher(k, s, m, expected_bytes=max(1, (hi - lo) * rb))
    if hi > lo:
        shard = torch.from_numpy(fq[lo * rb:hi * rb]).to("cuda:0")
        torch.cuda.synchronize()
        sk.push_device(shard.data_ptr(), shard.numel(), engine.FMT_FASTQ4)
        sk.sync()
    got_h, got_c = _exchange(sk, form, k, s, m)
    np.save(os.path.join(out_dir, f"h{rank}.npy"), got_h)
    np.save(os.path.join(out_dir, f"c{rank}.npy"), got_c)
    sk.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("form", ["device", "device-cuda", "host"])
@pytest.mark.parametrize("m", [1, 2])
def test_fewer_records_than_ranks(tmp_path, form, m):
    """Two reads over three ranks: one rank's sketcher never sees a byte (empty partial, threshold still the largest hash
    value); the merged sketch -- shorter than s, which is exact when no rank has ever rejected a hash -- is the oracle's
    on every rank."""
    from oracle import mash_oracle as mo

    world, k, s, n_reads = 3, 21, 500, 2
    mp.spawn(_sparse_worker, args=(world, _free_port(), k, s, m, n_reads, form, str(tmp_path)), nprocs=world, join=True)
    want, wc = mo.bruteforce_sketch(_seqs(_input(n_reads).tobytes()), k, s, m)
    assert len(want) < s
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"h{r}.npy"), want), f"rank {r}"
        assert np.array_equal(np.load(tmp_path / f"c{r}.npy"), wc), f"rank {r}"


def _files_error_worker(rank, world, port, paths, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from auriclass_amd import engine

    engine.init(0)
    try:
        multigpu.sketch_fastq_files(paths, 21, 1000, 1, os.path.join(out_dir, "x.msh"), torch.device("cpu"))
        verdict = "sketched"
    except engine.EngineError as e:
        verdict = "refused: " + e.message
    with open(os.path.join(out_dir, f"v{rank}.txt"), "w") as fh:
        fh.write(verdict)
    dist.destroy_process_group()


def test_sharded_file_level_sketch_refuses_a_last_record_without_qualities(tmp_path):
    """The sharded form pushes byte ranges of a file itself: it asks the library about the file's tail
    (mhx_fastq_tail_complete) and every rank refuses a FASTQ cut short inside its last record, like the one-GPU call."""
    data = _ragged_input()
    cut = data.rindex(b"\n+\n") + 3           # the last record keeps its '+' line and loses its qualities
    p = tmp_path / "cut.fq"
    p.write_bytes(data[:cut])
    mp.spawn(_files_error_worker, args=(2, _free_port(), [str(p)], str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert (tmp_path / f"v{r}.txt").read_text().startswith("refused: truncated quality string"), r
