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


def _input(n_reads):
    genome = synth.make_genome(200_000, seed=21)
    return synth.make_fastq(genome, n_reads, READ_LEN, seed=22, device="cpu").numpy()


def _worker(rank, world, port, k, s, m, n_reads, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from auriclass_amd import engine

    engine.init(0)
    fq = _input(n_reads)
    rb = synth.record_bytes(READ_LEN)
    lo, hi = multigpu.shard_bounds(n_reads, world, rank)
    shard = torch.from_numpy(fq[lo * rb:hi * rb]).to("cuda:0")
    torch.cuda.synchronize()
    sk = engine.Sketcher(k, s, m, expected_bytes=shard.numel())
    sk.push_device(shard.data_ptr(), shard.numel(), engine.FMT_FASTQ4)
    sk.sync()
    assert sk.record_count() == hi - lo
    got_h, got_c = multigpu.exchange_and_merge(sk.threshold(), sk.export, s, m, engine.merge_partials, torch.device("cpu"))
    np.save(os.path.join(out_dir, f"h{rank}.npy"), got_h)
    np.save(os.path.join(out_dir, f"c{rank}.npy"), got_c)
    sk.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,k,s,m", [(2, 21, 1000, 1), (2, 21, 1000, 3), (3, 27, 5000, 2)])
def test_sharded_gpu_sketch_plus_exchange_equals_the_oracle(tmp_path, world, k, s, m):
    from oracle import mash_oracle as mo

    n_reads = 60_000
    port = _free_port()
    mp.spawn(_worker, args=(world, port, k, s, m, n_reads, str(tmp_path)), nprocs=world, join=True)
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(_input(n_reads).tobytes())
    want, want_counts = ref.finish()
    assert len(want) == s
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"h{r}.npy"), want), f"rank {r}"
        assert np.load(tmp_path / f"c{r}.npy").min() >= m


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
    got_h, got_c = multigpu.exchange_and_merge_device(sk, s, m, engine.merge_partials, torch.device("cuda", 0))
    ref_h, ref_c = sk.finish()
    np.save(os.path.join(out_dir, "slab_h.npy"), got_h)
    np.save(os.path.join(out_dir, "slab_c.npy"), got_c)
    np.save(os.path.join(out_dir, "fin_h.npy"), ref_h)
    np.save(os.path.join(out_dir, "fin_c.npy"), ref_c)
    # a tiny shard never tightens its threshold: every rank takes the host-side exchange instead
    sk2 = engine.Sketcher(k, s, m, expected_bytes=0)
    rb = synth.record_bytes(READ_LEN)
    sk2.push_device(fq.data_ptr(), 5 * rb, engine.FMT_FASTQ4)
    tiny_h, _ = multigpu.exchange_and_merge_device(sk2, s, m, engine.merge_partials, torch.device("cuda", 0))
    want_h, _ = sk2.finish()
    assert np.array_equal(tiny_h, want_h)
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
    got_h, _ = multigpu.exchange_and_merge(sk.threshold(), sk.export, s, m, engine.merge_partials, torch.device("cpu"))
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
