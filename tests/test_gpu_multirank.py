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
