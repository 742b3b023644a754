"""The sharded path's merge (mhx_sketcher_export_begin / _pack / _merge_slabs and the one-collective pair _export_into /
_merge_gathered) with the ranks SIMULATED inside one process: R sketchers take the R record shards of one input, their
slabs are laid side by side in HBM as an all-gather would leave them, and every "rank" merges.  The merged sketch must be
the oracle's sketch of the whole input, multiplicities included -- for 2 ... 9 shards, uneven and tiny ones (thresholds
that never tightened), 32-bit hashes, all the merge paths (binned LDS merge, table path, host merge)."""
import numpy as np
import pytest
import torch

from auriclass_amd import engine, multigpu, synth
from oracle import mash_oracle as mo

pytestmark = pytest.mark.gpu


def _seqs(fq_bytes):
    return fq_bytes.split(b"\n")[1::4]


def _shards(n_reads, R, rng, uneven):
    if not uneven:
        return [multigpu.shard_bounds(n_reads, R, r) for r in range(R)]
    cuts = sorted(int(x) for x in rng.integers(0, n_reads + 1, size=R - 1))
    cuts = [0] + cuts + [n_reads]
    return [(cuts[r], cuts[r + 1]) for r in range(R)]


@pytest.mark.parametrize("seed", range(14))
def test_simulated_ranks_merge_to_the_oracle_sketch(seed):
    rng = np.random.default_rng(9000 + seed)
    k = int(rng.choice([16, 21, 27, 31]))
    s = int(rng.choice([200, 1000, 8000, 50000]))
    m = int(rng.choice([1, 1, 2, 3]))
    R = int(rng.integers(2, 10))
    read_len = int(rng.choice([100, 150, 250]))
    n_reads = int(rng.integers(3_000, 120_000))
    genome = synth.make_genome(int(rng.integers(30_000, 600_000)), seed=300 + seed)
    sub = float(rng.choice([0.0, 0.005, 0.03]))
    fq = synth.make_fastq(genome, n_reads, read_len, seed=400 + seed, sub_rate=sub, device="cpu").numpy()
    rb = synth.record_bytes(read_len)
    bounds = _shards(n_reads, R, rng, uneven=bool(seed % 2))
    dev = torch.from_numpy(fq).cuda()
    torch.cuda.synchronize()
    want_h, want_c = mo.bruteforce_sketch(_seqs(fq.tobytes()), k, s, m)

    sks, hdrs = [], []
    for lo, hi in bounds:
        sk = engine.Sketcher(k, s, m, expected_bytes=max(1, (hi - lo) * rb))
        if hi > lo:
            sk.push_device(dev.data_ptr() + lo * rb, (hi - lo) * rb, engine.FMT_FASTQ4)
        sk.sync()
        sks.append(sk)
        hdrs.append(sk.export_begin())
    all_hdr = np.stack(hdrs)
    cap = max(1024, (int(all_hdr[:, 0].max()) + 1023) // 1024 * 1024)
    words = cap + cap // 2
    gathered = torch.zeros(R * words, dtype=torch.int64, device="cuda")
    for r, sk in enumerate(sks):
        sk.export_pack(gathered.data_ptr() + r * words * 8, cap)
    torch.cuda.synchronize()
    inexact = None
    for r in sorted({0, R - 1, int(rng.integers(0, R))}):
        try:
            h, c = sks[r].merge_slabs(gathered.data_ptr(), True, R, cap, all_hdr, r)
        except engine.EngineError as e:           # fewer than s solid below a lowered T_min: every rank must say so
            assert e.code == engine.MHX_E_CAPACITY
            assert inexact in (None, True)
            inexact = True
            continue
        assert inexact in (None, False)
        inexact = False
        assert np.array_equal(h, want_h), (seed, k, s, m, R, r)
        assert np.array_equal(c, want_c), (seed, k, s, m, R, r)
    # the one-collective form on fresh sketchers: header-carrying slabs, written straight into the gathered buffer
    for sk in sks:
        sk.close()
    if inexact:
        return
    words2 = 8 + cap + cap // 2
    gathered2 = torch.zeros(R * words2, dtype=torch.int64, device="cuda")
    sks = []
    for r, (lo, hi) in enumerate(bounds):
        sk = engine.Sketcher(k, s, m, expected_bytes=max(1, (hi - lo) * rb))
        if hi > lo:
            sk.push_device(dev.data_ptr() + lo * rb, (hi - lo) * rb, engine.FMT_FASTQ4)
        sk.sync()
        sk.export_into(gathered2.data_ptr() + r * words2 * 8, cap)
        sks.append(sk)
    torch.cuda.synchronize()
    for r in sorted({0, R - 1}):
        h, c, need = sks[r].merge_gathered(gathered2.data_ptr(), R, cap, r)
        assert need == 0
        assert np.array_equal(h, want_h) and np.array_equal(c, want_c), (seed, "one collective", r)
    # a capacity that is too small is reported, with the size that is needed
    small = max(2, (int(all_hdr[:, 0].max()) // 2) & ~1)
    if small < int(all_hdr[:, 0].max()):
        words3 = 8 + small + small // 2
        g3 = torch.zeros(R * words3, dtype=torch.int64, device="cuda")
        probe = []
        for r, (lo, hi) in enumerate(bounds):
            sk = engine.Sketcher(k, s, m, expected_bytes=max(1, (hi - lo) * rb))
            if hi > lo:
                sk.push_device(dev.data_ptr() + lo * rb, (hi - lo) * rb, engine.FMT_FASTQ4)
            sk.sync()
            sk.export_into(g3.data_ptr() + r * words3 * 8, small)
            probe.append(sk)
        torch.cuda.synchronize()
        h, c, need = probe[0].merge_gathered(g3.data_ptr(), R, small, 0)
        assert h is None and need == int(all_hdr[:, 0].max())
        for sk in probe:
            sk.close()
    for sk in sks:
        sk.close()
