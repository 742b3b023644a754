"""BASELINE.json configurations at their full sizes on the GPU.

The oracle finishes 2 M reads in seconds, so exact oracle parity is checked there; at the full
10 M reads the checks are size-independent properties of the sketch (bottom-s of a union equals
the merge of the shards' partial results, any order, any split; counts add up)."""
import numpy as np
import pytest
import torch

from auriclass_amd import engine, synth
from oracle import mash_oracle as mo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3_reads():
    engine.build()
    engine.init(0)
    genome = synth.make_genome(12_000_000, seed=42)
    fq = synth.make_fastq(genome, 10_000_000, 150, seed=43, device="cuda")
    torch.cuda.synchronize()
    return fq


def gpu_sketch(fq, lo, hi, k, s, m):
    sk = engine.Sketcher(k, s, m, expected_bytes=hi - lo)
    sk.push_device(fq.data_ptr() + lo, hi - lo, engine.FMT_FASTQ4)
    out = sk.finish()
    st = sk.stats()
    thr = sk.threshold()
    part = sk.export(thr)
    sk.close()
    return out, st, thr, part


@pytest.mark.parametrize("m", [1, 3])
def test_c3_oracle_parity_on_2M_reads(c3_reads, m):
    rb = synth.record_bytes(150)
    n = 2_000_000
    (got, cnt), st, _, _ = gpu_sketch(c3_reads, 0, n * rb, 21, 1000, m)
    ref = mo.Sketcher(21, 1000, m)
    ref.add_fastx(c3_reads[: n * rb].cpu().numpy().tobytes())
    want, _ = ref.finish()
    assert np.array_equal(got, want)
    assert st["kmers"] == ref.kmers == n * 130 and st["lines"] == 4 * n and st["flags"] == 0
    assert cnt.min() >= m


@pytest.mark.parametrize("m,s", [(1, 1000), (3, 1000), (3, 50_000)])
def test_c3_full_10M_reads_shard_merge_property(c3_reads, m, s):
    rb = synth.record_bytes(150)
    nbytes = c3_reads.numel()
    (full, full_cnt), st, _, _ = gpu_sketch(c3_reads, 0, nbytes, 21, s, m)
    assert len(full) == s and st["kmers"] == 10_000_000 * 130 and st["flags"] == 0
    assert np.all(np.diff(full.astype(object)) > 0)
    # uneven 5-way record split, merged through the multi-GPU export path
    cuts = [0, 1_000_000, 1_000_001, 4_500_000, 7_777_777, 10_000_000]
    parts, thr = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        _, _, t, _ = gpu_sketch(c3_reads, a * rb, b * rb, 21, s, m)
        thr.append(t)
    tmin = min(thr)
    for a, b in zip(cuts[:-1], cuts[1:]):
        sk = engine.Sketcher(21, s, m, expected_bytes=(b - a) * rb)
        sk.push_device(c3_reads.data_ptr() + a * rb, (b - a) * rb, engine.FMT_FASTQ4)
        parts.append(sk.export(tmin))
        sk.close()
    h = np.concatenate([p[0] for p in parts])
    c = np.concatenate([p[1] for p in parts])
    merged, merged_cnt = engine.merge_partials(h, c, s, m)
    assert np.array_equal(merged, full)
    assert np.array_equal(merged_cnt, full_cnt)


def test_c5_full_size_batched_distances_sampled_against_oracle():
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import dist_c5

    engine.init(0)
    Q, q_len, R, r_len = dist_c5.make_c5(1024, 24, 50_000)
    pairs = 1024 * 24
    common = torch.zeros(pairs, dtype=torch.int32, device="cuda")
    denom = torch.zeros(pairs, dtype=torch.int32, device="cuda")
    dist = torch.zeros(pairs, dtype=torch.float64, device="cuda")
    engine.dist_batch_device(Q.data_ptr(), q_len.data_ptr(), 1024, R.data_ptr(), r_len.data_ptr(), 24, 50_000, 27, 50_000,
                             common.data_ptr(), denom.data_ptr(), dist.data_ptr())
    c, d, dd = common.cpu().numpy(), denom.cpu().numpy(), dist.cpu().numpy()
    Qh, Rh = Q.cpu().numpy().view(np.uint64), R.cpu().numpy().view(np.uint64)
    ql, rl = q_len.cpu().numpy(), r_len.cpu().numpy()
    rng = np.random.default_rng(3)
    for p in rng.choice(pairs, size=400, replace=False):
        qi, ri = divmod(int(p), 24)
        wc, wd, wdist = mo.compare(Rh[ri, :rl[ri]], Qh[qi, :ql[qi]], 50_000, 27)
        assert (int(c[p]), int(d[p])) == (wc, wd)
        assert abs(dd[p] - wdist) <= 2e-16 * max(1.0, abs(wdist)) + 1e-300   # device log(): <= 1 ulp (host path is exact)
    # identities that hold for every pair
    assert np.all(d <= 50_000) and np.all(c <= d)
