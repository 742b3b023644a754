"""GPU parity tests: every call goes through the C ABI of libmhx.so (auriclass_amd.engine)
and is compared bit for bit with the CPU oracle and with the reference's golden fixtures."""
import numpy as np
import pytest

from auriclass_amd import engine, synth
from oracle import mash_oracle as mo
from tests.conftest import GOLDEN, REFDATA

pytestmark = pytest.mark.gpu

import os

FUZZ = int(os.environ.get("MHX_FUZZ_SCALE", "1"))   # MHX_FUZZ_SCALE=10: ten times the randomised cases
FUZZ0 = int(os.environ.get("MHX_FUZZ_OFFSET", "0"))  # first seed index: other values explore other cases


def _seeds(n):
    return range(FUZZ0, FUZZ0 + n * FUZZ)


@pytest.fixture(scope="module", autouse=True)
def _engine():
    engine.build()
    engine.init(0)


def oracle_sketch(data: bytes, k, s, m):
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(data)
    return ref, ref.finish()


def random_reads(rng, n, lo, hi, p_n=0.1, p_lower=0.1):
    out = []
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        r = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L)
        if L and rng.random() < p_n:
            r[int(rng.integers(0, L))] = ord("N")
        r = bytes(r)
        out.append(r.lower() if rng.random() < p_lower else r)
    return out


# ---- buffer level -------------------------------------------------------------------------
@pytest.mark.parametrize("k,s,m", [(21, 1000, 1), (21, 1000, 3), (27, 50000, 1), (16, 500, 1), (11, 300, 2), (32, 200, 1), (5, 100, 1)])
def test_seq_stream_equals_oracle(k, s, m):
    rng = np.random.default_rng(k * 7 + s + m)
    genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=40_000)
    reads = []
    for _ in range(3000):
        st = int(rng.integers(0, len(genome) - 200))
        reads.append(bytes(genome[st:st + int(rng.integers(5, 200))]))
    reads += random_reads(rng, 200, 0, 300)
    data = b"\n".join(reads) + b"\n"
    sk = engine.Sketcher(k, s, m, expected_bytes=len(data))
    sk.push_host(data, engine.FMT_SEQ)
    got_h, got_c = sk.finish()
    stats = sk.stats()
    sk.close()
    want_h, want_c = mo.bruteforce_sketch(reads, k, s, m)
    assert np.array_equal(got_h, want_h)
    assert np.array_equal(got_c, want_c)          # device counts are exact multiplicities
    ref = mo.Sketcher(k, s, m)
    for r in reads:
        ref.add_seq(r)
    assert np.array_equal(got_h, ref.finish()[0])
    # windows hashed: every start with k bytes inside one record (those holding a non-ACGT byte are dropped by the
    # deferred base check before they reach the table)
    assert stats["kmers"] == sum(max(0, len(r) - k + 1) for r in reads) >= ref.kmers
    assert stats["flags"] == 0


@pytest.mark.parametrize("m", [1, 3])
def test_fastq4_stream_equals_oracle(m):
    genome = synth.make_genome(300_000, seed=1)
    fq = synth.make_fastq(genome, 60_000, 150, seed=2, device="cpu").numpy()
    sk = engine.Sketcher(21, 1000, m, expected_bytes=fq.size)
    sk.push_host(fq, engine.FMT_FASTQ4)
    got, cnt = sk.finish()
    st = sk.stats()
    sk.close()
    ref, (want, _) = oracle_sketch(fq.tobytes(), 21, 1000, m)
    assert len(got) == 1000
    assert np.array_equal(got, want)
    assert st["kmers"] == ref.kmers and st["lines"] == 4 * 60_000 and st["flags"] == 0
    assert st["launches"] >= 2   # the tightening schedule really ran


def test_fastq4_ragged_reads_and_unaligned_device_pointer():
    import torch

    rng = np.random.default_rng(11)
    reads = random_reads(rng, 4000, 1, 400, p_n=0.3, p_lower=0.3)
    quals = np.frombuffer(b"!#+@ACGTIJ5<?acgt", np.uint8)
    rec = [b"@r%d x y\n" % i + r + b"\n+\n" + bytes(rng.choice(quals, size=len(r))) + b"\n" for i, r in enumerate(reads)]
    data = b"".join(rec)
    dev = torch.empty(len(data) + 4096, dtype=torch.uint8, device="cuda")
    for lead in (0, 3, 1000 + 7):
        dev[lead:lead + len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        sk = engine.Sketcher(21, 2000, 1, expected_bytes=len(data))
        sk.push_device(dev.data_ptr() + lead, len(data), engine.FMT_FASTQ4)
        got, _ = sk.finish()
        sk.close()
        want, _ = mo.bruteforce_sketch(reads, 21, 2000, 1)
        assert np.array_equal(got, want)


def test_multiple_pushes_accumulate_into_one_reference():
    genome = synth.make_genome(100_000, seed=3)
    a = synth.make_fastq(genome, 20_000, 150, seed=4, device="cpu").numpy()
    b = synth.make_fastq(genome, 30_000, 100, seed=5, device="cpu").numpy()
    sk = engine.Sketcher(21, 1000, 3, expected_bytes=a.size + b.size)
    sk.push_host(a, engine.FMT_FASTQ4)
    sk.push_host(b, engine.FMT_FASTQ4)
    got, _ = sk.finish()
    sk.reset()
    sk.push_host(b, engine.FMT_FASTQ4)
    again, _ = sk.finish()
    sk.close()
    _, (want, _) = oracle_sketch(a.tobytes() + b.tobytes(), 21, 1000, 3)
    assert np.array_equal(got, want)
    _, (want_b, _) = oracle_sketch(b.tobytes(), 21, 1000, 3)
    assert np.array_equal(again, want_b)


def test_non_fastq4_is_flagged_not_mis_sketched():
    data = b"@r\nACGTACGTACGTACGTACGTACGTACGT\nACGTACGTACGTACGTACGTACGT\n+\n" + b"I" * 52 + b"\n"
    sk = engine.Sketcher(21, 100, 1, expected_bytes=len(data))
    sk.push_host(data * 50, engine.FMT_FASTQ4)
    with pytest.raises(engine.EngineError) as e:
        sk.finish()
    assert e.value.code == engine.MHX_E_FORMAT
    sk.close()


def test_shard_export_and_merge_equals_single_sketch():
    genome = synth.make_genome(150_000, seed=6)
    fq = synth.make_fastq(genome, 40_000, 150, seed=7, device="cpu").numpy()
    rb = synth.record_bytes(150)
    for m in (1, 3):
        parts, thr = [], []
        sks = []
        for r in range(4):
            lo, hi = r * 10_000 * rb, (r + 1) * 10_000 * rb
            sk = engine.Sketcher(21, 1000, m, expected_bytes=hi - lo)
            sk.push_host(fq[lo:hi], engine.FMT_FASTQ4)
            thr.append(sk.threshold())
            sks.append(sk)
        tmin = min(thr)
        for sk in sks:
            parts.append(sk.export(tmin))
            sk.close()
        h = np.concatenate([p[0] for p in parts])
        c = np.concatenate([p[1] for p in parts])
        got, _ = engine.merge_partials(h, c, 1000, m)
        _, (want, _) = oracle_sketch(fq.tobytes(), 21, 1000, m)
        assert np.array_equal(got, want)


# ---- file level: the reference's golden vectors ----------------------------
def test_K1_K2_reference_sketch_bytes(refcwd):
    out = refcwd / "ref.msh"
    stderr, _ = engine.sketch_files(["tests/data/NC_001416.1.fasta", "tests/data/NC_001604.1.fasta"], 27, 50000, out)
    assert out.read_bytes() == (REFDATA / "ref_sketch.msh").read_bytes()
    assert "Sketching tests/data/NC_001416.1.fasta..." in stderr


def test_K3_K4_fastq_workflow_rows(refcwd, golden):
    out = refcwd / "q.msh"
    stderr, est = engine.sketch_files(["tests/data/NC_001416.1_1.fq.gz", "tests/data/NC_001416.1_2.fq.gz"], 27, 50000, out,
                                      reads=True, min_mult=3)
    assert "Estimated genome size: 48454.7\n" in stderr         # test_correct_workflow.py:99
    assert float("%g" % est) == golden["scalars"]["fastq_estimated_genome_size"]
    text = engine.dist_files("tests/data/ref_sketch.msh", out)
    assert text == ("tests/data/NC_001416.1.fasta\ttests/data/NC_001416.1_1.fq.gz\t9.55405e-06\t0\t48451/48476\n"
                    "tests/data/NC_001604.1.fasta\ttests/data/NC_001416.1_1.fq.gz\t1\t1\t0/50000\n")
    # same bytes as the oracle's container for the reads-mode sketch
    osk, _ = mo.sketch_files(["tests/data/NC_001416.1_1.fq.gz", "tests/data/NC_001416.1_2.fq.gz"], 27, 50000, reads=True, m=3)
    assert out.read_bytes() == mo.msh_bytes(osk)


def test_estimated_coverage_line_is_the_mean_exact_multiplicity(refcwd):
    """`Estimated coverage:` (re-logged by /root/reference/auriclass/classes.py:602-606, never parsed): mash prints the mean
    of its heap's counters, which miss repeats of the heap's current maximum and so depend on the input order (the oracle
    reproduces that: 39.125 on the fixture); the engine prints the mean of the EXACT multiplicities of the sketch's
    hashes.  Stated in INTEGRATION.md section 2; this pins the engine's definition and its relation to mash's."""
    import gzip

    files = ["tests/data/NC_001416.1_1.fq.gz", "tests/data/NC_001416.1_2.fq.gz"]
    stderr, _ = engine.sketch_files(files, 27, 50000, refcwd / "c.msh", reads=True, min_mult=3)
    line = [l for l in stderr.splitlines() if l.startswith("Estimated coverage:")]
    assert len(line) == 1 and line[0].startswith("Estimated coverage:    ")
    got = float(line[0].split()[-1])
    reads = []
    for f in files:
        rec = gzip.open(f, "rb").read().split(b"\n")
        reads += rec[1::4]
    hashes, counts = mo.bruteforce_sketch(reads, 27, 50000, 3)
    assert line[0] == "Estimated coverage:    " + mo.fmt_g(counts.sum() / len(hashes))
    osk_sketcher = mo.Sketcher(27, 50000, 3)
    for f in files:
        osk_sketcher.add_fastx(gzip.open(f, "rb").read())
    osk_sketcher.finish()
    assert got >= osk_sketcher.multiplicity > 0.9 * got   # mash's figure is a slight under-count of the same quantity


def test_third_party_sketch_containers_reference_list_and_counts(tmp_path):
    """`.msh` files this engine did not write (docs/reference_genomes.md:3: the bundled clade references come from
    elsewhere): a hash seed other than 42 moves the references to the `referenceList` pointer (the root's fourth) and
    `mash sketch -M` adds a counts32 list to every reference.  Both must read back to the same distance table."""
    rng = np.random.default_rng(99)

    def refs(n, hi, with_counts):
        out = []
        for i in range(n):
            h = np.unique(rng.integers(0, hi, size=int(rng.integers(200, 1500)), dtype=np.uint64))
            c = rng.integers(1, 200, size=len(h)).astype(np.uint32) if with_counts else None
            out.append(mo.Reference("ref%d.fa" % i, "comment %d" % i, int(rng.integers(1000, 10 ** 7)), h, c))
        return out

    for k, seed, with_counts in ((21, 7, False), (21, 42, True), (27, 1234567, True), (16, 3, True)):
        hi = 2 ** 32 if k <= 16 else 2 ** 64
        shared = refs(3, hi, with_counts)
        R = mo.SketchFile(kmer_size=k, sketch_size=1000, hash_seed=seed, references=shared + refs(2, hi, with_counts))
        Q = mo.SketchFile(kmer_size=k, sketch_size=1000, hash_seed=seed, references=refs(2, hi, with_counts) + shared[:1])
        (tmp_path / "r.msh").write_bytes(mo.msh_bytes(R))
        (tmp_path / "q.msh").write_bytes(mo.msh_bytes(Q))
        assert engine.dist_files(tmp_path / "r.msh", tmp_path / "q.msh") == mo.dist_text(R, Q), (k, seed, with_counts)
    # different seeds: mash refuses to compare, so does the engine
    A = mo.SketchFile(kmer_size=21, sketch_size=1000, hash_seed=7, references=refs(1, 2 ** 64, False))
    B = mo.SketchFile(kmer_size=21, sketch_size=1000, hash_seed=42, references=refs(1, 2 ** 64, False))
    (tmp_path / "a.msh").write_bytes(mo.msh_bytes(A))
    (tmp_path / "b.msh").write_bytes(mo.msh_bytes(B))
    with pytest.raises(engine.EngineError) as e:
        engine.dist_files(tmp_path / "a.msh", tmp_path / "b.msh")
    assert e.value.code == engine.MHX_E_MISMATCH


def test_K5_fasta_workflow_rows(refcwd):
    out = refcwd / "q.msh"
    engine.sketch_files(["tests/data/NC_001416.1.fasta.gz"], 27, 50000, out)
    text = engine.dist_files("tests/data/ref_sketch.msh", out)
    assert text == ("tests/data/NC_001416.1.fasta\ttests/data/NC_001416.1.fasta.gz\t0\t0\t48476/48476\n"
                    "tests/data/NC_001604.1.fasta\ttests/data/NC_001416.1.fasta.gz\t1\t1\t0/50000\n")


def test_K8_empty_input_raises_no_records(refcwd):
    with pytest.raises(engine.NoRecordsError, match="ERROR: Did not find fasta records in"):
        engine.sketch_files(["tests/data/test_empty_1.fq.gz", "tests/data/test_empty_2.fq.gz"], 27, 50000, refcwd / "e.msh",
                            reads=True, min_mult=3)


def test_multiline_fastq_falls_back_to_record_parser(tmp_path):
    rng = np.random.default_rng(5)
    reads = random_reads(rng, 300, 60, 200, p_n=0, p_lower=0)
    rec = []
    for i, r in enumerate(reads):
        half = len(r) // 2
        rec.append(b"@m%d\n" % i + r[:half] + b"\n" + r[half:] + b"\n+\n" + b"I" * half + b"\n" + b"I" * (len(r) - half) + b"\n")
    p = tmp_path / "multi.fq"
    p.write_bytes(b"".join(rec))
    engine.sketch_files([p], 21, 1000, tmp_path / "m.msh", reads=True, min_mult=1)
    got = mo.read_msh(tmp_path / "m.msh").references[0].hashes
    want, _ = mo.bruteforce_sketch(reads, 21, 1000, 1)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("k,s", [(21, 1000), (27, 50000)])
def test_C2_synthetic_assembly_vs_24_refs(tmp_path, k, s):
    """BASELINE.json config 2 with the stand-ins of SURVEY.md 8(d): a 12 Mb genome in 20 contigs
    (70-column FASTA), 24 references mutated at rates 0.0005 .. 0.05, sketched by the oracle; the
    engine sketches the query and computes the 24 distance rows.  (k=21, s=1000) and AuriClass's
    defaults (k=27, s=50000)."""
    genome = synth.make_genome(12_000_000, seed=42)
    fa = tmp_path / "asm.fasta"
    fa.write_bytes(synth.genome_fasta(genome, 20))
    engine.sketch_files([fa], k, s, tmp_path / "a.msh")
    osk, _ = mo.sketch_files([fa], k, s)
    assert (tmp_path / "a.msh").read_bytes() == mo.msh_bytes(osk)
    assert osk.references[0].length == 12_000_000 and len(osk.references[0].hashes) == s
    rates = np.geomspace(0.0005, 0.05, 24)
    refs = []
    for i, rate in enumerate(rates):
        g = synth.mutate(genome, float(rate), 100 + i)
        rs = mo.Sketcher(k, s, 1)
        rs.add_seq(g.tobytes())
        refs.append(mo.Reference("ref_%02d.fasta" % i, "synthetic clade ref %d" % i, len(g), rs.finish()[0]))
    rsk = mo.SketchFile(kmer_size=k, sketch_size=s, references=refs)
    mo.write_msh(tmp_path / "refs.msh", rsk)
    text = engine.dist_files(tmp_path / "refs.msh", tmp_path / "a.msh")
    assert text == mo.dist_text(rsk, osk)
    d = [float(l.split("\t")[2]) for l in text.splitlines()]
    assert len(d) == 24 and d[0] < d[-1] and d[0] < 0.002        # closest ref first, distances grow with the mutation rate


# ---- batched distance -------------------------------------------------------------------------
def test_dist_batch_equals_oracle_compare():
    rng = np.random.default_rng(9)
    s, k = 3000, 21
    base = np.unique(rng.integers(0, 2 ** 63, size=4 * s, dtype=np.uint64))
    refs, qrys = [], []
    for i in range(5):
        refs.append(np.sort(rng.choice(base, size=s if i else s - 500, replace=False)))
    for f in (0.0, 0.01, 0.3, 0.9, 1.0):
        src = refs[int(rng.integers(0, 5))]
        keep = src[rng.random(len(src)) >= f]
        fresh = rng.integers(0, 2 ** 63, size=s - len(keep), dtype=np.uint64)
        qrys.append(np.unique(np.concatenate([keep, fresh])))
    qrys.append(refs[0].copy())                        # identical lists
    qrys.append(np.zeros(0, np.uint64))                # empty query
    qrys.append(refs[1][:10].copy())                   # short prefix
    stride = max(max(map(len, refs)), max(map(len, qrys)))
    R = np.zeros((len(refs), stride), np.uint64)
    Q = np.zeros((len(qrys), stride), np.uint64)
    for i, r in enumerate(refs):
        R[i, :len(r)] = r
    for i, q in enumerate(qrys):
        Q[i, :len(q)] = q
    common, denom, dist = engine.dist_batch(Q, [len(q) for q in qrys], R, [len(r) for r in refs], k, s)
    for qi, q in enumerate(qrys):
        for ri, r in enumerate(refs):
            c, d, dd = mo.compare(r, q, s, k)
            assert (common[qi, ri], denom[qi, ri]) == (c, d), (qi, ri)
            assert dist[qi, ri] == dd


def _pad_rows(lists, stride):
    M = np.zeros((len(lists), stride), np.uint64)
    for i, v in enumerate(lists):
        M[i, :len(v)] = v
    return M, np.array([len(v) for v in lists], np.uint32)


def _check_all_pairs(qrys, refs, k, s, pad=1):
    stride = (max(max(map(len, refs)), max(map(len, qrys)), 1) + pad - 1) // pad * pad
    Q, ql = _pad_rows(qrys, stride)
    R, rl = _pad_rows(refs, stride)
    common, denom, dist = engine.dist_batch(Q, ql, R, rl, k, s)
    for qi, q in enumerate(qrys):
        for ri, r in enumerate(refs):
            c, d, dd = mo.compare(r, q, s, k)
            assert (common[qi, ri], denom[qi, ri]) == (c, d), (qi, ri, len(q), len(r))
            assert dist[qi, ri] == dd


def _sketch_like(rng, n, hi=2 ** 64):
    return np.unique(rng.integers(0, hi, size=n, dtype=np.uint64))


def test_dist_all_vs_refs_fast_path_equals_oracle():
    """C5-shaped, reduced: 40 queries x 24 refs, s = 6000 (all-vs-refs LDS path)."""
    rng = np.random.default_rng(21)
    s = 6000
    base = _sketch_like(rng, s)
    refs = []
    for j in range(24):
        keep = rng.random(len(base)) >= (0.002 * (j + 1) if j < 11 else 0.5)
        refs.append(np.unique(np.concatenate([base[keep], _sketch_like(rng, int((~keep).sum()))])))
    refs[3] = refs[3][:s - 900]                       # a reference shorter than s
    qrys = []
    for i in range(40):
        src = refs[i % 24]
        keep = rng.random(len(src)) >= 0.6 * i / 39
        qrys.append(np.unique(np.concatenate([src[keep], _sketch_like(rng, int((~keep).sum()))])))
    qrys[5] = refs[5].copy()
    qrys[6] = qrys[6][:17]
    qrys[7] = np.zeros(0, np.uint64)
    _check_all_pairs(qrys, refs, 27, s)


@pytest.mark.parametrize("form", ["lane", "lane64", "walk", "wave"])
def test_dist_one_query_per_lane_forms_equal_oracle(monkeypatch, form):
    """Batches of >= 128 queries take the range pass with ONE QUERY PER LANE (dist_range_lane_kernel), large ones the walk
    over consecutive ranges without a split pass over the queries (dist_walk_kernel; forced here by MHX_DIST_WALK_MIN);
    `wave` is the slice-per-wave kernel of smaller batches on the same data.  Edge cases in the batch: empty and tiny
    queries, a query equal to a reference, queries that end inside the value space (their last ranges are empty), one
    whose hashes all lie in the upper half (its first ranges are empty)."""
    if form == "walk":
        monkeypatch.setenv("MHX_DIST_WALK_MIN", "128")
    if form == "wave":
        monkeypatch.setenv("MHX_DIST_NO_LANE", "1")
    rng = np.random.default_rng(41)
    s = 4000
    base = _sketch_like(rng, s)
    refs = []
    for j in range(24):
        keep = rng.random(len(base)) >= (0.002 * (j + 1) if j < 11 else 0.5)
        refs.append(np.unique(np.concatenate([base[keep], _sketch_like(rng, int((~keep).sum()))])))
    qrys = []
    for i in range(150):
        src = refs[i % 24]
        keep = rng.random(len(src)) >= 0.6 * i / 149
        qrys.append(np.unique(np.concatenate([src[keep], _sketch_like(rng, int((~keep).sum()))])))
    qrys[5] = refs[5].copy()
    qrys[6] = qrys[6][:17]
    qrys[7] = np.zeros(0, np.uint64)
    qrys[8] = qrys[8][:len(qrys[8]) // 3]                       # ends a third of the way through the value space
    qrys[9] = qrys[9][qrys[9] >= np.uint64(1 << 63)]            # nothing in the lower half
    qrys[10] = qrys[10][::7]
    # rows of whole 128-byte L2 lines: what the walk form and the lane form's main path ask for; `lane64`: rows of whole
    # 64-byte lines only (an odd number of them), the lane form's narrower path
    stride = (max(max(map(len, refs)), max(map(len, qrys))) + 15) // 16 * 16
    if form == "lane64":
        stride += 8
    Q, ql = _pad_rows(qrys, stride)
    R, rl = _pad_rows(refs, stride)
    common, denom, dist = engine.dist_batch(Q, ql, R, rl, 27, s)
    for qi in list(range(12)) + list(range(12, 150, 9)):
        for ri, r in enumerate(refs):
            c, d, dd = mo.compare(r, qrys[qi], s, 27)
            assert (common[qi, ri], denom[qi, ri]) == (c, d), (form, qi, ri, len(qrys[qi]))
            assert dist[qi, ri] == dd


@pytest.mark.parametrize("form", ["wave", "lane", "walk"])
def test_dist_non_uniform_values_fall_back_to_the_generic_kernel(monkeypatch, form):
    """All hashes crowded into one narrow value range overflow the per-range LDS table (more distinct keys than it may
    hold); the engine must notice and still return exact results (generic pair kernel) -- in every form of the range
    pass: slice per wave (10 queries), one query per lane (140), the walk over consecutive ranges (140, forced)."""
    if form == "walk":
        monkeypatch.setenv("MHX_DIST_WALK_MIN", "128")
    rng = np.random.default_rng(22)
    lo = 1 << 62
    refs = [lo + _sketch_like(rng, 3000, hi=2 ** 20) for _ in range(8)]
    refs.append(np.concatenate([refs[0][:1000], np.array([2 ** 64 - 5], np.uint64)]))   # one far outlier sets the scale
    nq = 10 if form == "wave" else 140
    qrys = [np.unique(np.concatenate([refs[i % 8][::2], lo + _sketch_like(rng, 1500, hi=2 ** 20)])) for i in range(nq)]
    if form == "wave":
        _check_all_pairs(qrys, refs, 21, 3000)
    else:
        stride = (max(max(map(len, refs)), max(map(len, qrys))) + 15) // 16 * 16
        Q, ql = _pad_rows(qrys, stride)
        R, rl = _pad_rows(refs, stride)
        common, denom, dist = engine.dist_batch(Q, ql, R, rl, 21, 3000)
        for qi in range(0, nq, 9):
            for ri, r in enumerate(refs):
                c, d, dd = mo.compare(r, qrys[qi], 3000, 21)
                assert (common[qi, ri], denom[qi, ri]) == (c, d), (form, qi, ri)
    assert engine.load().mhx_last_dist_fallback_blocks() >= 1   # (one block; more when MHX_DIST_QBATCH cuts the queries into batches)


def test_dist_more_than_32_refs_and_k16_32bit_hashes():
    rng = np.random.default_rng(23)
    refs = [_sketch_like(rng, 800, hi=2 ** 32) for _ in range(40)]
    qrys = [np.unique(np.concatenate([refs[i][:400], _sketch_like(rng, 400, hi=2 ** 32)])) for i in range(6)]
    _check_all_pairs(qrys, refs, 16, 800)
    _check_all_pairs(qrys, refs[:24], 16, 800)          # fast path on 32-bit hash values


@pytest.mark.parametrize("inflater", ["own", "zlib"])
def test_streaming_ingest_many_chunks_two_gz_files(tmp_path, monkeypatch, inflater):
    """Reads mode streams each file through its own inflate thread in 32 MiB record-aligned
    chunks; two files, several chunks each, the second one without a final newline.  The .gz goes
    through the engine's own DEFLATE decoder (matches reach back across chunk borders) or, with
    MHX_ZLIB_INFLATE=1, through zlib."""
    import gzip

    if inflater == "zlib":
        monkeypatch.setenv("MHX_ZLIB_INFLATE", "1")

    genome = synth.make_genome(400_000, seed=8)
    a = synth.make_fastq(genome, 230_000, 150, seed=9, device="cpu").numpy().tobytes()       # 72 MB -> 3 chunks
    b = synth.make_fastq(genome, 120_000, 150, seed=10, device="cpu", first_index=230_000).numpy().tobytes()
    b += b"".join(b"@short%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(7))      # shorter than k: not counted by mash
    b = b[:-1]
    pa, pb = tmp_path / "r1.fq.gz", tmp_path / "r2.fq"
    with gzip.open(pa, "wb", compresslevel=6) as fh:
        fh.write(a)
    pb.write_bytes(b)
    out = tmp_path / "s.msh"
    stderr, est = engine.sketch_files([pa, pb], 21, 5000, out, reads=True, min_mult=2)
    ref = mo.Sketcher(21, 5000, 2)
    ref.add_fastx(a)
    ref.add_fastx(b)
    want, _ = ref.finish()
    got = mo.read_msh(out)
    assert np.array_equal(got.references[0].hashes, want)
    assert ref.records == 350_000
    assert got.references[0].comment == ref.comment() == "[350000 seqs] r00000000  [...]"
    assert got.references[0].length == int(ref.set_size)
    assert "Estimated genome size: %g\n" % ref.set_size in stderr


@pytest.mark.parametrize("k,s", [(16, 400), (11, 200), (32, 300), (21, 1000)])
def test_file_level_multi_record_fasta_with_iupac_and_lowercase(tmp_path, k, s):
    """Unpinned-by-reference corners at the file boundary, engine vs oracle byte for byte: multi-record
    FASTA ('[N seqs] ... [...]' comment), records shorter than k, N / IUPAC / lower-case bases, CRLF
    line ends, 32-bit hash lists (k <= 16) and k = 32."""
    rng = np.random.default_rng(k * 31 + s)
    recs = []
    for i in range(40):
        L = int(rng.integers(5, 900))
        seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L)
        for _ in range(int(rng.integers(0, 3))):
            seq[int(rng.integers(0, L))] = rng.choice(np.frombuffer(b"NRYKMSW", np.uint8))
        seq = bytes(seq)
        if i % 3 == 0:
            seq = seq.lower()
        wrapped = b"\n".join(seq[j:j + 60] for j in range(0, L, 60))
        recs.append(b">rec%d some description %d\n" % (i, i) + wrapped + b"\n")
    fa = tmp_path / "multi.fa"
    fa.write_bytes(b"".join(recs))
    crlf = tmp_path / "crlf.fa"
    crlf.write_bytes(b"".join(recs[:5]).replace(b"\n", b"\r\n"))
    engine.sketch_files([fa, crlf], k, s, tmp_path / "e.msh")
    osk, _ = mo.sketch_files([fa, crlf], k, s)
    assert (tmp_path / "e.msh").read_bytes() == mo.msh_bytes(osk)
    assert osk.references[0].comment.startswith("[") and " seqs] rec" in osk.references[0].comment
    if k <= 16:
        assert max(int(r.hashes.max()) for r in osk.references) < 2 ** 32
    # the engine reads its own container back for distances
    assert engine.dist_files(tmp_path / "e.msh", tmp_path / "e.msh") == mo.dist_text(osk, osk)


def test_low_coverage_reads_with_multiplicity_filter_retry_path(tmp_path):
    """Coverage 0.3 with m = 3: fewer qualifying k-mers lie below the initial admission bound than
    the sketch needs, so the first attempt must be rejected (MHX_E_CAPACITY inside the library) and
    the file-level call must come back with the exact sketch from a wider bound."""
    genome = synth.make_genome(60_000_000, seed=77)
    fq = synth.make_fastq(genome, 120_000, 150, seed=78, device="cpu").numpy()
    p = tmp_path / "lowcov.fq"
    fq.tofile(p)
    # buffer level: the tight bound is detected, not silently accepted
    sk = engine.Sketcher(21, 20_000, 3, expected_bytes=fq.size)
    sk.push_host(fq, engine.FMT_FASTQ4)
    with pytest.raises(engine.EngineError) as e:
        sk.finish()
    assert e.value.code == engine.MHX_E_CAPACITY
    sk.close()
    # file level: retried internally
    engine.sketch_files([p], 21, 20_000, tmp_path / "l.msh", reads=True, min_mult=3)
    ref = mo.Sketcher(21, 20_000, 3)
    ref.add_fastx(fq.tobytes())
    want, _ = ref.finish()
    got = mo.read_msh(tmp_path / "l.msh").references[0].hashes
    assert len(want) == 20_000
    assert np.array_equal(got, want)


def test_many_pushes_far_beyond_the_expected_size_stay_exact():
    """expected_bytes is only a hint: 40 pushes totalling 50x the hint must neither overflow the
    candidate table nor lose exactness (the engine re-tightens its threshold as the input grows)."""
    genome = synth.make_genome(2_000_000, seed=90)
    chunks = [synth.make_fastq(genome, 25_000, 150, seed=100 + i, device="cpu", first_index=i * 25_000).numpy() for i in range(40)]
    for m in (1, 2):
        sk = engine.Sketcher(21, 2000, m, expected_bytes=chunks[0].size // 2)
        for c in chunks:
            sk.push_host(c, engine.FMT_FASTQ4)
        got, _ = sk.finish()
        st = sk.stats()
        sk.close()
        ref = mo.Sketcher(21, 2000, m)
        for c in chunks:
            ref.add_fastx(c.tobytes())
        want, _ = ref.finish()
        assert np.array_equal(got, want)
        assert st["flags"] == 0


@pytest.mark.parametrize("bulk", [True, False])
def test_plain_fastq_file_bulk_and_chunked_ingest(tmp_path, monkeypatch, bulk):
    """Uncompressed FASTQ: the whole file goes to one device buffer (parallel pread -> pinned ring ->
    H2D) and is parsed there in one push; MHX_NO_BULK=1 sends it through the 32 MiB chunk queue
    instead.  100 MB file that starts with records shorter than k (mash neither counts nor names
    them) and ends without a final newline."""
    if not bulk:
        monkeypatch.setenv("MHX_NO_BULK", "1")
    genome = synth.make_genome(300_000, seed=18)
    a = b"@tiny1 x\nACGT\n+\nIIII\n@tiny2\n\n+\n\n"
    a += synth.make_fastq(genome, 330_000, 150, seed=19, device="cpu").numpy().tobytes()     # 104 MB -> 4 chunks
    a += b"@tail\nACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIII"            # no final newline
    p = tmp_path / "big.fq"
    p.write_bytes(a)
    engine.sketch_files([p], 21, 3000, tmp_path / "b.msh", reads=True, min_mult=2)
    ref = mo.Sketcher(21, 3000, 2)
    ref.add_fastx(a)
    want, _ = ref.finish()
    got = mo.read_msh(tmp_path / "b.msh").references[0]
    assert np.array_equal(got.hashes, want)
    assert got.comment == ref.comment() == "[330001 seqs] r00000000  [...]"


@pytest.mark.parametrize("k", [16, 21, 27, 32])
def test_strand_decision_exact_path_on_device(k):
    """Windows whose first 8 bases equal those of their reverse complement leave the fast 8-base strand
    comparison and take the exact dword-by-dword one (rare branch: forced here on every window)."""
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    rng = np.random.default_rng(500 + k)
    reads = []
    for _ in range(4000):
        x = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=8)
        mid = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=k - 16)
        kmer = np.concatenate([x, mid, np.array([comp[int(c)] for c in x[::-1]], np.uint8)])
        left = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(0, 10)))
        right = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(0, 10)))
        reads.append(bytes(np.concatenate([left, kmer, right])))
    data = b"\n".join(reads) + b"\n"
    sk = engine.Sketcher(k, 5000, 1, expected_bytes=len(data))
    sk.push_host(data, engine.FMT_SEQ)
    got, cnt = sk.finish()
    sk.close()
    want, wcnt = mo.bruteforce_sketch(reads, k, 5000, 1)
    assert np.array_equal(got, want) and np.array_equal(cnt, wcnt)


@pytest.mark.parametrize("coverage,s,m", [(300, 5000, 3), (40, 20000, 3), (2, 1000, 3), (150, 2000, 5)])
def test_multiplicity_filter_is_exact_at_any_coverage(coverage, s, m):
    """m > 1: before s solid hashes exist the admission threshold is capped from the bytes seen so far
    (not from the expected total), so deep samples of small genomes stay exact without a retry; very
    shallow ones (fewer than s solid k-mers) too."""
    import torch

    genome = synth.make_genome(200_000, seed=31)
    n_reads = coverage * 200_000 // 150
    fq = synth.make_fastq(genome, n_reads, 150, seed=32, device="cuda")
    torch.cuda.synchronize()
    sk = engine.Sketcher(21, s, m, expected_bytes=fq.numel())
    sk.push_device(fq.data_ptr(), fq.numel(), engine.FMT_FASTQ4)
    got, cnt = sk.finish()
    sk.close()
    ref = mo.Sketcher(21, s, m)
    ref.add_fastx(fq.cpu().numpy().tobytes())
    want, want_cnt = ref.finish()
    assert np.array_equal(got, want)
    assert cnt.min() >= m if len(cnt) else True


def test_corrupt_gz_is_reported_after_zlib_had_the_last_word(tmp_path):
    """A .fq.gz the engine's own DEFLATE decoder refuses is run again through zlib; a stream that is
    really broken then fails with mash's wording instead of producing a sketch."""
    import gzip

    genome = synth.make_genome(50_000, seed=41)
    a = synth.make_fastq(genome, 20_000, 150, seed=42, device="cpu").numpy().tobytes()
    p = tmp_path / "broken.fq.gz"
    z = bytearray(gzip.compress(a, compresslevel=6))
    z[len(z) // 2] ^= 0x55
    p.write_bytes(bytes(z))
    with pytest.raises(engine.EngineError):
        engine.sketch_files([p], 21, 1000, tmp_path / "x.msh", reads=True, min_mult=1)


@pytest.mark.parametrize("seed", _seeds(12))
def test_randomised_sweep_of_parameters_formats_and_push_patterns(seed):
    """Random k (1..32), sketch size, multiplicity, read-length distribution, N / lower-case content,
    CRLF or LF, number of pushes and device-pointer alignment; every case against the oracle, counts
    and record count included."""
    import torch

    rng = np.random.default_rng(7000 + seed)
    k = int(rng.integers(1, 33))
    m = int(rng.choice([1, 1, 2, 3, 4]))
    s = int(rng.choice([1, 50, 1000, 20000]))
    genome = synth.make_genome(int(rng.integers(2_000, 60_000)), seed=seed)
    n_reads = int(rng.integers(200, 6000))
    lo, hi = sorted(int(x) for x in rng.integers(1, 300, 2))
    reads = []
    for _ in range(n_reads):
        L = int(rng.integers(lo, hi + 1))
        p0 = int(rng.integers(0, len(genome) - L))
        r = bytearray(genome[p0:p0 + L].tobytes())
        if rng.random() < 0.2 and L:
            r[int(rng.integers(0, L))] = ord(rng.choice(list("NRYKMSW")))
        r = bytes(r)
        reads.append(r.lower() if rng.random() < 0.15 else r)
    nl = b"\r\n" if rng.random() < 0.25 else b"\n"
    quals = np.frombuffer(b"!#+@ACGTIJ5<?acgt", np.uint8)
    recs = [b"@r%d d" % i + nl + r + nl + b"+" + nl + bytes(rng.choice(quals, size=len(r))) + nl for i, r in enumerate(reads)]
    n_push = int(rng.integers(1, 5))
    cuts = sorted(set(int(x) for x in rng.integers(0, len(recs) + 1, n_push - 1))) if n_push > 1 else []
    bounds = [0] + cuts + [len(recs)]
    total = sum(len(r) for r in recs)
    expected = total if rng.random() < 0.7 else 0
    spans = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        blob = b"".join(recs[a:b])
        if not blob:
            continue
        lead = int(rng.integers(0, 40))
        dev = torch.zeros(len(blob) + lead + 64, dtype=torch.uint8, device="cuda")
        dev[lead:lead + len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
        spans.append((dev, lead, len(blob)))
    torch.cuda.synchronize()
    for scale in (1, 16, 256, 4096):     # MHX_E_CAPACITY: fewer than s solid k-mers, repeat with a larger budget
        sk = engine.Sketcher(k, s, m, expected_bytes=expected, budget_scale=scale)
        for dev, lead, n in spans:
            sk.push_device(dev.data_ptr() + lead, n, engine.FMT_FASTQ4)
        try:
            got, got_c = sk.finish()
        except engine.EngineError as e:
            sk.close()
            if e.code != engine.MHX_E_CAPACITY:
                raise
            continue
        n_long = sk.record_count()
        sk.close()
        break
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(b"".join(recs))
    want, want_c = ref.finish()
    assert np.array_equal(got, want), (k, s, m)
    assert n_long == ref.records == sum(1 for r in reads if len(r) >= k)
    if m == 1:     # with m > 1 mash's own counts depend on the insertion order; exact here
        brute_h, brute_c = mo.bruteforce_sketch([r for r in reads], k, s, m)
        assert np.array_equal(got_c, brute_c)


@pytest.mark.parametrize("seed", _seeds(8))
def test_randomised_file_level_reads_mode(tmp_path, seed):
    """Random mixes of plain and gzipped FASTQ files (1..3 files, some empty of long reads, some with
    CRLF, some without a final newline) through mhx_sketch_files in reads mode: .msh bytes and the
    stderr text against the oracle."""
    import gzip

    rng = np.random.default_rng(8100 + seed)
    k = int(rng.choice([11, 16, 21, 27, 32]))
    s = int(rng.choice([200, 1000, 50000]))
    m = int(rng.choice([1, 2, 3]))
    genome = synth.make_genome(int(rng.integers(5_000, 80_000)), seed=100 + seed)
    paths, blobs = [], []
    for fi in range(int(rng.integers(1, 4))):
        n = int(rng.integers(300, 4000))
        L = int(rng.integers(30, 260))
        data = synth.make_fastq(genome, n, L, seed=int(rng.integers(1, 10_000)), device="cpu", first_index=fi * 10_000).numpy().tobytes()
        if rng.random() < 0.3:
            data = b"@s1\nAC\n+\nII\n" + data                      # a record mash skips (shorter than k)
        if rng.random() < 0.3:
            data = data.replace(b"\n", b"\r\n")
        if rng.random() < 0.3:
            data = data[:-1] if not data.endswith(b"\r\n") else data[:-2]
        p = tmp_path / ("f%d.fq" % fi)
        if rng.random() < 0.5:
            p = tmp_path / ("f%d.fq.gz" % fi)
            with gzip.open(p, "wb", compresslevel=int(rng.integers(1, 10))) as fh:
                fh.write(data)
        else:
            p.write_bytes(data)
        paths.append(p)
        blobs.append(data)
    ref = mo.Sketcher(k, s, m)
    for b in blobs:
        ref.add_fastx(b)
    if ref.records == 0:       # every read shorter than k: nothing to sketch, mash stops with an error as well
        with pytest.raises(engine.NoRecordsError):
            engine.sketch_files(paths, k, s, tmp_path / "o.msh", reads=True, min_mult=m)
        return
    stderr, est = engine.sketch_files(paths, k, s, tmp_path / "o.msh", reads=True, min_mult=m)
    want, _ = ref.finish()
    got = mo.read_msh(tmp_path / "o.msh").references[0]
    assert np.array_equal(got.hashes, want), (k, s, m)
    assert got.comment == ref.comment()
    assert got.length == int(ref.set_size)
    assert "Estimated genome size: %g\n" % ref.set_size in stderr


@pytest.mark.parametrize("seed", _seeds(6))
def test_randomised_dist_batches(seed, monkeypatch):
    """Random batch shapes for mash's compareSketches on the device: 1..40 references (both sides of the
    32-reference fast path), ragged list lengths, shared fractions from 0 to 1, tiny and full sketches; one batch in
    three is large enough for the one-query-per-lane form, half of those take the walk form; rows padded to nothing, to
    whole 64-byte or to whole 128-byte lines (the three load paths of the lane form)."""
    rng = np.random.default_rng(8200 + seed)
    s = int(rng.choice([64, 1000, 5000]))
    k = int(rng.choice([16, 21, 27]))
    hi = 2 ** 32 if k <= 16 else 2 ** 64
    nr = int(rng.integers(1, 41))
    nq = int(rng.integers(1, 90))
    if rng.random() < 0.33:
        nq = int(rng.integers(128, 260))
        if rng.random() < 0.5:
            monkeypatch.setenv("MHX_DIST_WALK_MIN", "128")
    pad = int(rng.choice([1, 8, 16]))
    base = _sketch_like(rng, 3 * s, hi)
    refs = [np.sort(rng.choice(base, size=int(rng.integers(1, s + 1)), replace=False)) for _ in range(nr)]
    qrys = []
    for _ in range(nq):
        src = refs[int(rng.integers(0, nr))]
        f = rng.random()
        keep = src[rng.random(len(src)) >= f]
        fresh = _sketch_like(rng, int(rng.integers(0, s // 2 + 1)), hi)
        q = np.unique(np.concatenate([keep, fresh]))[: s]
        qrys.append(q)
    _check_all_pairs(qrys, refs, k, s, pad)


@pytest.mark.parametrize("seed", _seeds(5))
def test_randomised_medium_inputs_with_multiplicity_filter(seed):
    """3-60 MB inputs (hundreds to thousands of tiles, several tighten stages) over random genome sizes
    and coverages from well below 1x to a few hundred x, m in 1..5, s from 100 to 50000, one to three
    pushes: the capped-admission phase and the look-back at scale."""
    rng = np.random.default_rng(8300 + seed)
    k = int(rng.choice([15, 21, 27, 31]))
    m = int(rng.choice([1, 2, 3, 3, 5]))
    s = int(rng.choice([100, 1000, 5000, 50000]))
    genome = synth.make_genome(int(10 ** rng.uniform(4.0, 6.5)), seed=200 + seed)
    L = int(rng.choice([75, 100, 150, 250]))
    n_reads = int(10 ** rng.uniform(4.0, 5.3))
    fq = synth.make_fastq(genome, n_reads, L, seed=300 + seed, device="cuda", sub_rate=float(rng.choice([0.0, 0.005, 0.02])))
    import torch

    torch.cuda.synchronize()
    rb = synth.record_bytes(L)
    n_push = int(rng.integers(1, 4))
    cuts = sorted(set(int(x) for x in rng.integers(1, n_reads, n_push - 1))) if n_push > 1 else []
    bounds = [0] + cuts + [n_reads]
    expected = fq.numel() if rng.random() < 0.7 else 0
    for scale in (1, 16, 256, 4096):
        # inputs with fewer than s solid k-mers need every one of them: the engine says so (MHX_E_CAPACITY) and the
        # caller repeats with a larger admission budget, as mhx_sketch_files does by itself
        sk = engine.Sketcher(k, s, m, expected_bytes=expected, budget_scale=scale)
        for a, b in zip(bounds[:-1], bounds[1:]):
            sk.push_device(fq.data_ptr() + a * rb, (b - a) * rb, engine.FMT_FASTQ4)
        try:
            got, _ = sk.finish()
        except engine.EngineError as e:
            sk.close()
            if e.code != engine.MHX_E_CAPACITY:
                raise
            continue
        assert sk.record_count() == (n_reads if L >= k else 0)
        sk.close()
        break
    else:
        raise AssertionError("no admission budget was large enough")
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(fq.cpu().numpy().tobytes())
    want, _ = ref.finish()
    assert np.array_equal(got, want), (k, s, m, len(genome), n_reads, L)


@pytest.mark.parametrize("gz", [False, True])
def test_shallow_sample_with_multiplicity_filter_needs_every_solid_kmer(tmp_path, gz):
    """0.06x coverage with m = 3: only a few hundred k-mers are solid, fewer than s, so the sketch is ALL of
    them and the capped admission must be repeated with a larger budget: in place from the device-resident
    buffers for a plain file, through the whole-file path for a .gz."""
    import gzip

    genome = synth.make_genome(40_000_000, seed=51)
    data = synth.make_fastq(genome, 16_000, 150, seed=52, device="cpu").numpy().tobytes()      # 5 MB, 0.06x
    p = tmp_path / ("shallow.fq.gz" if gz else "shallow.fq")
    if gz:
        with gzip.open(p, "wb", compresslevel=4) as fh:
            fh.write(data)
    else:
        p.write_bytes(data)
    engine.sketch_files([p], 21, 1000, tmp_path / "o.msh", reads=True, min_mult=3)
    ref = mo.Sketcher(21, 1000, 3)
    ref.add_fastx(data)
    want, _ = ref.finish()
    got = mo.read_msh(tmp_path / "o.msh").references[0]
    assert 0 < len(want) < 1000
    assert np.array_equal(got.hashes, want)
    assert got.comment == ref.comment()


@pytest.mark.parametrize("seed", _seeds(6))
def test_randomised_sequence_streams_with_arbitrary_bytes(seed):
    """MHX_FMT_SEQ: any byte that is not A/C/G/T (either case) ends a k-mer run -- newlines, NUL, 0xFF,
    '>', '@', CR, digits.  Random streams with such bytes sprinkled in, random k / s / m, random split into
    pushes at record separators; against the brute-force definition."""
    import re

    rng = np.random.default_rng(8400 + seed)
    k = int(rng.integers(1, 33))
    s = int(rng.choice([1, 100, 2000]))
    m = int(rng.choice([1, 1, 2, 3]))
    n = int(rng.integers(2_000, 400_000))
    genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(500, 50_000)))
    pieces = []
    total = 0
    while total < n:
        L = int(rng.integers(1, 400))
        st = int(rng.integers(0, max(1, len(genome) - L)))
        r = bytearray(genome[st:st + L].tobytes())
        if rng.random() < 0.3:
            r = bytearray(bytes(r).lower())
        for _ in range(int(rng.integers(0, 3))):
            if len(r):
                r[int(rng.integers(0, len(r)))] = int(rng.choice([0, 255, 10, 13, 32, ord("N"), ord("n"), ord(">"), ord("@"), ord("5"), ord("U")]))
        pieces.append(bytes(r))
        total += len(r) + 1
    data = b"\n".join(pieces) + b"\n"
    runs = [x for x in re.split(rb"[^ACGTacgt]+", data) if x]
    for scale in (1, 16, 256, 4096):
        sk = engine.Sketcher(k, s, m, expected_bytes=len(data), budget_scale=scale)
        cut = int(rng.integers(0, len(pieces)))
        first = b"\n".join(pieces[:cut]) + (b"\n" if cut else b"")
        if first:
            sk.push_host(first, engine.FMT_SEQ)
        rest = data[len(first):]
        if rest:
            sk.push_host(rest, engine.FMT_SEQ)
        try:
            got_h, got_c = sk.finish()
        except engine.EngineError as e:
            sk.close()
            if e.code != engine.MHX_E_CAPACITY:
                raise
            continue
        sk.close()
        break
    want_h, want_c = mo.bruteforce_sketch(runs, k, s, m)
    assert np.array_equal(got_h, want_h), (k, s, m, n)
    assert np.array_equal(got_c, want_c)


@pytest.mark.parametrize("seed", _seeds(5))
def test_randomised_fasta_files(tmp_path, seed):
    """FASTA mode (one reference per file): random record counts and lengths (some shorter than k), line
    widths, lower case, IUPAC codes, blank lines between records, CRLF, plain or gzip; .msh bytes against
    the oracle, then the distance table of the file against itself."""
    import gzip

    rng = np.random.default_rng(8500 + seed)
    k = int(rng.choice([9, 16, 21, 27, 32]))
    s = int(rng.choice([10, 1000, 50000]))
    paths = []
    for fi in range(int(rng.integers(1, 4))):
        recs = []
        for i in range(int(rng.integers(1, 30))):
            L = int(rng.integers(1, 3000))
            seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L)
            for _ in range(int(rng.integers(0, 4))):
                seq[int(rng.integers(0, L))] = rng.choice(np.frombuffer(b"NRYKMSWBDHV", np.uint8))
            seq = bytes(seq)
            if rng.random() < 0.3:
                seq = seq.lower()
            w = int(rng.choice([50, 60, 70, 80, 1 << 30]))
            body = b"\n".join(seq[j:j + w] for j in range(0, L, w))
            hdr = b">c%d_%d" % (fi, i) + (b" len=%d some words" % L if rng.random() < 0.7 else b"")
            recs.append(hdr + b"\n" + body + b"\n" + (b"\n" if rng.random() < 0.1 else b""))
        data = b"".join(recs)
        if rng.random() < 0.2:
            data = data.replace(b"\n", b"\r\n")
        if rng.random() < 0.2:
            data = data.rstrip(b"\r\n")
        p = tmp_path / ("g%d.fa" % fi)
        if rng.random() < 0.5:
            p = tmp_path / ("g%d.fa.gz" % fi)
            with gzip.open(p, "wb", compresslevel=int(rng.integers(1, 10))) as fh:
                fh.write(data)
        else:
            p.write_bytes(data)
        paths.append(p)
    try:
        osk, _ = mo.sketch_files(paths, k, s)
    except Exception as e:               # a file without any record of >= k bases: both sides must refuse it
        with pytest.raises(engine.EngineError):
            engine.sketch_files(paths, k, s, tmp_path / "e.msh")
        return
    engine.sketch_files(paths, k, s, tmp_path / "e.msh")
    assert (tmp_path / "e.msh").read_bytes() == mo.msh_bytes(osk), (k, s)
    assert engine.dist_files(tmp_path / "e.msh", tmp_path / "e.msh") == mo.dist_text(osk, osk)


def test_reference_is_named_after_the_first_counted_record_across_files(tmp_path):
    """When every record of the first file is shorter than k, mash's comment names the first record of the
    next file that has a long enough one (the count skips the short ones too)."""
    genome = synth.make_genome(30_000, seed=61)
    short = synth.make_fastq(genome, 500, 20, seed=62, device="cpu").numpy().tobytes()
    longer = synth.make_fastq(genome, 800, 90, seed=63, device="cpu", first_index=7000).numpy().tobytes()
    a, b = tmp_path / "a.fq", tmp_path / "b.fq"
    a.write_bytes(short)
    b.write_bytes(longer)
    engine.sketch_files([a, b], 27, 1000, tmp_path / "o.msh", reads=True, min_mult=1)
    ref = mo.Sketcher(27, 1000, 1)
    ref.add_fastx(short)
    ref.add_fastx(longer)
    got = mo.read_msh(tmp_path / "o.msh").references[0]
    assert got.comment == ref.comment() == "[800 seqs] r00007000  [...]"
    assert np.array_equal(got.hashes, ref.finish()[0])


def test_thousands_of_tiny_pushes_with_multiplicity_filter():
    """One read per push (300-byte spans, each its own launch): the stage accounting of the capped phase
    must follow the real bytes, not whole tiles."""
    import torch

    genome = synth.make_genome(20_000, seed=71)
    fq = synth.make_fastq(genome, 4000, 150, seed=72, device="cuda")
    torch.cuda.synchronize()
    rb = synth.record_bytes(150)
    sk = engine.Sketcher(21, 2000, 3, expected_bytes=0)
    for i in range(4000):
        sk.push_device(fq.data_ptr() + i * rb, rb, engine.FMT_FASTQ4)
    got, _ = sk.finish()
    sk.close()
    ref = mo.Sketcher(21, 2000, 3)
    ref.add_fastx(fq.cpu().numpy().tobytes())
    assert np.array_equal(got, ref.finish()[0])


@pytest.mark.parametrize("seed", _seeds(2))
def test_randomised_ragged_multi_chunk_gz(tmp_path, seed):
    """A ~70 MB .fq.gz of ragged reads (lengths 1..400, qualities that begin with '@' or '+', N's, some CRLF
    files): several 32 MiB chunks cut at record boundaries by newline counting, DEFLATE matches reaching
    back across the chunk borders, against the oracle."""
    import gzip

    rng = np.random.default_rng(8600 + seed)
    k = int(rng.choice([16, 21, 31]))
    s = int(rng.choice([1000, 20000]))
    m = int(rng.choice([1, 2]))
    genome = synth.make_genome(int(rng.integers(50_000, 400_000)), seed=400 + seed).tobytes()
    nl = b"\r\n" if rng.random() < 0.3 else b"\n"
    quals = np.frombuffer(b"@+!#IJ5<?ACGT", np.uint8)
    n_reads = 180_000
    lens = rng.integers(1, 401, n_reads)
    starts = rng.integers(0, len(genome) - 401, n_reads)
    qual_pool = bytes(rng.choice(quals, size=4096))
    parts = []
    for i in range(n_reads):
        L = int(lens[i])
        seq = genome[int(starts[i]):int(starts[i]) + L]
        if i % 97 == 0 and L > 3:
            seq = seq[:L // 2] + b"N" + seq[L // 2 + 1:]
        q0 = int(starts[i]) % (4096 - 401)
        parts.append(b"@r%d/1" % i + nl + seq + nl + b"+" + nl + qual_pool[q0:q0 + L] + nl)
    data = b"".join(parts)
    assert len(data) > (32 << 20) + (32 << 20)
    p = tmp_path / "ragged.fq.gz"
    with gzip.open(p, "wb", compresslevel=int(rng.integers(1, 7))) as fh:
        fh.write(data)
    engine.sketch_files([p], k, s, tmp_path / "o.msh", reads=True, min_mult=m)
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(data)
    want, _ = ref.finish()
    got = mo.read_msh(tmp_path / "o.msh").references[0]
    assert np.array_equal(got.hashes, want), (k, s, m)
    assert got.comment == ref.comment()


def test_multi_member_gz_and_wrong_crc_through_the_ingest(tmp_path):
    """The ingest checks every gzip member's CRC-32 on a helper thread that runs behind the decoder: a file of
    three concatenated members (bgzip style) sketches like their concatenation, and a member with a wrong CRC
    is refused (after zlib has had the last word)."""
    import gzip

    genome = synth.make_genome(60_000, seed=81)
    parts = [synth.make_fastq(genome, n, 120, seed=82 + i, device="cpu", first_index=i * 100_000).numpy().tobytes()
             for i, n in enumerate((30_000, 1, 45_000))]
    p = tmp_path / "members.fq.gz"
    p.write_bytes(b"".join(gzip.compress(x, compresslevel=5) for x in parts))
    engine.sketch_files([p], 21, 2000, tmp_path / "o.msh", reads=True, min_mult=2)
    ref = mo.Sketcher(21, 2000, 2)
    ref.add_fastx(b"".join(parts))
    got = mo.read_msh(tmp_path / "o.msh").references[0]
    assert np.array_equal(got.hashes, ref.finish()[0])
    assert got.comment == ref.comment()
    bad = bytearray(b"".join(gzip.compress(x, compresslevel=5) for x in parts))
    first_len = len(gzip.compress(parts[0], compresslevel=5))
    bad[first_len - 8] ^= 0xFF                      # CRC field of the first member
    q = tmp_path / "badcrc.fq.gz"
    q.write_bytes(bytes(bad))
    with pytest.raises(engine.EngineError):
        engine.sketch_files([q], 21, 2000, tmp_path / "x.msh", reads=True, min_mult=2)


def _long_read_fastq(rng, genome: bytes, n, lo, hi):
    reads, rec = [], []
    for i in range(n):
        L = int(rng.integers(lo, hi + 1))
        o = int(rng.integers(0, len(genome) - L))
        r = genome[o:o + L]
        reads.append(r)
        q = bytes(rng.choice(np.frombuffer(b"@+5I#", np.uint8), size=L))
        rec.append(b"@ont%d ch=%d\n" % (i, i % 512) + r + b"\n+\n" + q + b"\n")
    return reads, b"".join(rec)


@pytest.mark.parametrize("lo,hi,m", [(3_000, 60_000, 1), (100, 9_000, 2), (2_000, 3_500, 1)])
def test_long_read_fastq_takes_the_look_back_repair_pass(lo, hi, m):
    """Reads beyond ~2.7 kb: a 32 KiB tile sees fewer than six line starts and cannot find its line phase by itself;
    such tiles are left out of the first pass and sketched by the look-back pass that settle()/finish() start.  Pushes
    from device memory (finish() repairs) and through the host staging buffer (the next push repairs)."""
    import torch

    rng = np.random.default_rng(lo + hi)
    genome = synth.make_genome(400_000, seed=31).tobytes()
    reads, data = _long_read_fastq(rng, genome, 400, lo, hi)
    ref, (want, want_c) = oracle_sketch(data, 21, 2000, m)
    dev = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    sk = engine.Sketcher(21, 2000, m, expected_bytes=2 * len(data))
    sk.push_device(dev.data_ptr(), len(data), engine.FMT_FASTQ4)
    got, cnt = sk.finish()
    st = sk.stats()
    assert np.array_equal(got, want) and np.all(cnt >= want_c)   # exact multiplicities; mash's heap forgets evicted ones
    assert st["lines"] == 4 * len(reads) and st["flags"] == 0 and sk.record_count() == ref.records
    # two host pushes, halves cut at a record border: the first is repaired before the staging buffer is reused
    cut = data.index(b"\n@ont200 ") + 1
    sk.reset()
    sk.push_host(data[:cut], engine.FMT_FASTQ4)
    sk.push_host(data[cut:], engine.FMT_FASTQ4)
    again, cnt2 = sk.finish()
    st2 = sk.stats()
    sk.close()
    assert np.array_equal(again, want) and np.array_equal(cnt2, cnt)
    assert st2["lines"] == 4 * len(reads) and st2["kmers"] == st["kmers"]


def test_pushed_buffer_may_be_reused_after_sync_not_before(tmp_path):
    """include/mhx.h: a pushed span has to stay valid and unchanged until sync() / finish() has returned -- long-read FASTQ
    is read a second time by the repair pass those calls start.  After sync() the caller may overwrite the buffer: the
    sketch must be that of the original bytes."""
    import torch

    rng = np.random.default_rng(9)
    genome = synth.make_genome(200_000, seed=33).tobytes()
    reads, data = _long_read_fastq(rng, genome, 300, 2000, 30_000)   # every tile of these reads needs the look-back pass
    buf = torch.frombuffer(bytearray(data + b"\0" * 64), dtype=torch.uint8).cuda()
    sk = engine.Sketcher(21, 2000, 1, expected_bytes=len(data))
    sk.push_device(buf.data_ptr(), len(data), engine.FMT_FASTQ4, keep=buf)
    sk.sync()                     # the repair pass has run: the span is the caller's again
    buf.fill_(ord("N"))
    torch.cuda.synchronize()
    got, _ = sk.finish()
    sk.close()
    _, (want, _) = oracle_sketch(data, 21, 2000, 1)
    assert np.array_equal(got, want)


def test_long_read_fastq_gz_through_the_chunked_ingest(tmp_path, monkeypatch):
    """The streaming ingest reuses its device slots: the repair pass of a chunk has to run before the slot is refilled."""
    import gzip

    rng = np.random.default_rng(8)
    genome = synth.make_genome(300_000, seed=32).tobytes()
    reads, data = _long_read_fastq(rng, genome, 2500, 500, 40_000)   # ~100 MB: four 32 MiB chunks
    p = tmp_path / "ont.fastq.gz"
    p.write_bytes(gzip.compress(data, 1))
    engine.sketch_files([p], 27, 5000, tmp_path / "o.msh", reads=True, min_mult=1)
    got = mo.read_msh(tmp_path / "o.msh").references[0]
    ref, (want, _) = oracle_sketch(data, 27, 5000, 1)
    assert np.array_equal(got.hashes, want)
    assert got.comment == ref.comment()


def test_crlf_reads_one_base_shorter_than_k_are_no_records(tmp_path):
    """A CRLF FASTQ whose reads hold k - 1 bases: every sequence line has k bytes (the bases and the '\\r'), none has k
    bases.  mash stops with its "no records" error, and so must the file-level call (found by the randomised sweep)."""
    k = 32
    genome = synth.make_genome(20_000, seed=61)
    data = synth.make_fastq(genome, 500, k - 1, seed=62, device="cpu").numpy().tobytes().replace(b"\n", b"\r\n")
    ref = mo.Sketcher(k, 1000, 1)
    ref.add_fastx(data)
    assert ref.records == 0
    for name in ("a.fq", "a.fq.gz"):
        p = tmp_path / name
        if name.endswith(".gz"):
            import gzip
            p.write_bytes(gzip.compress(data))
        else:
            p.write_bytes(data)
        with pytest.raises(engine.NoRecordsError):
            engine.sketch_files([p], k, 1000, tmp_path / "o.msh", reads=True, min_mult=1)
    # one base more and the file is sketched
    ok = synth.make_fastq(genome, 500, k, seed=62, device="cpu").numpy().tobytes().replace(b"\n", b"\r\n")
    (tmp_path / "b.fq").write_bytes(ok)
    engine.sketch_files([tmp_path / "b.fq"], k, 1000, tmp_path / "b.msh", reads=True, min_mult=1)
    _, (want, _) = oracle_sketch(ok, k, 1000, 1)
    assert np.array_equal(mo.read_msh(tmp_path / "b.msh").references[0].hashes, want)


def test_concatenated_gzip_members_through_the_ingest(tmp_path, monkeypatch):
    """`cat a.fq.gz b.fq.gz > ab.fq.gz` is one file of two members (mash reads it through gzread like any other): each large
    member gets the decoding threads in turn, a small one in between goes through the sequential decoder, and a record
    may be cut by a member border."""
    import gzip

    monkeypatch.setenv("MHX_PINFLATE_MIN", "1000000")
    genome = synth.make_genome(300_000, seed=51)
    a = synth.make_fastq(genome, 150_000, 150, seed=52, device="cpu").numpy().tobytes()   # 47 MB
    cut = len(a) // 2 + 77                                   # inside a record
    tiny = b"@t1\n" + genome[:60].tobytes() + b"\n+\n" + b"I" * 60 + b"\n"
    p = tmp_path / "ab.fq.gz"
    p.write_bytes(gzip.compress(a[:cut], 1) + gzip.compress(a[cut:], 6) + gzip.compress(tiny, 9))
    engine.sketch_files([p], 21, 2000, tmp_path / "o.msh", reads=True, min_mult=2)
    got = mo.read_msh(tmp_path / "o.msh").references[0]
    ref, (want, _) = oracle_sketch(a + tiny, 21, 2000, 2)
    assert np.array_equal(got.hashes, want)
    assert got.comment == ref.comment()


def test_record_without_qualities_in_front_of_a_tile_border(tmp_path):
    """'@h / SEQ / @h2 ...': kseq reads a record without qualities.  When the cut record's sequence line straddles a tile
    border, every tile's own first lines look regular; the chain check over the tiles' phases refuses the 4-line path
    (MHX_E_FORMAT on a raw push) and the file-level call sketches the file with the general record parser."""
    rng = np.random.default_rng(77)
    reads = random_reads(rng, 600, 100, 150, p_n=0, p_lower=0)
    recs = [b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n" for i, r in enumerate(reads)]
    body, i = b"", 0
    while len(body) < 32768 - 900:
        body += recs[i]
        i += 1
    L = (32768 - 100 - len(body) - 7) // 2
    filler = b"C" * L
    body += b"@f\n" + filler + b"\n+\n" + b"I" * L + b"\n"
    cut_seq = b"ACGGTCA" * 34
    data = body + b"@cut\n" + cut_seq + b"\n" + b"".join(recs[i:])
    assert len(body) + 5 < 32768 < len(body) + 5 + len(cut_seq)
    import torch
    dev = torch.frombuffer(bytearray(data + b"\0" * 64), dtype=torch.uint8).cuda()   # 16-byte aligned: tile borders as computed
    sk = engine.Sketcher(21, 1000, 1, expected_bytes=len(data))
    sk.push_device(dev.data_ptr(), len(data), engine.FMT_FASTQ4)
    with pytest.raises(engine.EngineError) as e:
        sk.finish()
    assert e.value.code == engine.MHX_E_FORMAT
    sk.close()
    p = tmp_path / "cut.fq"
    p.write_bytes(data)
    engine.sketch_files([p], 21, 1000, tmp_path / "c.msh", reads=True, min_mult=1)
    got = mo.read_msh(tmp_path / "c.msh").references[0].hashes
    want, _ = mo.bruteforce_sketch(reads[:i] + [filler, cut_seq] + reads[i:], 21, 1000, 1)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("s,m", [(8192, 1), (120_000, 1), (300_000, 2)])
def test_large_sketches_come_back_in_hash_order_from_the_device(s, m):
    """s >= 8192: finish() has the device order the result block (counting sort on the leading bits, ranks inside the
    buckets) and store it into the pinned block; the host only checks the order.  Sizes with 2^14, 2^17 and 2^19 buckets,
    a second finish() on the same table (the bucket counters must be back at zero), and a sketch shorter than s."""
    genome = synth.make_genome(1_500_000, seed=41)
    fq = synth.make_fastq(genome, 120_000, 150, seed=42, device="cpu").numpy()
    sk = engine.Sketcher(21, s, m, expected_bytes=fq.size)
    sk.push_host(fq, engine.FMT_FASTQ4)
    got, cnt = sk.finish()
    again, cnt2 = sk.finish()
    sk.close()
    _, (want, _) = oracle_sketch(fq.tobytes(), 21, s, m)
    assert len(got) == min(s, len(want)) and np.array_equal(got, want)
    assert np.array_equal(again, got) and np.array_equal(cnt2, cnt)
    assert np.all(cnt >= m)


@pytest.mark.parametrize("seed", _seeds(6))
def test_randomised_mixtures_of_short_and_long_reads(seed):
    """Streams in which stretches of short reads (tiles that find their line phase by themselves) alternate with
    stretches of reads of up to 12 kb (tiles left to the look-back repair pass), pushed in one to four spans from device
    memory or through the host staging buffer, LF or CRLF, quality lines that begin with '@' or '+'."""
    import torch

    rng = np.random.default_rng(9900 + seed)
    k = int(rng.choice([11, 16, 21, 27, 31, 32]))
    m = int(rng.choice([1, 1, 2, 3]))
    s = int(rng.choice([100, 1000, 10000]))
    genome = synth.make_genome(int(rng.integers(20_000, 150_000)), seed=300 + seed).tobytes()
    nl = b"\r\n" if rng.random() < 0.25 else b"\n"
    reads, recs = [], []
    for stretch in range(int(rng.integers(2, 7))):
        long_ones = bool(rng.integers(0, 2))
        for _ in range(int(rng.integers(5, 40)) if long_ones else int(rng.integers(50, 1500))):
            L = int(rng.integers(2_500, 12_000)) if long_ones else int(rng.integers(1, 400))
            L = min(L, len(genome) - 1)
            p0 = int(rng.integers(0, len(genome) - L))
            r = genome[p0:p0 + L]
            q = bytearray(rng.choice(np.frombuffer(b"@+I5#", np.uint8), size=L).tobytes())
            reads.append(r)
            recs.append(b"@x%d" % len(recs) + nl + r + nl + b"+" + nl + bytes(q) + nl)
    n_push = int(rng.integers(1, 5))
    cuts = sorted(set(int(x) for x in rng.integers(0, len(recs) + 1, n_push - 1))) if n_push > 1 else []
    bounds = [0] + cuts + [len(recs)]
    blobs = [b"".join(recs[a:b]) for a, b in zip(bounds[:-1], bounds[1:])]
    blobs = [b for b in blobs if b]
    via_host = bool(rng.integers(0, 2))
    devs = [] if via_host else [torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() for b in blobs]
    torch.cuda.synchronize()
    got = None
    for scale in (1, 16, 256, 4096):
        sk = engine.Sketcher(k, s, m, expected_bytes=sum(len(b) for b in blobs), budget_scale=scale)
        for i, b in enumerate(blobs):
            if via_host:
                sk.push_host(b, engine.FMT_FASTQ4)
            else:
                sk.push_device(devs[i].data_ptr(), len(b), engine.FMT_FASTQ4)
        try:
            got, _ = sk.finish()
        except engine.EngineError as e:
            sk.close()
            if e.code != engine.MHX_E_CAPACITY:
                raise
            continue
        st = sk.stats()
        n_long = sk.record_count()
        sk.close()
        break
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(b"".join(blobs))
    want, _ = ref.finish()
    assert got is not None and np.array_equal(got, want), (k, s, m)
    assert st["lines"] == 4 * len(reads) and st["flags"] == 0
    assert n_long == ref.records == sum(1 for r in reads if len(r) >= k)


def test_bgzf_fastq_through_the_ingest(tmp_path):
    """A bgzip'ed FASTQ (independent 64 KiB members): the ingest decodes its blocks side by side; mash reads the same file
    through zlib as one multi-member stream.  Followed by an ordinary gzip member, as concatenated files are."""
    import gzip

    from tests.test_lib_cpu import _bgzf

    genome = synth.make_genome(200_000, seed=51)
    a = synth.make_fastq(genome, 120_000, 150, seed=52, device="cpu").numpy().tobytes()      # 38 MB: ~580 blocks, two groups
    b = synth.make_fastq(genome, 5_000, 100, seed=53, device="cpu", first_index=500_000).numpy().tobytes()
    p = tmp_path / "reads.fastq.gz"
    p.write_bytes(_bgzf(a, 4) + gzip.compress(b, 6))
    engine.sketch_files([p], 21, 2000, tmp_path / "o.msh", reads=True, min_mult=2)
    got = mo.read_msh(tmp_path / "o.msh").references[0]
    ref, (want, _) = oracle_sketch(a + b, 21, 2000, 2)
    assert np.array_equal(got.hashes, want)
    assert got.comment == ref.comment()


@pytest.mark.parametrize("nq", [1, 200])
def test_dist_references_of_one_clade_stay_on_the_fast_path(nq):
    """AuriClass's reference set is 24 genomes of five clades: the references share most of their hashes, so the slices
    of one value range, summed over the references, exceed the LDS table many times over while its distinct keys are few.
    The range pass decides by the keys: no block may fall back to the generic kernel (it did for every such set while the
    sum was the test: 0.48 instead of 0.05 ms for the 1 x 24 comparison AuriClass makes, 62 ms for 1024 queries)."""
    rng = np.random.default_rng(77)
    s = 50000   # AuriClass's sketch size: ~49 hashes per list and value range, 24 x 49 (and any fluctuation, 24-fold) against 1536 slots
    base = _sketch_like(rng, s)
    refs = []
    for j in range(24):
        keep = rng.random(len(base)) >= 0.0005 * (j + 1)          # 99.9 .. 98.8 % shared
        refs.append(np.unique(np.concatenate([base[keep], _sketch_like(rng, int((~keep).sum()))])))
    qrys = []
    for i in range(nq):
        src = refs[i % 24]
        keep = rng.random(len(src)) >= 0.3 * i / max(1, nq - 1)
        qrys.append(np.unique(np.concatenate([src[keep], _sketch_like(rng, int((~keep).sum()))])))
    stride = (max(max(map(len, refs)), max(map(len, qrys))) + 15) // 16 * 16
    Q, ql = _pad_rows(qrys, stride)
    R, rl = _pad_rows(refs, stride)
    common, denom, dist = engine.dist_batch(Q, ql, R, rl, 27, s)
    assert engine.load().mhx_last_dist_fallback_blocks() == 0
    for qi in range(0, nq, max(1, nq // 7)):
        for ri, r in enumerate(refs):
            c, d, dd = mo.compare(r, qrys[qi], s, 27)
            assert (common[qi, ri], denom[qi, ri]) == (c, d), (qi, ri)
            assert dist[qi, ri] == dd


def test_fastq4_with_empty_reads_and_illumina_style_headers():
    """Reads trimmed to nothing (`@h / (empty) / + / (empty)`, which kseq reads as a record without bases) among short and
    long ones, headers in Illumina's style (they hold `:`, blanks, `+` and `@`), qualities that begin with `@` or `+`:
    the tile-local line-phase search and the per-record layout check must neither trip over the empty lines nor take a
    quality line for a header."""
    import torch

    rng = np.random.default_rng(314)
    lens = rng.choice([0, 0, 1, 20, 21, 22, 75, 151], size=6000)
    reads = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(L))) for L in lens]
    quals = np.frombuffer(b"@+FF:,#", np.uint8)
    rec = []
    for i, r in enumerate(reads):
        q = bytes(rng.choice(quals, size=len(r)))
        hdr = b"A00%d:45:HXX+DSXX:1:1101:%d:1000 1:N:0:ACGT+TG@A" % (i % 7, 1000 + i)
        rec.append(b"@" + hdr + b"\n" + r + b"\n+" + (hdr if i % 3 == 0 else b"") + b"\n" + q + b"\n")   # old style: the `+` line repeats the name
    data = b"".join(rec)
    dev = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    for k, s, m in ((21, 1000, 1), (21, 500, 2)):
        sk = engine.Sketcher(k, s, m, expected_bytes=len(data))
        sk.push_device(dev.data_ptr(), len(data), engine.FMT_FASTQ4)
        got, cnt = sk.finish()
        st = sk.stats()
        n_rec = sk.record_count()
        sk.close()
        want, wc = mo.bruteforce_sketch(reads, k, s, m)
        assert st["flags"] == 0
        assert np.array_equal(got, want) and np.array_equal(cnt, wc)
        assert n_rec == sum(1 for r in reads if len(r) >= k)


@pytest.mark.parametrize("s", [1000, 50000])
def test_dist_files_on_a_large_reference_file_and_a_damaged_one(tmp_path, s):
    """`mash dist REF QUERY` at file level with a reference sketch file of 24 clade-like references: at s = 50 000 (9.6 MB)
    the file goes through the pinned image / in-place parse / row-by-row copy path, at s = 1000 through the heap; the text
    must be the oracle's.  The same file with two neighbouring hashes of one reference swapped (an order the merge
    kernels cannot work on) must be refused, by the parallel order check of the large file and by the small one's."""
    rng = np.random.default_rng(5 + s)
    base = _sketch_like(rng, s)
    refs = []
    for j in range(24):
        keep = rng.random(len(base)) >= 0.001 * (j + 1)
        h = np.unique(np.concatenate([base[keep], _sketch_like(rng, int((~keep).sum()))]))
        refs.append(mo.Reference("ref%d.fa" % j, "clade %d" % (j % 5), 12_000_000 + j, h))
    q = np.unique(np.concatenate([refs[7].hashes[::2], _sketch_like(rng, s // 2)]))[:s]
    R = mo.SketchFile(kmer_size=27, sketch_size=s, references=refs)
    Q = mo.SketchFile(kmer_size=27, sketch_size=s, references=[mo.Reference("sample.fa", "query", 12_300_000, q)])
    (tmp_path / "r.msh").write_bytes(mo.msh_bytes(R))
    (tmp_path / "q.msh").write_bytes(mo.msh_bytes(Q))
    for _ in range(2):   # the second call reuses the pinned image
        assert engine.dist_files(tmp_path / "r.msh", tmp_path / "q.msh") == mo.dist_text(R, Q)
    assert engine.load().mhx_last_dist_fallback_blocks() <= 0   # fast path without a fallback (s = 50 000) or the tiny-batch kernel
    bad = refs[13].hashes.copy()
    bad[[len(bad) // 2, len(bad) // 2 + 1]] = bad[[len(bad) // 2 + 1, len(bad) // 2]]
    damaged = list(refs)   # (R holds `refs` itself)
    damaged[13] = mo.Reference(refs[13].name, refs[13].comment, refs[13].length, bad)
    (tmp_path / "bad.msh").write_bytes(mo.msh_bytes(mo.SketchFile(kmer_size=27, sketch_size=s, references=damaged)))
    with pytest.raises(engine.EngineError, match="not ascending"):
        engine.dist_files(tmp_path / "bad.msh", tmp_path / "q.msh")
    assert engine.dist_files(tmp_path / "r.msh", tmp_path / "q.msh") == mo.dist_text(R, Q)   # and the engine is fine afterwards


def test_dist_reference_sets_beyond_one_group_of_blocks():
    """One query against 135 000 tiny reference sketches: 4219 slices of 32 references -- more than the 4096 blocks whose
    flag words come back in one copy (such a call used to be refused)."""
    rng = np.random.default_rng(99)
    s, nr = 12, 135_000
    pool = _sketch_like(rng, 400)
    R = np.zeros((nr, 16), np.uint64)
    rl = rng.integers(0, s + 1, size=nr).astype(np.uint32)
    for i in range(nr):
        R[i, :rl[i]] = np.sort(rng.choice(pool, size=int(rl[i]), replace=False))
    q = np.sort(rng.choice(pool, size=s, replace=False))
    Q = np.zeros((1, 16), np.uint64)
    Q[0, :s] = q
    common, denom, dist = engine.dist_batch(Q, np.array([s], np.uint32), R, rl, 21, s)
    for i in list(range(0, nr, 997)) + [nr - 1, 131_071, 131_072, 131_073]:
        c, d, dd = mo.compare(R[i, :rl[i]], q, s, 21)
        assert (common[0, i], denom[0, i]) == (c, d), i
        assert dist[0, i] == dd


@pytest.mark.parametrize("seed", _seeds(4))
def test_randomised_concatenated_gzip_members(tmp_path, monkeypatch, seed):
    """`cat a.gz b.gz ...`: 1 .. 6 members of a few records to a few MB each, any compression level, cut at arbitrary
    record boundaries or in the middle of a record; through the streaming ingest and (every other seed) through the
    whole-file reader that takes over when the ingest declines; the member that announces its length last is often not
    the longest (the whole-file reader sizes its buffer from it)."""
    import gzip

    rng = np.random.default_rng(9900 + seed)
    if seed % 2:
        monkeypatch.setenv("MHX_NO_STREAMING", "1")
    k, s, m = int(rng.choice([21, 27])), int(rng.choice([500, 5000])), int(rng.choice([1, 2]))
    genome = synth.make_genome(int(rng.integers(40_000, 200_000)), seed=500 + seed)
    n_members = int(rng.integers(1, 7))
    reads = [int(rng.choice([1, 3, 50, 4000, 20_000, 60_000])) for _ in range(n_members)]
    data = b"".join(synth.make_fastq(genome, n, 100, seed=1000 * seed + i, device="cpu", first_index=i * 100_000).numpy().tobytes() for i, n in enumerate(reads))
    cuts = sorted(set([0, len(data)] + [int(rng.integers(0, len(data))) for _ in range(n_members - 1)]))   # members need not end with a record
    p = tmp_path / "cat.fq.gz"
    p.write_bytes(b"".join(gzip.compress(data[a:b], compresslevel=int(rng.integers(1, 10))) for a, b in zip(cuts, cuts[1:])))
    engine.sketch_files([p], k, s, tmp_path / "o.msh", reads=True, min_mult=m)
    ref = mo.Sketcher(k, s, m)
    ref.add_fastx(data)
    got = mo.read_msh(tmp_path / "o.msh").references[0]
    assert np.array_equal(got.hashes, ref.finish()[0]), (seed, reads, cuts)
    assert got.comment == ref.comment()


@pytest.mark.parametrize("which", ["first", "second", "both", "truncated", "missing"])
def test_one_bad_file_of_a_pair_ends_the_call_not_the_process(tmp_path, which):
    """Two `.fq.gz` files decoded side by side (AuriClass's paired reads), each several ingest chunks long: when one of
    them is damaged in the middle, cut short, or not there at all, the call must come back with an error -- the other
    file's producer threads, the chunk queue and the device slots wound down -- and the engine must serve the next call."""
    import gzip

    genome = synth.make_genome(80_000, seed=61)
    data = [synth.make_fastq(genome, 130_000, 150, seed=62 + i, device="cpu", first_index=i * 10**6).numpy().tobytes() for i in range(2)]
    z = [bytearray(gzip.compress(d, compresslevel=3)) for d in data]
    paths = [tmp_path / "r1.fq.gz", tmp_path / "r2.fq.gz"]
    for i in (0, 1):
        if which in (("first", "second")[i], "both"):
            z[i][len(z[i]) // 2] ^= 0x41
            z[i][len(z[i]) // 2 + 1000] ^= 0x17
    if which == "truncated":
        z[1] = z[1][: len(z[1]) * 2 // 3]
    for pth, zz in zip(paths, z):
        pth.write_bytes(bytes(zz))
    if which == "missing":
        paths[1].unlink()
    if which == "truncated":
        # a file cut short is not an error to mash: zlib hands kseq every byte it can decode, the read that fails ends
        # the stream like an end of file, a last record without its qualities is dropped -- the sketch of the prefix
        import zlib

        prefix = zlib.decompressobj(wbits=31).decompress(bytes(z[1]))
        assert 0 < len(prefix) < len(data[1])
        engine.sketch_files(paths, 27, 5000, tmp_path / "x.msh", reads=True, min_mult=2)
        ref = mo.Sketcher(27, 5000, 2)
        ref.add_fastx(data[0])
        ref.add_fastx(prefix)
        assert np.array_equal(mo.read_msh(tmp_path / "x.msh").references[0].hashes, ref.finish()[0])
    else:
        with pytest.raises(engine.EngineError):
            engine.sketch_files(paths, 27, 5000, tmp_path / "x.msh", reads=True, min_mult=2)
    good = tmp_path / "good.fq.gz"
    good.write_bytes(gzip.compress(data[0][: 3_000_000 // 315 * 315], compresslevel=3))
    engine.sketch_files([good], 27, 5000, tmp_path / "y.msh", reads=True, min_mult=1)
    ref = mo.Sketcher(27, 5000, 1)
    ref.add_fastx(data[0][: 3_000_000 // 315 * 315])
    assert np.array_equal(mo.read_msh(tmp_path / "y.msh").references[0].hashes, ref.finish()[0])


@pytest.mark.parametrize("gz", [False, True])
def test_last_record_without_its_quality_string_is_reported(tmp_path, gz):
    """A FASTQ cut short inside its last record: with the `+` line but no qualities, or with fewer qualities than bases,
    kseq_read answers -2 and the oracle raises -- the engine must not sketch that record's bases (its device parser checks
    the four-line layout, not the quality lengths; the tail of every file is looked at on the host).  Cut inside the
    header or the sequence line the record is one without qualities, which kseq returns: sketched, like the oracle does."""
    import gzip

    genome = synth.make_genome(50_000, seed=71)
    body = synth.make_fastq(genome, 3000, 150, seed=72, device="cpu").numpy().tobytes()
    last_seq = bytes(genome[100:250].tobytes())
    cases = {
        "plus_only": (b"@last\n" + last_seq + b"\n+\n", True),
        "plus_no_newline": (b"@last\n" + last_seq + b"\n+", True),
        "short_quality": (b"@last\n" + last_seq + b"\n+\n" + b"I" * 100 + b"\n", True),
        "short_quality_no_newline": (b"@last\n" + last_seq + b"\n+\n" + b"I" * 149, True),
        "long_quality": (b"@last\n" + last_seq + b"\n+\n" + b"I" * 151 + b"\n", True),
        "cut_in_sequence": (b"@last\n" + last_seq[:77], False),
        "cut_after_sequence": (b"@last\n" + last_seq + b"\n", False),
        "header_only": (b"@last\n", False),
        "complete_no_newline": (b"@last\n" + last_seq + b"\n+\n" + b"I" * 150, False),
        "complete_crlf": (b"@last\r\n" + last_seq + b"\r\n+\r\n" + b"I" * 150 + b"\r\n", False),
    }
    for name, (tail, bad) in cases.items():
        for lead in (body, b""):
            data = lead + tail
            p = tmp_path / (name + (".fq.gz" if gz else ".fq"))
            p.write_bytes(gzip.compress(data, 3) if gz else data)
            ref = mo.Sketcher(21, 1000, 1)
            try:
                ref.add_fastx(data)
                oracle_raises = False
            except ValueError:
                oracle_raises = True
            assert oracle_raises == bad, name
            if bad:
                with pytest.raises(engine.EngineError):
                    engine.sketch_files([p], 21, 1000, tmp_path / "o.msh", reads=True, min_mult=1)
                    print("NOT REFUSED:", name, len(lead), "gz" if gz else "plain")
                continue
            want = ref.finish()[0]
            if len(want) == 0:   # nothing of k bases: "Did not find fasta records"
                with pytest.raises(engine.EngineError):
                    engine.sketch_files([p], 21, 1000, tmp_path / "o.msh", reads=True, min_mult=1)
                continue
            engine.sketch_files([p], 21, 1000, tmp_path / "o.msh", reads=True, min_mult=1)
            assert np.array_equal(mo.read_msh(tmp_path / "o.msh").references[0].hashes, want), (name, len(lead))


def test_damaged_sketch_containers_end_in_an_error_or_a_table_never_in_a_crash(tmp_path):
    """`mash dist` on `.msh` files cut short or with bytes flipped (header and pointer area, anywhere): the reference's own
    fixture (the copying reader) and a 9.6 MB container (the pinned image, parsed in place).  Every call must come back --
    with an error, or with a table when the damage left the structure valid (flipped hash or name bytes) -- and the engine
    must answer the undamaged question afterwards."""
    rng = np.random.default_rng(12)
    small = (REFDATA / "ref_sketch.msh").read_bytes()
    refs = [mo.Reference("r%d" % j, "c", 10 ** 7, _sketch_like(rng, 40000)) for j in range(30)]
    big = mo.msh_bytes(mo.SketchFile(kmer_size=27, sketch_size=40000, references=refs))
    (tmp_path / "q.msh").write_bytes(small)
    (tmp_path / "qb.msh").write_bytes(mo.msh_bytes(mo.SketchFile(kmer_size=27, sketch_size=40000, references=refs[:1])))
    (tmp_path / "big.msh").write_bytes(big)
    good = {"small": engine.dist_files(tmp_path / "q.msh", tmp_path / "q.msh"), "big": engine.dist_files(tmp_path / "big.msh", tmp_path / "qb.msh")}
    outcomes = {"error": 0, "table": 0}
    for name, blob, query in (("small", small, "q.msh"), ("big", big, "qb.msh")):
        for trial in range(45 * FUZZ):
            b = bytearray(blob)
            if trial % 3 == 0:
                b = b[: int(rng.integers(0, len(b)))]
            else:
                span = min(len(b), 4096) if trial % 3 == 1 else len(b)
                for _ in range(int(rng.integers(1, 4))):
                    b[int(rng.integers(0, span))] ^= int(rng.integers(1, 256))
            (tmp_path / "m.msh").write_bytes(bytes(b))
            try:
                engine.dist_files(tmp_path / "m.msh", tmp_path / query)
                outcomes["table"] += 1
            except engine.EngineError:
                outcomes["error"] += 1
    assert outcomes["error"] > 0
    assert engine.dist_files(tmp_path / "q.msh", tmp_path / "q.msh") == good["small"]
    assert engine.dist_files(tmp_path / "big.msh", tmp_path / "qb.msh") == good["big"]
