"""Pins the CPU oracle (oracle/) against every golden vector the reference's own tests
hold for the sketch/dist/bounds path."""
import numpy as np
import pytest

from oracle import mash_oracle as mo
from tests.conftest import GOLDEN, REFDATA


@pytest.fixture(scope="module")
def ref_sketch():
    return mo.read_msh(REFDATA / "ref_sketch.msh")


def test_murmur3_known_answers():
    # Appleby's MurmurHash3_x64_128 reference vectors (smhasher verification style):
    out = (mo.ctypes.c_uint64 * 2)()
    mo.lib().mo_murmur3_x64_128(b"", 0, 0, out)
    assert (out[0], out[1]) == (0, 0)
    mo.lib().mo_murmur3_x64_128(b"hello", 5, 0, out)
    assert out[0] == 0xCBD8A7B341BD9B02 and out[1] == 0x5B1E906A48AE1D19
    mo.lib().mo_murmur3_x64_128(b"The quick brown fox jumps over the lazy dog", 43, 0, out)
    assert out[0] == 0xE34BBC7BBC071B6C and out[1] == 0x7A433CA9C49A9347


def test_read_reference_msh(ref_sketch):
    assert ref_sketch.kmer_size == 27 and ref_sketch.sketch_size == 50000
    assert ref_sketch.hash_seed == 42 and ref_sketch.concatenated and not ref_sketch.noncanonical
    assert [r.name for r in ref_sketch.references] == ["tests/data/NC_001416.1.fasta", "tests/data/NC_001604.1.fasta"]
    assert [r.length for r in ref_sketch.references] == [48502, 39937]
    assert [len(r.hashes) for r in ref_sketch.references] == [48476, 39770]
    assert ref_sketch.references[0].comment == "NC_001416.1 Enterobacteria phage lambda, complete genome"
    for r in ref_sketch.references:
        assert np.all(np.diff(r.hashes.astype(object)) > 0)


def test_K1_sketch_hashes_match_reference_msh(ref_sketch):
    for i, fn in enumerate(["NC_001416.1.fasta.gz", "NC_001604.1.fasta.gz"]):
        sk, _ = mo.sketch_files([REFDATA / fn], 27, 50000)
        assert np.array_equal(sk.references[0].hashes, ref_sketch.references[i].hashes)
        assert sk.references[0].length == ref_sketch.references[i].length
        assert sk.references[0].comment == ref_sketch.references[i].comment


def test_K2_msh_bytes_identical(ref_sketch, tmp_path):
    sk, _ = mo.sketch_files([REFDATA / "NC_001416.1.fasta.gz", REFDATA / "NC_001604.1.fasta.gz"], 27, 50000)
    sk.references[0].name = "tests/data/NC_001416.1.fasta"
    sk.references[1].name = "tests/data/NC_001604.1.fasta"
    assert mo.msh_bytes(sk) == (REFDATA / "ref_sketch.msh").read_bytes()
    # and a pure re-serialisation of the parsed file
    assert mo.msh_bytes(ref_sketch) == (REFDATA / "ref_sketch.msh").read_bytes()


def test_K3_K4_fastq_m3(ref_sketch, refcwd, golden):
    sk, stderr = mo.sketch_files(["tests/data/NC_001416.1_1.fq.gz", "tests/data/NC_001416.1_2.fq.gz"],
                                 27, 50000, reads=True, m=3)
    assert "Estimated genome size: 48454.7\n" in stderr          # K4 (test_correct_workflow.py:99)
    text = mo.dist_text(ref_sketch, sk)
    want = golden["mash_output_to_dict_fastq"]
    rows = [l.split("\t") for l in text.splitlines()]
    assert [r[0] for r in rows] == [want["Reference"]["0"], want["Reference"]["1"]]
    assert [r[1] for r in rows] == [want["Query"]["0"], want["Query"]["1"]]
    assert [r[2] for r in rows] == ["9.55405e-06", "1"]
    assert [r[3] for r in rows] == ["0", "1"]
    assert [r[4] for r in rows] == ["48451/48476", "0/50000"]
    # round trip through the container
    mo.write_msh("q.msh", sk)
    assert mo.dist_text(ref_sketch, mo.read_msh("q.msh")) == text


def test_K5_fasta(ref_sketch, refcwd):
    sk, _ = mo.sketch_files(["tests/data/NC_001416.1.fasta.gz"], 27, 50000)
    text = mo.dist_text(ref_sketch, sk)
    assert text == ("tests/data/NC_001416.1.fasta\ttests/data/NC_001416.1.fasta.gz\t0\t0\t48476/48476\n"
                    "tests/data/NC_001604.1.fasta\ttests/data/NC_001416.1.fasta.gz\t1\t1\t0/50000\n")
    assert sk.references[0].length == 48502                      # K9


def test_K6_bounds_text():
    assert mo.bounds_text(27, 0.99) == (GOLDEN / "mash_bounds_k27_p0.99.txt").read_text()


def test_K8_empty_input(refcwd):
    with pytest.raises(mo.NoRecordsError, match="ERROR: Did not find fasta records in"):
        mo.sketch_files(["tests/data/test_empty_1.fq.gz", "tests/data/test_empty_2.fq.gz"], 27, 50000, reads=True, m=3)


@pytest.mark.parametrize("k,s,m", [(21, 1000, 1), (21, 1000, 3), (27, 500, 2), (16, 300, 1), (11, 200, 2), (32, 100, 1)])
def test_heap_restatement_equals_bruteforce_definition(k, s, m):
    """Unpinned-by-reference cases (truncation, k<=16, N/lower-case): the MinHashHeap
    restatement must equal the definition-level brute force."""
    rng = np.random.default_rng(k * 1000 + s + m)
    genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=6000)
    seqs = []
    for _ in range(400):
        st = int(rng.integers(0, len(genome) - 150))
        r = genome[st:st + int(rng.integers(10, 150))].copy()
        if rng.random() < 0.3:
            r[int(rng.integers(0, len(r)))] = ord("N")
        if rng.random() < 0.3:
            r = np.frombuffer(bytes(r).lower(), np.uint8).copy()
        if rng.random() < 0.5:
            comp = {65: 84, 67: 71, 71: 67, 84: 65, 78: 78, 97: 116, 99: 103, 103: 99, 116: 97, 110: 110}
            r = np.array([comp[int(c)] for c in r[::-1]], np.uint8)
        seqs.append(bytes(r))
    sk = mo.Sketcher(k, s, m)
    for q in seqs:
        sk.add_seq(q)
    h, c = sk.finish()
    bh, bc = mo.bruteforce_sketch(seqs, k, s, m)
    assert np.array_equal(h, bh)
    assert len(h) > 0
    if k <= 16:
        assert h.max() < 2 ** 32
