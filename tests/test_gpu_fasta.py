"""FASTA parsed on the device (auriclass_amd/csrc/mhx_fasta.hip): header lines dropped, line breaks squeezed out inside a
record so that k-mers span them, one separator in front of every record.  Every case is sketched three ways -- device
parser, host record parser (MHX_HOST_FASTA=1, the flagged fallback) and the oracle -- and the .msh files must be the same
bytes (`mash sketch -o OUT -k K -s S files`, /root/reference/auriclass/classes.py:696-713)."""
import os

import numpy as np
import pytest

from auriclass_amd import engine, synth
from oracle import mash_oracle as mo

pytestmark = pytest.mark.gpu


def _three_ways(tmp_path, paths, k, s, monkeypatch, bytes_vs_oracle=True):
    engine.sketch_files(paths, k, s, tmp_path / "dev.msh")
    monkeypatch.setenv("MHX_HOST_FASTA", "1")
    engine.sketch_files(paths, k, s, tmp_path / "host.msh")
    monkeypatch.delenv("MHX_HOST_FASTA")
    osk, _ = mo.sketch_files(paths, k, s)
    dev, host = (tmp_path / "dev.msh").read_bytes(), (tmp_path / "host.msh").read_bytes()
    assert dev == host, "device FASTA parser and host record parser disagree"
    if bytes_vs_oracle:
        assert dev == mo.msh_bytes(osk), "engine and oracle disagree"
    else:   # headers of tens of kilobytes: the oracle keeps 4 KiB of a header line, so only the hashes and lengths compare
        got = mo.read_msh(tmp_path / "dev.msh")
        for a, b in zip(got.references, osk.references):
            assert a.length == b.length and np.array_equal(a.hashes, b.hashes)
    return osk


def _acgt(rng, n):
    return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n))


def test_wrapped_assembly_like_fasta(tmp_path, monkeypatch):
    """C2's shape: a 3 Mb genome in 12 contigs, 70-column lines; k-mers across every line break must be there."""
    g = synth.make_genome(3_000_000, seed=5)
    p = tmp_path / "asm.fa"
    p.write_bytes(synth.genome_fasta(g, n_contigs=12, width=70))
    osk = _three_ways(tmp_path, [p], 21, 1000, monkeypatch)
    assert osk.references[0].length == 3_000_000 and osk.references[0].comment.startswith("[12 seqs] contig_1 synthetic")
    # the same genome unwrapped (one line per contig) has the same k-mers: same hashes
    q = tmp_path / "flat.fa"
    q.write_bytes(synth.genome_fasta(g, n_contigs=12, width=1 << 30))
    engine.sketch_files([q], 21, 1000, tmp_path / "flat.msh")
    assert np.array_equal(mo.read_msh(tmp_path / "flat.msh").references[0].hashes, osk.references[0].hashes)


@pytest.mark.parametrize("k,s", [(21, 1000), (27, 50000), (16, 300), (32, 100), (5, 50)])
def test_lines_and_headers_longer_than_a_tile_short_records_blanks_crlf(tmp_path, monkeypatch, k, s):
    rng = np.random.default_rng(k * 100 + s)
    recs = [
        b">first " + b"x" * 3_000 + b"\n" + _acgt(rng, 50_000) + b"\n",               # a line longer than a 16 KiB tile
        b">tiny\n" + _acgt(rng, max(1, k - 1)) + b"\n",                               # shorter than k: not counted
        b">empty\n",                                                                   # no sequence at all
        b">wrapped with spaces\n" + b"\n".join(_acgt(rng, 61) for _ in range(300)) + b"\n\n\n",
        b">blanks inside\n" + _acgt(rng, 30) + b" " + _acgt(rng, 30) + b"\t\n" + _acgt(rng, 100) + b"\n",
        b">lower and iupac\n" + _acgt(rng, 400).lower() + b"NNNNRYKM" + _acgt(rng, 400) + b"\n",
        b">gt inside > a line\n" + _acgt(rng, 100) + b">" + _acgt(rng, 100) + b"\n",
        b">high bytes\n" + _acgt(rng, 50) + bytes([0x80, 0xFF, 0x7F, 0x01]) + _acgt(rng, 50) + b"\n",
    ]
    for nl, tail in ((b"\n", b""), (b"\r\n", b""), (b"\n", b">dangling header without newline")):
        data = b"".join(recs).replace(b"\n", nl) + tail
        p = tmp_path / "odd.fa"
        p.write_bytes(data)
        osk = _three_ways(tmp_path, [p], k, s, monkeypatch)
        assert osk.references[0].comment.startswith("[")
    # a header line longer than a tile, and one that ends exactly on a tile boundary
    for hl in (40_000, 16384 - 1, 2 * 16384 - 1):
        p.write_bytes(b">" + b"h" * (hl - 1) + b"\n" + _acgt(rng, 20_000) + b"\n>next\n" + _acgt(rng, 1000) + b"\n")
        _three_ways(tmp_path, [p], k, s, monkeypatch, bytes_vs_oracle=False)
    # first counted record is not the first record of the file
    p.write_bytes(b">short\nAC\n>second one\n" + _acgt(rng, 500) + b"\n")
    osk = _three_ways(tmp_path, [p], k, s, monkeypatch)
    assert osk.references[0].comment == "second one"


def test_many_small_records_and_every_alignment_of_the_line_width(tmp_path, monkeypatch):
    rng = np.random.default_rng(77)
    recs = []
    for i in range(20_000):
        L = int(rng.integers(1, 120))
        w = int(rng.integers(1, 90))
        seq = _acgt(rng, L)
        recs.append(b">r%d\n" % i + b"\n".join(seq[j:j + w] for j in range(0, L, w)) + b"\n")
    p = tmp_path / "many.fa"
    p.write_bytes(b"".join(recs))
    _three_ways(tmp_path, [p], 21, 5000, monkeypatch)


def test_fastq_syntax_in_a_fasta_falls_back_to_the_record_parser(tmp_path, monkeypatch):
    """A line that starts with '+' or '@' is FASTQ syntax to kseq: the device parser must notice and hand the file to the
    host record parser, whose result is the oracle's."""
    rng = np.random.default_rng(3)
    seq = _acgt(rng, 300)
    data = b">a\n" + seq[:100] + b"\n+\n" + b"I" * 100 + b"\n>b\n" + seq[100:] + b"\n"
    p = tmp_path / "mixed.fa"
    p.write_bytes(data)
    _three_ways(tmp_path, [p], 21, 100, monkeypatch)
    # and a file that does not start with '>'
    p.write_bytes(b"\n\n>late start\n" + seq + b"\n")
    _three_ways(tmp_path, [p], 21, 100, monkeypatch)


def test_no_counted_record_is_an_error_on_both_paths(tmp_path, monkeypatch):
    p = tmp_path / "none.fa"
    p.write_bytes(b">a\nACGT\n>b\nAC\n")
    for host in (False, True):
        if host:
            monkeypatch.setenv("MHX_HOST_FASTA", "1")
        with pytest.raises(engine.NoRecordsError):
            engine.sketch_files([p], 21, 100, tmp_path / "x.msh")


def test_bgzf_and_plain_gz_assemblies(tmp_path, monkeypatch):
    """The same assembly plain, gzip'ed and bgzip'ed (BGZF blocks decoded side by side): one sketch, the oracle's."""
    import gzip

    from tests.test_lib_cpu import _bgzf

    g = synth.make_genome(2_000_000, seed=9)
    fa = synth.genome_fasta(g, n_contigs=7, width=80)
    plain, gz, bg = tmp_path / "asm.fa", tmp_path / "asm.fa.gz", tmp_path / "asm.bgzf.fa.gz"
    plain.write_bytes(fa)
    gz.write_bytes(gzip.compress(fa, 6))
    bg.write_bytes(_bgzf(fa, 6))
    osk = _three_ways(tmp_path, [plain], 27, 5000, monkeypatch)
    for p in (gz, bg):
        engine.sketch_files([p], 27, 5000, tmp_path / "z.msh")
        got = mo.read_msh(tmp_path / "z.msh").references[0]
        assert np.array_equal(got.hashes, osk.references[0].hashes) and got.length == 2_000_000
        assert got.comment == osk.references[0].comment
