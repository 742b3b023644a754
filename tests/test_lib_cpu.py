"""CPU-side checks of libmhx.so: it loads, exports every entry point include/mhx.h declares,
its host-only pieces (bounds table, .msh container, sniffers) match the reference's goldens,
and every compute entry point refuses to run without a GPU (no CPU fallback)."""
import ctypes

import os
from pathlib import Path

import numpy as np
import pytest
import torch

from auriclass_amd import engine
from oracle import mash_oracle as mo
from tests.conftest import GOLDEN, REFDATA

HAVE_GPU = torch.cuda.is_available()


@pytest.fixture(scope="module")
def lib():
    engine.build()
    return engine.load()


def test_exports_every_declared_symbol(lib):
    names = engine.declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libmhx.so does not export {n}"


def test_bounds_text_matches_reference_golden(lib):
    # /root/reference/tests/test_correct_workflow.py:26
    assert engine.bounds(27, 0.99) == (GOLDEN / "mash_bounds_k27_p0.99.txt").read_text()
    assert engine.bounds(21, 0.99) == mo.bounds_text(21, 0.99)
    assert engine.bounds(16, 0.95) == mo.bounds_text(16, 0.95)


def test_msh_writer_reproduces_reference_fixture_bytes(lib, tmp_path):
    ref = mo.read_msh(REFDATA / "ref_sketch.msh")
    out = tmp_path / "x.msh"
    engine.msh_write(out, 27, 50000, [r.name for r in ref.references], [r.comment for r in ref.references],
                     [r.length for r in ref.references], [r.hashes for r in ref.references])
    assert out.read_bytes() == (REFDATA / "ref_sketch.msh").read_bytes()


@pytest.mark.parametrize("k,nrefs,nh", [(21, 1, 1000), (21, 3, 400), (16, 2, 300), (27, 30, 1000), (21, 1, 0), (31, 200, 7)])
def test_msh_writer_equals_oracle_writer(lib, tmp_path, k, nrefs, nh):
    rng = np.random.default_rng(k + nrefs + nh)
    hi = 2 ** 32 if k <= 16 else 2 ** 64
    refs = []
    for i in range(nrefs):
        h = np.unique(rng.integers(0, hi, size=nh, dtype=np.uint64))
        refs.append(mo.Reference("name_%d.fa" % i, "c" * int(rng.integers(0, 40)), int(rng.integers(1, 10 ** 7)), h))
    sk = mo.SketchFile(kmer_size=k, sketch_size=max(nh, 1), references=refs)
    out = tmp_path / "y.msh"
    engine.msh_write(out, k, max(nh, 1), [r.name for r in refs], [r.comment for r in refs], [r.length for r in refs],
                     [r.hashes for r in refs])
    assert out.read_bytes() == mo.msh_bytes(sk)
    back = mo.read_msh(out)
    assert [r.name for r in back.references] == [r.name for r in refs]
    assert all(np.array_equal(a.hashes, b.hashes) for a, b in zip(back.references, refs))


def test_p_value_against_oracle(lib):
    for x, lr, lq, k, n in [(0, 1000, 1000, 21, 1000), (5, 5_000_000, 4_000_000, 16, 1000), (20, 12_000_000, 12_000_000, 14, 1000),
                            (48451, 48502, 48454, 27, 48476), (3, 10 ** 9, 10 ** 9, 16, 400)]:
        got = engine.p_value(x, lr, lq, k, n)
        want = mo.p_value(x, lr, lq, 4.0 ** k, n)
        assert got == pytest.approx(want, rel=1e-9, abs=1e-300)
        assert "%g" % got == "%g" % want


def test_sniffers_and_fasta_size(lib):
    assert engine.sniff_fastq(REFDATA / "NC_001416.1_1.fq.gz") and not engine.sniff_fasta(REFDATA / "NC_001416.1_1.fq.gz")
    assert engine.sniff_fasta(REFDATA / "NC_001416.1.fasta.gz") and not engine.sniff_fastq(REFDATA / "NC_001416.1.fasta.gz")
    assert not engine.sniff_fasta(REFDATA / "ref_sketch.msh") and not engine.sniff_fastq(REFDATA / "ref_sketch.msh")
    assert not engine.sniff_fastq(REFDATA / "test_empty_1.fq.gz")
    assert engine.fasta_total_bases(REFDATA / "NC_001416.1.fasta.gz") == 48502   # test_correct_workflow.py:197
    assert engine.fasta_total_bases(REFDATA / "NC_001604.1.fasta.gz") == 39937


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-GPU behaviour")
def test_compute_calls_fail_loudly_without_gpu(lib, tmp_path):
    with pytest.raises(engine.EngineError) as e:
        engine.sketch_files([REFDATA / "NC_001416.1.fasta.gz"], 27, 50000, tmp_path / "o.msh")
    assert e.value.code == engine.MHX_E_NO_DEVICE
    with pytest.raises(engine.EngineError):
        engine.dist_files(REFDATA / "ref_sketch.msh", REFDATA / "ref_sketch.msh")
    with pytest.raises(engine.EngineError):
        engine.Sketcher(21, 1000)
    with pytest.raises(engine.EngineError):
        engine.Sketcher(21, 1000, 3, budget_scale=16)
    q = np.arange(1, 101, dtype=np.uint64).reshape(1, 100)
    with pytest.raises(engine.EngineError) as e:
        engine.dist_batch(q, [100], q, [100], 21, 100)
    assert e.value.code == engine.MHX_E_NO_DEVICE
    # the host-only entry points keep working: bounds text, container writer, gunzip
    assert "Parameters (run with -h for details)" in engine.bounds(27, 0.99)


# ---- the ingest's own gzip/DEFLATE decoder against zlib ------------------------------------------
def _gz(data: bytes, level: int = 6, **kw) -> bytes:
    import gzip
    import io

    b = io.BytesIO()
    with gzip.GzipFile(fileobj=b, mode="wb", compresslevel=level, **kw) as f:
        f.write(data)
    return b.getvalue()


def _fastq_like(rng, n):
    from auriclass_amd import synth

    g = synth.make_genome(50_000, seed=3)
    return synth.make_fastq(g, n, 150, seed=int(rng.integers(1, 1000)), device="cpu").numpy().tobytes()


@pytest.mark.parametrize("level", [0, 1, 4, 6, 9])
def test_gunzip_equals_zlib_on_fastq_text_and_binary(lib, level):
    rng = np.random.default_rng(level)
    cases = [b"", b"A", b"ACGT" * 70000, bytes(rng.integers(0, 256, 300_000, dtype=np.uint8)),   # empty, tiny, long runs, incompressible
             _fastq_like(rng, 20_000),                                                                # 6.3 MB: many dynamic blocks
             bytes(rng.choice(np.frombuffer(b"ACGTN\n", np.uint8), 1_500_000))]
    for data in cases:
        assert engine.gunzip(_gz(data, level)) == data


def test_gunzip_block_types_members_and_header_fields(lib):
    import zlib

    rng = np.random.default_rng(77)
    text = _fastq_like(rng, 3000)
    # fixed-Huffman blocks (Z_FIXED), stored blocks (level 0), raw long-distance matches (32 KiB window edge)
    for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
        co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, strategy)
        assert engine.gunzip(co.compress(text) + co.flush()) == text
    far = bytes(rng.integers(0, 256, 32768, dtype=np.uint8))
    data = far + b"x" * 5 + far + far[:100] * 3
    assert engine.gunzip(_gz(data, 9)) == data
    # several members back to back (bgzip-style), with FNAME / mtime header fields, then trailing garbage
    parts = [text[:100_000], b"", text[100_000:250_000], text[250_000:]]
    blob = b"".join(_gz(p, 5, filename="reads.fq", mtime=12345) for p in parts)
    assert engine.gunzip(blob) == b"".join(parts)
    assert engine.gunzip(blob + b"\0" * 100) == b"".join(parts)
    # sync-flushed stream: empty stored blocks in the middle
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    z = b"".join(co.compress(text[i:i + 50_000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(text), 50_000)) + co.flush()
    assert engine.gunzip(z) == text


def test_gunzip_rejects_corrupt_streams(lib):
    rng = np.random.default_rng(5)
    text = _fastq_like(rng, 2000)
    z = bytearray(_gz(text, 6))
    bad = bytes(z[:-8]) + bytes([z[-8] ^ 1]) + bytes(z[-7:])           # wrong CRC
    with pytest.raises(engine.EngineError):
        engine.gunzip(bad)
    with pytest.raises(engine.EngineError):
        engine.gunzip(bytes(z[: len(z) // 2]))                          # truncated
    flipped = 0
    for pos in rng.integers(20, len(z) - 9, 40):                        # random bit flips: error or (rarely) CRC catches it
        y = bytearray(z)
        y[int(pos)] ^= 1 << int(rng.integers(0, 8))
        try:
            out = engine.gunzip(bytes(y))
            assert out == text      # a flip that does not change the output (e.g. in a header field) is fine
        except engine.EngineError:
            flipped += 1
    assert flipped >= 30


def test_gunzip_never_reads_stored_block_lengths_from_the_pad(lib):
    """a file that ends in a stored-block header `01 FF FF` used to take NLEN = 0 from the zero pad, pass the
    LEN ^ NLEN test with LEN = 65535 and memcpy 65535 bytes from beyond the compressed buffer."""
    hdr = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3])
    stored = bytes([0x00, 0x04, 0x00, 0xFB, 0xFF]) + b"ACGT"
    for tail in (b"\x01\xff\xff", b"\x01\xff", b"\x01", b"\x00\xff\xff", b"\x01\x00\x00"):
        z = hdr + stored + tail
        z += b"\0" * max(0, 18 - len(z))
        with pytest.raises(engine.EngineError):
            engine.gunzip(z)
    # a complete little file of the same shape is still fine
    import zlib
    body = stored[:0] + bytes([0x01, 0x04, 0x00, 0xFB, 0xFF]) + b"ACGT"
    z = hdr + body + zlib.crc32(b"ACGT").to_bytes(4, "little") + (4).to_bytes(4, "little")
    assert engine.gunzip(z) == b"ACGT"


def test_gunzip_header_truncation_sweep(lib):
    """Every prefix of a multi-block .gz (cut inside the member header, the HLIT/HDIST/HCLEN fields, the code-length
    code, the length runs, the data, the trailer) is refused or -- cut between members -- decodes to a prefix; none
    crashes.  The same sweep runs under AddressSanitizer in test_inflate_under_address_sanitizer."""
    rng = np.random.default_rng(9)
    text = _fastq_like(rng, 1500)
    z = _gz(text[:200_000], 6) + _gz(text[200_000:], 1)
    first = len(_gz(text[:200_000], 6))
    cuts = list(range(0, 700)) + list(range(700, len(z), 211)) + list(range(len(z) - 30, len(z)))
    for cut in cuts:
        try:
            out = engine.gunzip(z[:cut])
        except engine.EngineError:
            continue
        # accepted prefixes: nothing at all (shorter than a member), or exactly the first member
        assert out == b"" and cut < 18 or (out == text[:200_000] and first <= cut < first + 18), cut


def test_inflate_under_address_sanitizer(tmp_path):
    """CPU ASan/UBSan build of mhx_inflate.cpp with tests/asan/inflate_fuzz.cpp: exact-size input blocks (n + kInputPad),
    the crafted stored-block tails, a truncation sweep and mutation fuzzing.  Any out-of-bounds access aborts."""
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "inflate_fuzz"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           str(root / "tests" / "asan" / "inflate_fuzz.cpp"), str(root / "auriclass_amd" / "csrc" / "mhx_inflate.cpp"),
           str(root / "auriclass_amd" / "csrc" / "mhx_pinflate.cpp"), "-o", str(exe), "-lpthread", "-lz"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr and "cannot find" in b.stderr:
        pytest.skip("no sanitizer runtime on this host")
    assert b.returncode == 0, b.stderr[-2000:]
    rng = np.random.default_rng(12)
    text = _fastq_like(rng, 3000)   # ~200 KB compressed, a dozen dynamic blocks: the parallel decoder cuts it into segments
    seeds = {"dyn.gz": _gz(text, 6), "fixed_stored.gz": _gz(text[:3000], 0) + _gz(b"ACGT" * 50, 9, filename="x.fq"),
             "blocks.bgzf.gz": _bgzf(text[:60_000], 6, 3000)}   # 20 BGZF blocks: BgzfReader's threads
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               MHX_PINFLATE_MIN="1", MHX_PINFLATE_SEGMENT="8192")   # the parallel decoder engages on these small seeds too
    for i, (name, z) in enumerate(seeds.items()):
        p = tmp_path / name
        p.write_bytes(z)
        r = subprocess.run([str(exe), str(p), "600", str(17 + i)], capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, (name, r.stdout[-500:], r.stderr[-3000:])
        assert r.stdout.startswith("ok ")
        if name == "dyn.gz":   # the parallel path really ran: the intact seed decoded to its full length, mutants were refused
            assert "(seed: %d)" % len(text) in r.stdout, r.stdout
            assert int(r.stdout.split("parallel: ok ")[1].split()[2]) > 50, r.stdout


def test_parallel_gunzip_equals_zlib(lib, monkeypatch):
    """One gzip member decoded by several threads (mhx_pinflate.cpp: block search, symbolic windows, resolution) against
    zlib: FASTQ text at several levels, long runs, incompressible stretches (stored blocks in the middle), several
    members, trailing garbage; segments far smaller than in production so that every hand-off is exercised."""
    import gzip
    import zlib

    monkeypatch.setenv("MHX_PINFLATE_MIN", "1")
    monkeypatch.setenv("MHX_PINFLATE_SEGMENT", "65536")
    rng = np.random.default_rng(21)
    text = _fastq_like(rng, 40_000)                                    # 12.6 MB
    noise = bytes(rng.integers(0, 256, 400_000, dtype=np.uint8))
    mixed = text[:3_000_000] + noise + text[3_000_000:6_000_000] + b"A" * 500_000 + noise[:70_000] + text[6_000_000:]
    for data, level in ((text, 1), (text, 6), (text, 9), (mixed, 6), (b"ACGT" * 2_000_000, 6)):
        z = _gz(data, level)
        for threads in (2, 3, 8):
            assert engine.gunzip(z, threads) == data, (len(data), level, threads)
    # sync-flushed (empty stored blocks), fixed-Huffman and RLE streams
    for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_RLE, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY):
        co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, strategy)
        z = b"".join(co.compress(text[i:i + 700_000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(text), 700_000)) + co.flush()
        assert engine.gunzip(z, 4) == text, strategy
    # several members (`cat a.gz b.gz`): every large one gets the threads, the small ones the sequential decoder; then trailing padding
    parts = [text[:5_000_000], b"", text[5_000_000:9_000_000], text[9_000_000:]]
    blob = b"".join(_gz(p, 6, filename="r.fq", mtime=7) for p in parts)
    assert engine.gunzip(blob, 4) == b"".join(parts)
    assert engine.gunzip(blob + b"\0" * 333, 4) == b"".join(parts)


def test_parallel_gunzip_declines_a_stream_that_would_not_fit_in_memory(lib, monkeypatch, capfd):
    """A highly repetitive member (all-N reads: 160:1) never leaves the symbolic form, and every segment of the look-ahead
    would hold 3 bytes per inflated byte: the multi-threaded decoder gives such a member up at 16x its compressed size per
    segment (>= 8 MiB) and the sequential decoder, which streams in constant memory like zlib, produces the bytes."""
    monkeypatch.setenv("MHX_PINFLATE_MIN", "1")
    monkeypatch.setenv("MHX_PINFLATE_SEGMENT", "131072")
    monkeypatch.setenv("MHX_PINFLATE_DEBUG", "1")
    rec = b"@r\n" + b"N" * 150 + b"\n+\n" + b"#" * 150 + b"\n"
    data = rec * (120_000_000 // len(rec))
    z = _gz(data, 6)
    assert len(z) * 100 < len(data)
    capfd.readouterr()
    assert engine.gunzip(z, 4, size_hint=len(data)) == data
    assert "FAILED" in capfd.readouterr().err          # the parallel decoder did engage, and gave up
    # ordinary FASTQ text stays far below the bound
    rng = np.random.default_rng(5)
    text = _fastq_like(rng, 20_000)
    assert engine.gunzip(_gz(text, 6), 4) == text
    assert "FAILED" not in capfd.readouterr().err


def test_parallel_gunzip_refuses_what_zlib_refuses(lib, monkeypatch):
    """Truncations and bit flips anywhere in the stream: an error, never different bytes (every segment hand-off is checked
    against the successor's block boundary, the member's CRC-32 and length at the end)."""
    monkeypatch.setenv("MHX_PINFLATE_MIN", "1")
    monkeypatch.setenv("MHX_PINFLATE_SEGMENT", "32768")
    rng = np.random.default_rng(22)
    text = _fastq_like(rng, 8_000)
    z = bytearray(_gz(text, 6))
    for cut in (len(z) - 1, len(z) - 8, len(z) - 9, len(z) // 2, len(z) // 3, 100_000):
        with pytest.raises(engine.EngineError):
            engine.gunzip(bytes(z[:cut]), 4)
    refused = 0
    for pos in rng.integers(20, len(z) - 9, 60):
        y = bytearray(z)
        y[int(pos)] ^= 1 << int(rng.integers(0, 8))
        try:
            assert engine.gunzip(bytes(y), 4) == text
        except engine.EngineError:
            refused += 1
    assert refused >= 50


def test_msh_writer_equals_the_oracle_writer_on_random_containers(lib, tmp_path):
    """The C++ container writer (arena with far pointers and landing pads, like Cap'n Proto's malloc
    builder) against the independent Python one, byte for byte: 1..7 references, names / comments of
    0..300 bytes, 0..s hashes around the first-segment boundary (s near 1000), 32- and 64-bit lists."""
    rng = np.random.default_rng(4242)
    for _ in range(250):
        k = int(rng.choice([9, 16, 21, 27, 32]))
        s = int(rng.choice([1, 10, 100, 120, 130, 1000, 1010, 1020, 5000]))
        refs = []
        for i in range(int(rng.integers(1, 8))):
            nh = int(rng.integers(0, s + 1)) if rng.random() < 0.8 else s
            h = np.unique(rng.integers(0, 2 ** 32 if k <= 16 else 2 ** 64, size=nh, dtype=np.uint64))
            refs.append(mo.Reference("n" * int(rng.integers(0, 200)) + str(i), "c" * int(rng.integers(0, 300)), int(rng.integers(0, 2 ** 40)), h))
        sk = mo.SketchFile(kmer_size=k, sketch_size=s, references=refs)
        p = tmp_path / "x.msh"
        engine.msh_write(p, k, s, [r.name for r in refs], [r.comment for r in refs], [r.length for r in refs], [r.hashes for r in refs])
        assert p.read_bytes() == mo.msh_bytes(sk)


def _bgzf(data: bytes, level: int = 6, block: int = 0xFF00, eof_block: bool = True) -> bytes:
    """bgzip's container: gzip members of <= 64 KiB each with a 'BC' extra subfield holding the member's size."""
    import struct
    import zlib

    out = bytearray()
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        d = c.compress(chunk) + c.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(d) + 25) + d
        out += struct.pack("<II", zlib.crc32(chunk), len(chunk))
    if eof_block:
        out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    return bytes(out)


def test_bgzf_blocks_are_decoded_side_by_side():
    """bgzip output (independent members that announce their size): decoded by several threads, every block's CRC-32 and
    length checked; other members may follow the run of blocks; a handful of blocks is left to the other decoders."""
    import gzip

    rng = np.random.default_rng(21)
    text = bytes(rng.choice(np.frombuffer(b"ACGT\n@+I#5", np.uint8), size=3_300_000))
    for level, block in ((1, 0xFF00), (6, 0xFF00), (9, 10_000), (0, 65_280)):
        z = _bgzf(text, level, block)
        assert gzip.decompress(z) == text
        for threads in (2, 5):
            assert engine.gunzip(z, threads=threads, size_hint=len(text)) == text
    z = _bgzf(text)
    assert engine.gunzip(z + gzip.compress(b"an ordinary member behind the blocks\n"), threads=4) == text + b"an ordinary member behind the blocks\n"
    assert engine.gunzip(_bgzf(text[:200_000], eof_block=False), threads=4) == text[:200_000]          # 4 blocks: not worth the threads
    assert engine.gunzip(_bgzf(b""), threads=4) == b""
    for pos in (len(z) // 3, len(z) // 2, len(z) - 40):                                                 # payload or trailer damaged
        bad = bytearray(z)
        bad[pos] ^= 0x5A
        with pytest.raises(engine.EngineError):
            engine.gunzip(bytes(bad), threads=4)
    with pytest.raises(engine.EngineError):                                                             # cut inside a block
        engine.gunzip(z[: len(z) // 2], threads=4)


def test_fasta_reader_takes_bgzf(tmp_path):
    """A bgzip'ed FASTA (kept indexable by faidx), with and without an ordinary member behind its blocks, through the
    whole-file reader (mhx_fasta_total_bases; the FASTA sketch path reads files the same way)."""
    import gzip

    rng = np.random.default_rng(31)
    seq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=1_300_000))
    fa = b">c1 some contig\n" + b"\n".join(seq[i:i + 60] for i in range(0, len(seq), 60)) + b"\n"
    p = tmp_path / "a.fasta.gz"
    p.write_bytes(_bgzf(fa, 6))
    assert engine.fasta_total_bases(p) == len(seq)
    p.write_bytes(_bgzf(fa, 6) + gzip.compress(b">c2\nACGTNACGT\n"))
    assert engine.fasta_total_bases(p) == len(seq) + 9
    bad = bytearray(_bgzf(fa, 6))
    bad[len(bad) // 2] ^= 0x11
    p.write_bytes(bytes(bad))
    with pytest.raises(engine.EngineError):
        engine.fasta_total_bases(p)


@pytest.mark.timeout(120)
def test_whole_file_reader_on_concatenated_members_longer_than_the_last_one_announces(tmp_path):
    """`cat a.gz b.gz c.gz` through the whole-file reader (FASTA inputs, and FASTQ the streaming ingest declines): the
    output buffer is sized from the LAST member's length field, so the members before it make it grow while the
    sequential decoder is at work -- whose last call may run a few bytes past the room it was given.  The room left
    was computed as an unsigned difference and wrapped: the reader then spun for ever (found by running the GPU suite with
    the streaming ingest switched off).  Large first member: the multi-threaded decoder takes it, the rest goes to the
    sequential one; small members: all sequential; both must deliver every base."""
    import gzip

    rng = np.random.default_rng(3)

    def fasta(n_rec, length, tag):
        return b"".join(b">%s%d\n" % (tag, i) + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=length)) + b"\n" for i in range(n_rec))

    for sizes in ((400, 1, 700), (3, 40, 5), (900, 2)):
        parts = [fasta(n, 9000, b"r%d_" % j) for j, n in enumerate(sizes)]
        p = tmp_path / ("m%d.fa.gz" % sizes[0])
        p.write_bytes(b"".join(gzip.compress(x, compresslevel=4) for x in parts))
        assert engine.fasta_total_bases(p) == 9000 * sum(sizes)
        assert engine.sniff_fasta(p) and not engine.sniff_fastq(p)


def test_fastq_tail_completeness_rule(lib):
    """The host's look at the end of a 4-line FASTQ stream (`fastq_tail_complete`, mhx_fastx.cpp): a last record with its
    `+` line but no, too short or too long a quality string is incomplete (kseq_read: -2); a header alone, a record cut in
    its sequence line, a complete record with or without its final newline, with CRLF, are records."""
    def f(data, n):
        return engine.fastq_tail_complete(data[:n])

    seq = b"ACGT" * 37 + b"AC"
    body = b"@r1 x\n" + seq + b"\n+\n" + b"@" * 150 + b"\n"      # (a quality line that begins with '@')
    cases = {
        b"@l\n" + seq + b"\n+\n": False, b"@l\n" + seq + b"\n+": False, b"@l\n" + seq + b"\n+l\n" + b"I" * 100 + b"\n": False,
        b"@l\n" + seq + b"\n+\n" + b"I" * 149: False, b"@l\n" + seq + b"\n+\n" + b"I" * 151 + b"\n": False,
        b"@l\n" + seq[:77]: True, b"@l\n" + seq + b"\n": True, b"@l\n": True, b"@l": True, b"": True,
        b"@l\n" + seq + b"\n+\n" + b"I" * 150: True, b"@l\r\n" + seq + b"\r\n+\r\n" + b"I" * 150 + b"\r\n": True,
        b"@l\n" + seq + b"\n+\n" + b"@" * 150 + b"\n": True, b"@l\n\n+\n\n": True, b"@l\n\n+\n": True,
    }
    for tail, want in cases.items():
        for lead in (body, b"", body * 3):
            data = lead + tail
            assert f(data, len(data)) == want, (tail[:20], len(lead))
