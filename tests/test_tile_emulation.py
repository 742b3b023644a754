"""CPU emulation of the sketch kernel's tile phases (auriclass_amd/csrc/mhx_tile.h, the very
functions the HIP kernel runs) against the oracle's definition-level window hashes."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import mash_oracle as mo

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "emul" / "tile_emul.cpp"
SO = ROOT / "tests" / "emul" / "_tile_emul.so"
MAXT = 2 ** 64 - 1


@pytest.fixture(scope="module")
def emul():
    hdr = ROOT / "auriclass_amd" / "csrc" / "mhx_tile.h"
    if not SO.exists() or SO.stat().st_mtime < max(SRC.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(SO), str(SRC)], check=True)
    L = ctypes.CDLL(str(SO))
    L.emul_sketch.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                              ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    return L


def run_emul(L, data: bytes, k, fmt, T=MAXT, lead=0):
    """lead: bytes of foreign data placed before the span (span starts at an unaligned offset)."""
    pad = 65536
    raw = np.zeros(lead + len(data) + pad + 64, dtype=np.uint8)
    base_off = (-raw.ctypes.data) % 16
    buf = raw[base_off:]
    rng = np.random.default_rng(7)
    buf[:lead] = rng.choice(np.frombuffer(b"ACGT\n@+I", np.uint8), size=lead)
    buf[lead:lead + len(data)] = np.frombuffer(data, np.uint8)
    buf[lead + len(data):lead + len(data) + 48] = np.frombuffer(b"ACGT" * 12, np.uint8)  # foreign tail
    out = np.zeros(len(data) + 16, dtype=np.uint64)
    n = ctypes.c_uint64()
    stats = np.zeros(8, dtype=np.uint64)
    rc = L.emul_sketch(buf.ctypes.data, lead, lead + len(data), k, fmt, T, out.ctypes.data, len(out),
                       ctypes.byref(n), stats.ctypes.data)
    assert rc == 0
    return np.sort(out[:n.value]), stats


def oracle_hashes(seqs, k):
    parts = []
    for s in seqs:
        if len(s) < k:
            continue
        o = np.zeros(len(s), dtype=np.uint64)
        b = ctypes.create_string_buffer(s, len(s))
        n = mo.lib().mo_all_window_hashes(b, len(s), k, o.ctypes.data)
        parts.append(o[:n])
    return np.sort(np.concatenate(parts)) if parts else np.zeros(0, np.uint64)


def random_reads(rng, n, lo, hi, p_n=0.2, p_lower=0.2):
    reads = []
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        r = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L)
        if L and rng.random() < p_n:
            for _ in range(int(rng.integers(1, 3))):
                r[int(rng.integers(0, L))] = rng.choice(np.frombuffer(b"NRYKM-*", np.uint8))
        r = bytes(r)
        if rng.random() < p_lower:
            r = r.lower()
        reads.append(r)
    return reads


def fastq_bytes(rng, reads):
    qual_alphabet = np.frombuffer(b"!#+@ACGTIJ5<?acgt", np.uint8)
    out = []
    for i, r in enumerate(reads):
        q = bytes(rng.choice(qual_alphabet, size=len(r)))
        out.append(b"@r%d/1 x\n" % i + r + b"\n+\n" + q + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("k", [21, 27, 16, 11, 32, 31, 5, 1])
def test_seq_stream(emul, k):
    rng = np.random.default_rng(100 + k)
    reads = random_reads(rng, 300, 0, 400)
    data = b"\n".join(reads) + b"\n"
    got, stats = run_emul(emul, data, k, fmt=0)
    want = oracle_hashes(reads, k)
    # windows hashed = every start with k bytes inside one record; the ones holding a non-ACGT byte are dropped
    # by the deferred base check, so the admitted hashes are exactly the oracle's
    assert stats[0] == sum(max(0, len(r) - k + 1) for r in reads) >= len(want)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("k,lead", [(21, 0), (21, 37), (27, 5000), (16, 32768 + 11), (31, 1)])
def test_fastq4_stream(emul, k, lead):
    rng = np.random.default_rng(200 + k + lead)
    reads = random_reads(rng, 700, 1, 260)
    data = fastq_bytes(rng, reads)
    assert len(data) > 3 * 32768
    got, stats = run_emul(emul, data, k, fmt=1, lead=lead)
    want = oracle_hashes(reads, k)
    assert int(stats[3]) == 0, "format flag raised on a valid FASTQ"
    assert int(stats[5]) == 0, "a tile found a line phase that is not the running line count's"
    assert int(stats[6]) == 0, "short reads: every tile finds its phase by itself"
    assert int(stats[2]) == 4 * len(reads)
    assert stats[0] == sum(max(0, len(r) - k + 1) for r in reads) >= len(want)
    assert np.array_equal(got, want)


def test_fastq4_no_trailing_newline_and_long_lines(emul):
    rng = np.random.default_rng(5)
    reads = random_reads(rng, 6, 30000, 70000, p_n=0.5)   # lines longer than a tile
    data = fastq_bytes(rng, reads)[:-1]
    got, stats = run_emul(emul, data, 21, fmt=1, lead=3)
    assert np.array_equal(got, oracle_hashes(reads, 21))
    assert int(stats[3]) == 0
    assert int(stats[5]) == 0 and int(stats[6]) > 0   # such tiles are left to the look-back (repair) pass


@pytest.mark.parametrize("seed", range(6))
def test_selfsync_phase_is_the_running_line_count(emul, seed):
    """Quality lines that begin with '@' or '+', empty reads, reads of 1..3000 bases: whenever a tile names a phase it
    is the true one, and the chain check has nothing to flag."""
    rng = np.random.default_rng(900 + seed)
    hi = [40, 300, 3000, 3000, 9000, 120][seed]
    reads = random_reads(rng, 400 if hi > 1000 else 3000, 0, hi)
    out = []
    for i, r in enumerate(reads):
        q = bytearray(rng.choice(np.frombuffer(b"@+I#", np.uint8), size=len(r)).tobytes())
        if q and rng.random() < 0.5:
            q[0] = ord("@") if rng.random() < 0.5 else ord("+")
        out.append(b"@%d\n" % i + r + b"\n+" + (b"@x" if rng.random() < 0.3 else b"") + b"\n" + bytes(q) + b"\n")
    data = b"".join(out)
    got, stats = run_emul(emul, data, 21, fmt=1, lead=int(rng.integers(0, 70000)))
    assert int(stats[3]) == 0 and int(stats[5]) == 0
    assert np.array_equal(got, oracle_hashes(reads, 21))


def test_record_cut_short_in_front_of_a_tile_border_breaks_the_chain(emul):
    """kseq reads '@h\\nSEQ\\n@h2...' as a record without qualities; the 4-line fast path must refuse it even when the
    cut falls so that every tile's own first lines look regular."""
    rng = np.random.default_rng(77)
    reads = random_reads(rng, 600, 100, 150, p_n=0, p_lower=0)
    recs = [b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n" for i, r in enumerate(reads)]
    body, i = b"", 0
    while len(body) < 32768 - 900:
        body += recs[i]
        i += 1
    gap = 32768 - 100 - len(body)            # the cut record's header starts ~100 bytes in front of the border
    L = (gap - 7) // 2
    body += b"@f\n" + b"C" * L + b"\n+\n" + b"I" * L + b"\n"
    data = body + b"@cut\n" + b"ACGT" * 60 + b"\n" + b"".join(recs[i:])
    assert len(body) + 5 < 32768 < len(body) + 5 + 240
    _, stats = run_emul(emul, data, 21, fmt=1)
    assert int(stats[5]) & 1, "the tile after the border takes the rest of the cut line for a quality line"
    assert (int(stats[3]) & 2) or (int(stats[5]) & 2)


def test_fastq_layout_violation_is_flagged(emul):
    rng = np.random.default_rng(6)
    reads = random_reads(rng, 50, 50, 100)
    good = fastq_bytes(rng, reads)
    bad = good.replace(b"\n+\n", b"\nX\n", 1)
    _, stats = run_emul(emul, bad, 21, fmt=1)
    assert int(stats[3]) & 2
    multi = b"@r\nACGT\nACGT\n+\nIIIIIIII\n" + good   # multi-line record
    _, stats = run_emul(emul, multi, 21, fmt=1)
    assert int(stats[3]) & 2


def test_threshold_filter(emul):
    rng = np.random.default_rng(8)
    reads = random_reads(rng, 200, 100, 200, p_n=0, p_lower=0)
    data = b"\n".join(reads)
    T = 2 ** 60
    got, stats = run_emul(emul, data, 21, fmt=0, T=T)
    want = oracle_hashes(reads, 21)
    assert np.array_equal(got, want[want <= T])
    assert stats[1] == len(got)


def test_empty_and_tiny(emul):
    for data in (b"", b"A", b"ACGTACGTACGTACGTACGT", b"ACGTACGTACGTACGTACGTA"):
        got, _ = run_emul(emul, data, 21, fmt=0)
        assert np.array_equal(got, oracle_hashes([data], 21))


def palindromic_edge_reads(rng, k, n):
    """Windows whose first 8 bases equal the first 8 bases of their reverse complement, so the
    strand decision falls through the fast 8-base comparison into the exact one."""
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    out = []
    for _ in range(n):
        x = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=8)
        mid = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=max(0, k - 16))
        kmer = np.concatenate([x, mid, np.array([comp[int(c)] for c in x[::-1]], np.uint8)])[:k] if k >= 16 else None
        if kmer is None:      # 8 < k < 16: overlap the two ends
            kmer = np.concatenate([x, np.array([comp[int(c)] for c in x[::-1]], np.uint8)[16 - k:]])
            # make the overlapping part self-consistent: retry until first 8 == rc first 8
            rc = np.array([comp[int(c)] for c in kmer[::-1]], np.uint8)
            if not np.array_equal(kmer[:8], rc[:8]):
                continue
        flank_l = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(0, 12)))
        flank_r = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(0, 12)))
        out.append(bytes(np.concatenate([flank_l, kmer, flank_r])))
    return out


@pytest.mark.parametrize("k", [16, 17, 20, 21, 24, 27, 31, 32])
def test_strand_decision_fallback_branch(emul, k):
    rng = np.random.default_rng(900 + k)
    reads = palindromic_edge_reads(rng, k, 3000)
    assert len(reads) > 20
    data = b"\n".join(reads) + b"\n"
    got, _ = run_emul(emul, data, k, fmt=0)
    assert np.array_equal(got, oracle_hashes(reads, k))


def test_fastq4_crlf_and_plus_line_with_text(emul):
    rng = np.random.default_rng(77)
    reads = random_reads(rng, 500, 20, 200)
    quals = np.frombuffer(b"!#+@ACGTIJ5<?", np.uint8)
    rec = [b"@r%d\r\n" % i + r + b"\r\n+r%d repeated\r\n" % i + bytes(rng.choice(quals, size=len(r))) + b"\r\n"
           for i, r in enumerate(reads)]
    data = b"".join(rec)
    got, stats = run_emul(emul, data, 21, fmt=1, lead=5)
    assert int(stats[3]) == 0
    assert np.array_equal(got, oracle_hashes(reads, 21))


def test_fastq4_empty_reads_and_single_record(emul):
    rec = b"@a\n\n+\n\n" + b"@b\nACGTACGTACGTACGTACGTACGTA\n+\nIIIIIIIIIIIIIIIIIIIIIIIII\n" + b"@c\n\n+\n\n"
    got, stats = run_emul(emul, rec, 21, fmt=1)
    assert int(stats[3]) == 0 and int(stats[2]) == 12
    assert np.array_equal(got, oracle_hashes([b"ACGTACGTACGTACGTACGTACGTA"], 21))
    one = b"@only\n" + b"ACGT" * 10 + b"\n+\n" + b"I" * 40
    got, stats = run_emul(emul, one, 21, fmt=1, lead=9)
    assert np.array_equal(got, oracle_hashes([b"ACGT" * 10], 21))


@pytest.mark.parametrize("k", [21, 27, 32, 5])
def test_fastq_counts_records_with_sequence_of_at_least_k_bytes(emul, k):
    """mash counts a sequence only when it is at least k long (the "[N seqs]" comment); the device
    counts them at the header-ending newlines, across word, thread and tile borders."""
    rng = np.random.default_rng(500 + k)
    reads = random_reads(rng, 700, 0, 90, p_n=0.3)
    data = fastq_bytes(rng, reads)
    assert len(data) > 2 * 16384
    got, stats = run_emul(emul, data, k, fmt=1, lead=13)
    assert int(stats[3]) == 0
    assert int(stats[7]) == sum(1 for r in reads if len(r) >= k)
    # last record without its final newline
    got, stats = run_emul(emul, data[:-1], k, fmt=1)
    assert int(stats[7]) == sum(1 for r in reads if len(r) >= k)


@pytest.mark.parametrize("k", [21, 32])
def test_fastq_with_crlf_line_ends(emul, k):
    """CRLF files: the CR is a non-base byte (no k-mer crosses it) and does not count towards the
    sequence length (kseq drops it), so reads of exactly k-1 bases stay uncounted."""
    rng = np.random.default_rng(900 + k)
    reads = random_reads(rng, 400, k - 3, k + 3, p_n=0.1)
    data = fastq_bytes(rng, reads).replace(b"\n", b"\r\n")
    got, stats = run_emul(emul, data, k, fmt=1, lead=5)
    ref = mo.Sketcher(k, 1 << 20, 1)
    ref.add_fastx(data)
    want, _ = ref.finish()
    assert int(stats[3]) == 0
    assert np.array_equal(np.unique(got), want)
    assert int(stats[7]) == ref.records == sum(1 for r in reads if len(r) >= k)
    got, stats = run_emul(emul, data[:-1], k, fmt=1)     # ends with a bare CR
    assert int(stats[7]) == ref.records


@pytest.mark.parametrize("seed", range(40))
def test_damaged_fastq_never_gets_a_wrong_phase_past_the_checks(emul, seed):
    """Lines deleted, duplicated or cut, bytes overwritten: whenever some tile's self-found phase differs from the running
    line count, the chain check or the per-record layout check must have fired (the engine then drops the result and
    parses with the record parser), so a silently mis-phased tile is impossible."""
    rng = np.random.default_rng(4000 + seed)
    reads = random_reads(rng, 900, 20, 260)
    lines = fastq_bytes(rng, reads).split(b"\n")[:-1]
    for _ in range(int(rng.integers(1, 4))):
        i = int(rng.integers(0, len(lines)))
        what = int(rng.integers(0, 5))
        if what == 0:
            del lines[i]
        elif what == 1:
            lines.insert(i, lines[i])
        elif what == 2:
            lines[i] = lines[i][: len(lines[i]) // 2]
            del lines[i + 1: i + 1 + int(rng.integers(0, 3))]
        elif what == 3:
            lines[i] = bytes(rng.choice(np.frombuffer(b"@+ACGT\n", np.uint8), size=max(1, len(lines[i]))))
        else:
            lines[i] = b"@" + lines[i][1:] if lines[i] else b"@"
    data = b"\n".join(lines) + b"\n"
    _, stats = run_emul(emul, data, 21, fmt=1, lead=int(rng.integers(0, 20000)))
    if int(stats[5]) & 1:
        assert (int(stats[5]) & 2) or (int(stats[3]) & 2)


def test_tiles_of_very_short_lines(emul):
    """Reads of 0..3 bases under one-letter names: ~6000 newlines per 16 KiB tile; counts of records with >= k bases and
    the layout check included."""
    rng = np.random.default_rng(13)
    reads = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(0, 4)))) for _ in range(30000)]
    data = b"".join(b"@r\n" + r + b"\n+\n" + b"I" * len(r) + b"\n" for r in reads)
    for k in (1, 2, 3):
        got, stats = run_emul(emul, data, k, fmt=1, lead=5)
        assert int(stats[3]) == 0 and int(stats[5]) == 0
        assert int(stats[2]) == 4 * len(reads)
        assert int(stats[7]) == sum(1 for r in reads if len(r) >= k)
        assert np.array_equal(got, oracle_hashes(reads, k))
    bad = data.replace(b"\n+\n", b"\n-\n", 1)   # one plus line damaged
    _, stats = run_emul(emul, bad, 2, fmt=1)
    assert int(stats[3]) & 2
