"""Seed-pinned synthetic inputs of BASELINE.md §3 / SURVEY.md §8(d): an iid ACGT genome and
FASTQ reads drawn from it (uniform start, random strand, per-base substitutions), laid out as
315-byte records for 150 bp reads: '@r%08d\\n' + bases + '\\n+\\n' + 'I' * len + '\\n'.

Random draws come from numpy (so the bytes do not depend on the device); the byte assembly
runs in torch on whatever device is asked for, in chunks, so 10 M reads take seconds on a GPU.
"""
from __future__ import annotations

import numpy as np
import torch

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_genome(n_bases: int = 12_000_000, seed: int = 42) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return _ACGT[rng.integers(0, 4, size=n_bases, dtype=np.uint8)]


def genome_fasta(genome: np.ndarray, n_contigs: int = 20, width: int = 70, name: str = "contig") -> bytes:
    """The genome cut into contigs, wrapped FASTA."""
    out = []
    bounds = np.linspace(0, len(genome), n_contigs + 1).astype(np.int64)
    for i in range(n_contigs):
        seq = genome[bounds[i]:bounds[i + 1]].tobytes()
        out.append(b">%s_%d synthetic\n" % (name.encode(), i + 1))
        out.append(b"\n".join(seq[j:j + width] for j in range(0, len(seq), width)))
        out.append(b"\n")
    return b"".join(out)


def mutate(genome: np.ndarray, rate: float, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    g = genome.copy()
    n = rng.binomial(len(g), rate)
    pos = rng.integers(0, len(g), size=n)
    code = np.searchsorted(_ACGT, g[pos])
    g[pos] = _ACGT[(code + rng.integers(1, 4, size=n)) % 4]
    return g


def record_bytes(read_len: int = 150) -> int:
    return 11 + read_len + 3 + read_len + 1


def make_fastq(genome: np.ndarray, n_reads: int, read_len: int = 150, seed: int = 43, sub_rate: float = 0.005,
               device: str = "cpu", first_index: int = 0, chunk_reads: int = 1_000_000) -> torch.Tensor:
    """uint8 tensor of n_reads * record_bytes(read_len) bytes on `device`."""
    rng = np.random.default_rng(seed)
    G = len(genome)
    starts = rng.integers(0, G - read_len + 1, size=n_reads, dtype=np.int64)
    strand = rng.integers(0, 2, size=n_reads, dtype=np.uint8)
    n_sub = int(rng.binomial(n_reads * read_len, sub_rate))
    sub_pos = rng.integers(0, n_reads * read_len, size=n_sub, dtype=np.int64)
    sub_shift = rng.integers(1, 4, size=n_sub, dtype=np.uint8)
    order = np.argsort(sub_pos, kind="stable")
    sub_pos, sub_shift = sub_pos[order], sub_shift[order]

    dev = torch.device(device)
    rb = record_bytes(read_len)
    out = torch.empty(n_reads * rb, dtype=torch.uint8, device=dev)
    g_codes = torch.from_numpy(np.searchsorted(_ACGT, genome).astype(np.uint8)).to(dev)
    acgt = torch.from_numpy(_ACGT.copy()).to(dev)
    ar = torch.arange(read_len, device=dev, dtype=torch.int64)
    pow10 = torch.tensor([10 ** (7 - i) for i in range(8)], device=dev, dtype=torch.int64)
    for c0 in range(0, n_reads, chunk_reads):
        c1 = min(n_reads, c0 + chunk_reads)
        n = c1 - c0
        st = torch.from_numpy(starts[c0:c1]).to(dev)
        sd = torch.from_numpy(strand[c0:c1]).to(dev).to(torch.bool)
        # forward strand: genome[start + i]; reverse strand: 3 - genome[start + L-1-i]
        idx = st[:, None] + torch.where(sd[:, None], (read_len - 1) - ar[None, :], ar[None, :])
        codes = g_codes[idx]
        codes = torch.where(sd[:, None], 3 - codes, codes)
        lo, hi = np.searchsorted(sub_pos, [c0 * read_len, c1 * read_len])
        if hi > lo:
            p = torch.from_numpy(sub_pos[lo:hi] - c0 * read_len).to(dev)
            sh = torch.from_numpy(sub_shift[lo:hi]).to(dev)
            flat = codes.reshape(-1)
            flat[p] = (flat[p] + sh) % 4
        rec = out[c0 * rb:c1 * rb].view(n, rb)
        rec[:, 0] = ord("@")
        rec[:, 1] = ord("r")
        ids = torch.arange(first_index + c0, first_index + c1, device=dev, dtype=torch.int64)
        rec[:, 2:10] = ((ids[:, None] // pow10[None, :]) % 10 + ord("0")).to(torch.uint8)
        rec[:, 10] = ord("\n")
        rec[:, 11:11 + read_len] = acgt[codes.to(torch.int64)]
        rec[:, 11 + read_len] = ord("\n")
        rec[:, 12 + read_len] = ord("+")
        rec[:, 13 + read_len] = ord("\n")
        rec[:, 14 + read_len:14 + 2 * read_len] = ord("I")
        rec[:, 14 + 2 * read_len] = ord("\n")
    return out


def make_fastq_range(genome: np.ndarray, lo: int, hi: int, read_len: int = 150, seed0: int = 43, block_reads: int = 10_000_000,
                     device: str = "cpu") -> torch.Tensor:
    """Reads [lo, hi) of the C4 read set: block b = reads [b * block_reads, (b + 1) * block_reads) drawn
    with seed seed0 + b, so the union over any record-aligned partition is the same byte stream whatever the number of
    ranks.  A block is generated whole and the wanted part kept (the draws of one seed are not addressable by position)."""
    rb = record_bytes(read_len)
    out = torch.empty((hi - lo) * rb, dtype=torch.uint8, device=torch.device(device))
    pos = 0
    for b in range(lo // block_reads, (max(hi, lo + 1) - 1) // block_reads + 1):
        b0 = b * block_reads
        a, e = max(lo, b0), min(hi, b0 + block_reads)
        if e <= a:
            continue
        blk = make_fastq(genome, block_reads, read_len, seed=seed0 + b, device=device, first_index=b0)
        n = (e - a) * rb
        out[pos:pos + n] = blk[(a - b0) * rb:(e - b0) * rb]
        pos += n
        del blk
    return out
