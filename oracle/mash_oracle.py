"""
oracle/mash_oracle.py -- CPU ORACLE (test infrastructure, NOT product code).

Python half of the restatement of what AuriClass gets from the `mash` 2.x binary
(call sites: /root/reference/auriclass/classes.py:92-104 dist, 305-318 bounds,
576-596 sketch -r -m, 696-713 sketch).  The hot loops (MurmurHash3, k-mer walk,
MinHashHeap, merge) live in oracle/mashcore.c and are reached through ctypes; this
file restates the parts that are text or bytes: the unpacked Cap'n Proto `.msh`
container (layout decoded from /root/reference/tests/data/ref_sketch.msh), the
`mash dist` row format, the p-value and the `mash bounds` table.

Pinned by the reference's goldens in tests/test_oracle_golden.py.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under auriclass_amd/ does.
"""
from __future__ import annotations

import ctypes
import gzip
import math
import os
import struct
import subprocess
from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "libmashoracle.so"
_lib = None


def build() -> Path:
    """Compile oracle/mashcore.c with the committed Makefile (gcc)."""
    subprocess.run(["make", "-s", "-C", str(_HERE)], check=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < (_HERE / "mashcore.c").stat().st_mtime:
        build()
    L = ctypes.CDLL(str(_LIB_PATH))
    u64p = ctypes.POINTER(ctypes.c_uint64)
    u32p = ctypes.POINTER(ctypes.c_uint32)
    L.mo_murmur3_x64_128.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, u64p]
    L.mo_kmer_hash.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint32]
    L.mo_kmer_hash.restype = ctypes.c_uint64
    L.mo_sketch_new.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
    L.mo_sketch_new.restype = ctypes.c_void_p
    L.mo_sketch_free.argtypes = [ctypes.c_void_p]
    L.mo_sketch_add_seq.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
    L.mo_sketch_add_fastx.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.mo_sketch_add_fastx.restype = ctypes.c_int64
    L.mo_sketch_finish.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.mo_sketch_finish.restype = ctypes.c_uint32
    for fn in ("mo_sketch_set_size", "mo_sketch_multiplicity"):
        getattr(L, fn).argtypes = [ctypes.c_void_p]
        getattr(L, fn).restype = ctypes.c_double
    for fn in ("mo_sketch_records", "mo_sketch_length", "mo_sketch_kmers"):
        getattr(L, fn).argtypes = [ctypes.c_void_p]
        getattr(L, fn).restype = ctypes.c_uint64
    L.mo_sketch_skipped_short.argtypes = [ctypes.c_void_p]
    L.mo_sketch_skipped_short.restype = ctypes.c_int
    for fn in ("mo_sketch_first_name", "mo_sketch_first_comment"):
        getattr(L, fn).argtypes = [ctypes.c_void_p]
        getattr(L, fn).restype = ctypes.c_char_p
    L.mo_all_window_hashes.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
    L.mo_all_window_hashes.restype = ctypes.c_uint64
    L.mo_compare.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                             ctypes.c_uint64, ctypes.c_int, u64p, u64p, ctypes.POINTER(ctypes.c_double)]
    _lib = L
    return L


# --------------------------------------------------------------------------- #
# text helpers
# --------------------------------------------------------------------------- #
def fmt_g(x: float) -> str:
    """C++ `ostream << double` at default precision == printf("%g")."""
    return "%g" % x


def read_maybe_gz(path) -> bytes:
    with open(path, "rb") as fh:
        head = fh.read(2)
    if head == b"\x1f\x8b":
        with gzip.open(path, "rb") as fh:
            return fh.read()
    with open(path, "rb") as fh:
        return fh.read()


# --------------------------------------------------------------------------- #
# sketching
# --------------------------------------------------------------------------- #
@dataclass
class Reference:
    name: str
    comment: str
    length: int
    hashes: np.ndarray  # ascending uint64 (values < 2^32 when k <= 16)
    counts: Optional[np.ndarray] = None


@dataclass
class SketchFile:
    kmer_size: int
    sketch_size: int
    references: List[Reference] = field(default_factory=list)
    concatenated: bool = True
    noncanonical: bool = False
    preserve_case: bool = False
    hash_seed: int = 42
    window_size: int = 0
    error: float = 0.0
    alphabet: str = "ACGT"

    @property
    def use64(self) -> bool:
        return 4.0 ** self.kmer_size > 2.0 ** 32


class Sketcher:
    """One mash MinHashHeap + sketchFile() bookkeeping (ctypes handle)."""

    def __init__(self, k: int, s: int, m: int = 1):
        self.k, self.s, self.m = k, s, m
        self._h = lib().mo_sketch_new(k, s, m)

    def add_fastx(self, data: bytes) -> int:
        buf = ctypes.create_string_buffer(data, len(data))
        n = lib().mo_sketch_add_fastx(self._h, buf, len(data))
        if n < 0:
            raise ValueError("truncated quality string")
        return n

    def add_seq(self, seq: bytes) -> None:
        buf = ctypes.create_string_buffer(seq, len(seq))
        lib().mo_sketch_add_seq(self._h, buf, len(seq))

    def finish(self) -> Tuple[np.ndarray, np.ndarray]:
        hashes = np.zeros(self.s + 1, dtype=np.uint64)
        counts = np.zeros(self.s + 1, dtype=np.uint32)
        n = lib().mo_sketch_finish(self._h, hashes.ctypes.data, counts.ctypes.data)
        return hashes[:n].copy(), counts[:n].copy()

    set_size = property(lambda self: lib().mo_sketch_set_size(self._h))
    multiplicity = property(lambda self: lib().mo_sketch_multiplicity(self._h))
    records = property(lambda self: lib().mo_sketch_records(self._h))
    length = property(lambda self: lib().mo_sketch_length(self._h))
    kmers = property(lambda self: lib().mo_sketch_kmers(self._h))
    skipped_short = property(lambda self: bool(lib().mo_sketch_skipped_short(self._h)))
    first_name = property(lambda self: lib().mo_sketch_first_name(self._h).decode())
    first_comment = property(lambda self: lib().mo_sketch_first_comment(self._h).decode())

    def comment(self) -> str:
        """mash sketchFile(): '<name> <comment>' of the first counted record,
        wrapped as '[N seqs] ... [...]' when more than one record counted."""
        c = self.first_name + " " + self.first_comment
        if self.records > 1:
            c = "[%d seqs] %s [...]" % (self.records, c)
        return c

    def __del__(self):
        try:
            lib().mo_sketch_free(self._h)
        except Exception:
            pass


class NoRecordsError(ValueError):
    pass


def sketch_files(paths: Sequence, k: int, s: int, reads: bool = False, m: int = 1
                 ) -> Tuple[SketchFile, str]:
    """`mash sketch [-r -m M] -k K -s S paths...` -> (sketch, stderr text).

    reads=False: one reference per file, length = sum of record lengths.
    reads=True : one reference over all files, length = uint64(estimateSetSize).
    """
    out = SketchFile(kmer_size=k, sketch_size=s)
    stderr: List[str] = []
    if reads:
        sk = Sketcher(k, s, m)
        for p in paths:
            sk.add_fastx(read_maybe_gz(p))
        if sk.records == 0:
            raise NoRecordsError('ERROR: Did not find fasta records in "%s".' % paths[0])
        hashes, counts = sk.finish()
        # counts32 is only stored with `mash sketch -M`, which AuriClass never passes
        out.references.append(Reference(str(paths[0]), sk.comment(), int(sk.set_size), hashes, None))
        stderr.append("Estimated genome size: %s" % fmt_g(sk.set_size))
        stderr.append("Estimated coverage:    %s" % fmt_g(sk.multiplicity))
    else:
        for p in paths:
            stderr.append("Sketching %s..." % p)
            sk = Sketcher(k, s, 1)
            sk.add_fastx(read_maybe_gz(p))
            if sk.records == 0:
                raise NoRecordsError('ERROR: Did not find fasta records in "%s".' % p)
            hashes, counts = sk.finish()
            out.references.append(Reference(str(p), sk.comment(), int(sk.length), hashes, None))
    return out, "\n".join(stderr) + "\n"


def bruteforce_sketch(seqs: Sequence[bytes], k: int, s: int, m: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    """Definition-level reference: all window hashes -> sort -> RLE -> count>=m -> first s."""
    parts = []
    for seq in seqs:
        if len(seq) < k:
            continue
        out = np.zeros(len(seq), dtype=np.uint64)
        buf = ctypes.create_string_buffer(seq, len(seq))
        n = lib().mo_all_window_hashes(buf, len(seq), k, out.ctypes.data)
        parts.append(out[:n])
    if not parts:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint32)
    allh = np.concatenate(parts)
    vals, cnts = np.unique(allh, return_counts=True)
    keep = cnts >= m
    return vals[keep][:s], cnts[keep][:s].astype(np.uint32)


# --------------------------------------------------------------------------- #
# .msh container (unpacked Cap'n Proto, mash MinHash.capnp)
# --------------------------------------------------------------------------- #
class _Arena:
    """capnp MallocMessageBuilder arena behaviour that decides the bytes: first
    segment 1024 words; GROW_HEURISTICALLY (next size = total allocated so far);
    an object is placed in its pointer's segment if it fits, else (with one extra
    landing-pad word) in the most recently created segment, else in a new one."""

    def __init__(self):
        self.segs: List[bytearray] = []
        self.caps: List[int] = []
        self.next_size = 1024

    def _new_seg(self, min_words: int) -> int:
        size = max(min_words, self.next_size)
        if not self.segs:
            self.next_size = size
        else:
            self.next_size += size
        self.segs.append(bytearray())
        self.caps.append(size)
        return len(self.segs) - 1

    def _try(self, seg: int, n: int) -> Optional[int]:
        used = len(self.segs[seg]) // 8
        if used + n > self.caps[seg]:
            return None
        self.segs[seg] += b"\0" * (8 * n)
        return used

    def alloc_root(self) -> Tuple[int, int]:
        seg = self._new_seg(1)
        return seg, self._try(seg, 1)

    def alloc_for(self, ptr_seg: int, n: int) -> Tuple[int, int, bool]:
        """-> (segment, word offset of the allocation, is_far). When far, the first
        word of the allocation is the landing pad and the object follows it."""
        off = self._try(ptr_seg, n)
        if off is not None:
            return ptr_seg, off, False
        last = len(self.segs) - 1
        off = self._try(last, n + 1) if last != ptr_seg else None
        if off is not None:
            return last, off, True
        seg = self._new_seg(n + 1)
        return seg, self._try(seg, n + 1), True

    def put(self, seg: int, word: int, value: int) -> None:
        struct.pack_into("<Q", self.segs[seg], word * 8, value & 0xFFFFFFFFFFFFFFFF)

    def put_bytes(self, seg: int, word: int, data: bytes) -> None:
        self.segs[seg][word * 8: word * 8 + len(data)] = data


def _struct_ptr(off: int, dwords: int, nptrs: int) -> int:
    return ((off & 0x3FFFFFFF) << 2) | (dwords << 32) | (nptrs << 48)


def _list_ptr(off: int, elem_code: int, count: int) -> int:
    return 1 | ((off & 0x3FFFFFFF) << 2) | (elem_code << 32) | (count << 35)


def _far_ptr(seg: int, pad_word: int) -> int:
    return 2 | (pad_word << 3) | (seg << 32)


class _Builder:
    def __init__(self):
        self.a = _Arena()

    def _place(self, pseg: int, pword: int, nwords: int, make_ptr) -> Tuple[int, int]:
        """Allocate nwords for the pointer at (pseg, pword); write the (far) pointer.
        make_ptr(offset_words) -> encoded near pointer. Returns (seg, word) of object."""
        seg, off, far = self.a.alloc_for(pseg, nwords)
        if not far:
            self.a.put(pseg, pword, make_ptr(off - (pword + 1)))
            return seg, off
        self.a.put(pseg, pword, _far_ptr(seg, off))
        self.a.put(seg, off, make_ptr(0))
        return seg, off + 1

    def struct(self, pseg, pword, dwords, nptrs):
        return self._place(pseg, pword, dwords + nptrs, lambda o: _struct_ptr(o, dwords, nptrs))

    def text(self, pseg, pword, s: str):
        raw = s.encode("utf-8") + b"\0"
        seg, w = self._place(pseg, pword, (len(raw) + 7) // 8, lambda o: _list_ptr(o, 2, len(raw)))
        self.a.put_bytes(seg, w, raw)

    def prim_list(self, pseg, pword, arr: np.ndarray, elem_code: int):
        raw = arr.tobytes()
        seg, w = self._place(pseg, pword, (len(raw) + 7) // 8, lambda o: _list_ptr(o, elem_code, len(arr)))
        self.a.put_bytes(seg, w, raw)

    def composite_list(self, pseg, pword, count, dwords, nptrs):
        words = count * (dwords + nptrs)
        seg, w = self._place(pseg, pword, words + 1, lambda o: _list_ptr(o, 7, words))
        self.a.put(seg, w, ((count & 0x3FFFFFFF) << 2) | (dwords << 32) | (nptrs << 48))
        return seg, w + 1


def msh_bytes(sk: SketchFile) -> bytes:
    """mash Sketch::writeToCapnp, in its allocation order."""
    b = _Builder()
    rseg, rword = b.a.alloc_root()
    seg, root = b.struct(rseg, rword, 3, 4)
    pbase = root + 3
    # referenceListOld when seed == 42, else referenceList
    lseg, lst = b.struct(seg, pbase + (0 if sk.hash_seed == 42 else 3), 0, 1)
    eseg, elems = b.composite_list(lseg, lst, len(sk.references), 2, 7)
    for i, ref in enumerate(sk.references):
        e = elems + i * 9
        b.text(eseg, e + 2 + 2, ref.name)
        b.text(eseg, e + 2 + 3, ref.comment)
        b.a.put(eseg, e + 1, ref.length)
        if len(ref.hashes):
            if sk.use64:
                b.prim_list(eseg, e + 2 + 5, ref.hashes.astype("<u8"), 5)
            else:
                b.prim_list(eseg, e + 2 + 4, ref.hashes.astype("<u4"), 4)
            if ref.counts is not None and len(ref.counts):
                b.prim_list(eseg, e + 2 + 6, ref.counts.astype("<u4"), 4)
    cseg, loc = b.struct(seg, pbase + 1, 0, 1)
    b.composite_list(cseg, loc, 0, 3, 0)
    flags = int(sk.concatenated) | (int(sk.noncanonical) << 1) | (int(sk.preserve_case) << 2)
    b.a.put(seg, root + 0, sk.kmer_size | (sk.window_size << 32))
    b.a.put(seg, root + 1, sk.sketch_size | (flags << 32))
    err_bits = struct.unpack("<I", struct.pack("<f", sk.error))[0]
    b.a.put(seg, root + 2, err_bits | ((sk.hash_seed ^ 42) << 32))
    b.text(seg, pbase + 2, sk.alphabet)
    segs = b.a.segs
    hdr = struct.pack("<I", len(segs) - 1) + b"".join(struct.pack("<I", len(s) // 8) for s in segs)
    if len(hdr) % 8:
        hdr += b"\0\0\0\0"
    return hdr + b"".join(bytes(s) for s in segs)


def write_msh(path, sk: SketchFile) -> None:
    with open(path, "wb") as fh:
        fh.write(msh_bytes(sk))


class _Reader:
    def __init__(self, data: bytes):
        nseg = struct.unpack_from("<I", data, 0)[0] + 1
        sizes = struct.unpack_from("<%dI" % nseg, data, 4)
        off = (4 + 4 * nseg + 7) // 8 * 8
        self.segs = []
        for s in sizes:
            self.segs.append(data[off: off + 8 * s])
            off += 8 * s

    def word(self, seg, w):
        return struct.unpack_from("<Q", self.segs[seg], w * 8)[0]

    def resolve(self, seg, w):
        """Follow the pointer at (seg, w) -> (kind, seg, target_word, hi32) or None."""
        p = self.word(seg, w)
        if p == 0:
            return None
        kind = p & 3
        if kind == 2:
            if p & 4:
                raise ValueError("double-far pointers not supported")
            seg, w = p >> 32, (p >> 3) & 0x1FFFFFFF
            p = self.word(seg, w)
            kind = p & 3
        off = (p >> 2) & 0x3FFFFFFF
        if off & 0x20000000:
            off -= 0x40000000
        return kind, seg, w + 1 + off, p >> 32

    def text(self, seg, w) -> str:
        r = self.resolve(seg, w)
        if r is None:
            return ""
        _, s, t, hi = r
        n = hi >> 3
        return self.segs[s][t * 8: t * 8 + n - 1].decode("utf-8")

    def prim(self, seg, w, dtype) -> Optional[np.ndarray]:
        r = self.resolve(seg, w)
        if r is None:
            return None
        _, s, t, hi = r
        n = hi >> 3
        return np.frombuffer(self.segs[s], dtype=dtype, count=n, offset=t * 8).copy()


def read_msh(path) -> SketchFile:
    with open(path, "rb") as fh:
        rd = _Reader(fh.read())
    _, seg, root, hi = rd.resolve(0, 0)
    dwords = hi & 0xFFFF
    w0, w1, w2 = rd.word(seg, root), rd.word(seg, root + 1), rd.word(seg, root + 2)
    sk = SketchFile(kmer_size=w0 & 0xFFFFFFFF, sketch_size=w1 & 0xFFFFFFFF)
    sk.window_size = w0 >> 32
    sk.concatenated = bool((w1 >> 32) & 1)
    sk.noncanonical = bool((w1 >> 33) & 1)
    sk.preserve_case = bool((w1 >> 34) & 1)
    sk.error = struct.unpack("<f", struct.pack("<I", w2 & 0xFFFFFFFF))[0]
    sk.hash_seed = (w2 >> 32) ^ 42
    pbase = root + dwords
    sk.alphabet = rd.text(seg, pbase + 2)
    rl = rd.resolve(seg, pbase + 0) or rd.resolve(seg, pbase + 3)
    if rl is not None:
        _, lseg, lw, _ = rl
        lst = rd.resolve(lseg, lw)
        if lst is not None:
            _, eseg, tagw, _ = lst
            tag = rd.word(eseg, tagw)
            count, ed, ep = (tag >> 2) & 0x3FFFFFFF, (tag >> 32) & 0xFFFF, tag >> 48
            for i in range(count):
                e = tagw + 1 + i * (ed + ep)
                length = rd.word(eseg, e + 1) or (rd.word(eseg, e) & 0xFFFFFFFF)
                pp = e + ed
                h64 = rd.prim(eseg, pp + 5, "<u8")
                h32 = rd.prim(eseg, pp + 4, "<u4")
                hashes = h64 if h64 is not None else (h32.astype(np.uint64) if h32 is not None else np.zeros(0, np.uint64))
                counts = rd.prim(eseg, pp + 6, "<u4") if ep > 6 else None
                sk.references.append(Reference(rd.text(eseg, pp + 2), rd.text(eseg, pp + 3), length, hashes, counts))
    return sk


# --------------------------------------------------------------------------- #
# mash dist
# --------------------------------------------------------------------------- #
def compare(ref: np.ndarray, qry: np.ndarray, sketch_size: int, k: int) -> Tuple[int, int, float]:
    ref = np.ascontiguousarray(ref, dtype=np.uint64)
    qry = np.ascontiguousarray(qry, dtype=np.uint64)
    c, d, dist = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_double()
    lib().mo_compare(ref.ctypes.data, len(ref), qry.ctypes.data, len(qry), sketch_size, k,
                     ctypes.byref(c), ctypes.byref(d), ctypes.byref(dist))
    return c.value, d.value, dist.value


def p_value(x: int, len_ref: int, len_qry: int, kmer_space: float, sketch_size: int) -> float:
    """mash CommandDistance pValue(): P[Binomial(sketchSize, r) >= x]."""
    if x == 0:
        return 1.0
    from scipy.stats import binom

    px = 1.0 / (1.0 + kmer_space / len_ref)
    py = 1.0 / (1.0 + kmer_space / len_qry)
    r = px * py / (px + py - px * py)
    return float(binom.sf(x - 1, sketch_size, r))


def dist_text(ref: SketchFile, qry: SketchFile) -> str:
    """`mash dist REF QUERY` stdout: query-major, ref-minor rows."""
    if ref.kmer_size != qry.kmer_size:
        raise ValueError("k-mer sizes differ")
    k = ref.kmer_size
    s = min(ref.sketch_size, qry.sketch_size)
    kmer_space = 4.0 ** k
    rows = []
    for q in qry.references:
        for r in ref.references:
            common, denom, d = compare(r.hashes, q.hashes, s, k)
            p = p_value(common, r.length, q.length, kmer_space, denom)
            rows.append("%s\t%s\t%s\t%s\t%d/%d\n" % (r.name, q.name, fmt_g(d), fmt_g(p), common, denom))
    return "".join(rows)


# --------------------------------------------------------------------------- #
# mash bounds
# --------------------------------------------------------------------------- #
_SKETCH_SIZES = (100, 500, 1000, 5000, 10000, 50000, 100000, 500000, 1000000)
_DISTS = (0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.35, 0.4)


def bounds_text(k: int, prob: float) -> str:
    """mash CommandBounds::run(): for each sketch size and distance, the smallest
    x with BinomialCDF(x; s, j(d)) > (1-p)/2, converted back to a distance error."""
    from scipy.stats import binom

    q2 = (1.0 - prob) / 2.0
    out = ["", "Parameters (run with -h for details):", "   k:   %d" % k, "   p:   %s" % fmt_g(prob), ""]
    for cont in (0, 1):
        out.append("\tScreen distance" if cont else "\tMash distance")
        out.append("Sketch" + "".join("\t" + fmt_g(d) for d in _DISTS))
        for s in _SKETCH_SIZES:
            row = str(s)
            for d in _DISTS:
                m2j = (1.0 - d) ** k if cont else 1.0 / (2.0 * math.exp(k * d) - 1.0)
                # smallest x in [0, s) with cdf(x) > q2, else s
                x = int(binom.ppf(q2, s, m2j))
                x = max(x - 2, 0)
                while x < s and not (binom.cdf(x, s, m2j) > q2):
                    x += 1
                je = x / s
                if cont:
                    j2m = 1.0 - je ** (1.0 / k)
                else:
                    j2m = math.inf if je == 0 else -1.0 / k * math.log(2.0 * je / (1.0 + je))
                row += "\t" + fmt_g(j2m - d)
            out.append(row)
        out.append("")
    return "\n".join(out) + "\n"
