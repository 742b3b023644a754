"""ctypes binding of libmhx.so (include/mhx.h): the GPU engine that stands where AuriClass
shells out to `mash` (/root/reference/auriclass/classes.py:92-104, 305-318, 576-596, 696-713).

There is no CPU fallback: if the library is missing or no HIP device is usable, the calls
raise :class:`EngineError` instead of computing anything another way.
"""
from __future__ import annotations

import ctypes
import os
import re
import subprocess
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MHX_LIB", _PKG / "lib" / "libmhx.so"))  # MHX_LIB: experiment builds
HEADER_PATH = _PKG.parent / "include" / "mhx.h"

MHX_OK = 0
MHX_E_NO_DEVICE = -1
MHX_E_ARG = -2
MHX_E_IO = -3
MHX_E_NO_RECORDS = -4
MHX_E_FORMAT = -5
MHX_E_HIP = -6
MHX_E_CAPACITY = -7
MHX_E_MISMATCH = -8
MHX_E_INTERNAL = -9

FMT_SEQ = 0
FMT_FASTQ4 = 1


class EngineError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"mhx error {code}: {message}")
        self.code = code
        self.message = message


class NoRecordsError(EngineError):
    """mash: 'ERROR: Did not find fasta records in ...'"""


def build(force: bool = False) -> Path:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-s", "-C", str(_PKG / "csrc"), "clean"], check=True)
    subprocess.run(["make", "-s", "-j4", "-C", str(_PKG / "csrc")], check=True)
    return LIB_PATH


_lib: Optional[ctypes.CDLL] = None


def declared_symbols() -> List[str]:
    """Entry points declared in include/mhx.h."""
    text = HEADER_PATH.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mhx_[a-z0-9_]+)\s*\(", text)))


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise EngineError(MHX_E_NO_DEVICE, f"{LIB_PATH} not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                                           "there is no CPU fallback")
    L = ctypes.CDLL(str(LIB_PATH))
    c = ctypes
    u64p, u32p = c.POINTER(c.c_uint64), c.POINTER(c.c_uint32)
    L.mhx_init.argtypes = [c.c_int]
    L.mhx_last_error.restype = c.c_char_p
    L.mhx_version.restype = c.c_char_p
    L.mhx_device_name.argtypes = [c.c_char_p, c.c_size_t]
    L.mhx_sketch_files.argtypes = [c.POINTER(c.c_char_p), c.c_int, c.c_int, c.c_uint32, c.c_int, c.c_uint32, c.c_char_p,
                                   c.c_char_p, c.c_size_t, c.POINTER(c.c_size_t), c.POINTER(c.c_double)]
    L.mhx_dist_files.argtypes = [c.c_char_p, c.c_char_p, c.c_char_p, c.c_size_t, c.POINTER(c.c_size_t)]
    L.mhx_bounds.argtypes = [c.c_int, c.c_double, c.c_char_p, c.c_size_t, c.POINTER(c.c_size_t)]
    L.mhx_fasta_total_bases.argtypes = [c.c_char_p, u64p]
    L.mhx_sniff_fastq.argtypes = [c.c_char_p]
    L.mhx_sniff_fasta.argtypes = [c.c_char_p]
    L.mhx_fastq_tail_complete.argtypes = [c.c_char_p, c.c_size_t]
    L.mhx_fastq_tail_complete.restype = c.c_int
    L.mhx_sketcher_create.argtypes = [c.c_int, c.c_uint32, c.c_uint32, c.c_uint64, c.POINTER(c.c_void_p)]
    L.mhx_sketcher_create_scaled.argtypes = [c.c_int, c.c_uint32, c.c_uint32, c.c_uint64, c.c_uint32, c.POINTER(c.c_void_p)]
    L.mhx_sketcher_destroy.argtypes = [c.c_void_p]
    L.mhx_sketcher_destroy.restype = None
    L.mhx_sketcher_reset.argtypes = [c.c_void_p]
    L.mhx_sketcher_push_device.argtypes = [c.c_void_p, c.c_void_p, c.c_uint64, c.c_int]
    L.mhx_sketcher_push_host.argtypes = [c.c_void_p, c.c_void_p, c.c_uint64, c.c_int]
    L.mhx_sketcher_sync.argtypes = [c.c_void_p]
    L.mhx_sketcher_finish.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, u32p]
    L.mhx_sketcher_stats.argtypes = [c.c_void_p, c.c_void_p]
    L.mhx_sketcher_record_count.argtypes = [c.c_void_p, c.c_void_p]
    L.mhx_set_profiling.argtypes = [c.c_int]
    L.mhx_stream.restype = c.c_void_p
    L.mhx_sketcher_threshold.argtypes = [c.c_void_p, u64p]
    L.mhx_sketcher_export.argtypes = [c.c_void_p, c.c_uint64, c.c_void_p, c.c_void_p, c.c_uint32, u32p]
    L.mhx_merge_partials.argtypes = [c.c_void_p, c.c_void_p, c.c_uint64, c.c_uint32, c.c_uint32, c.c_void_p, c.c_void_p, u32p]
    L.mhx_merge_shard_partials.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_uint32, c.c_int, c.c_uint32, c.c_uint32,
                                           c.c_void_p, c.c_void_p, u32p]
    L.mhx_dist_batch.argtypes = [c.c_void_p, c.c_void_p, c.c_uint32, c.c_void_p, c.c_void_p, c.c_uint32, c.c_uint32,
                                 c.c_int, c.c_uint32, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int]
    L.mhx_last_dist_kernel_ms.restype = c.c_double
    L.mhx_last_dist_fallback_blocks.restype = c.c_int
    L.mhx_p_value.argtypes = [c.c_uint64, c.c_uint64, c.c_uint64, c.c_int, c.c_uint64]
    L.mhx_p_value.restype = c.c_double
    L.mhx_msh_write.argtypes = [c.c_char_p, c.c_int, c.c_uint32, c.c_uint32, c.POINTER(c.c_char_p), c.POINTER(c.c_char_p),
                                u64p, c.POINTER(u64p), u32p]
    L.mhx_sketcher_export_slab.argtypes = [c.c_void_p, c.c_void_p, c.c_uint32]
    L.mhx_sketcher_export_begin.argtypes = [c.c_void_p, c.c_void_p]
    L.mhx_sketcher_export_pack.argtypes = [c.c_void_p, c.c_void_p, c.c_uint64]
    L.mhx_sketcher_export_into.argtypes = [c.c_void_p, c.c_void_p, c.c_uint64, c.c_void_p]
    L.mhx_sketcher_merge_gathered.argtypes = [c.c_void_p, c.c_void_p, c.c_uint32, c.c_uint64, c.c_uint32, c.c_void_p, c.c_void_p, u32p, u64p]
    L.mhx_sketcher_merge_slabs.argtypes = [c.c_void_p, c.c_void_p, c.c_int, c.c_uint32, c.c_uint64, c.c_void_p, c.c_uint32,
                                           c.c_void_p, c.c_void_p, u32p]
    L.mhx_gunzip_buffer.argtypes = [c.c_char_p, c.c_size_t, c.c_void_p, c.c_size_t, c.POINTER(c.c_size_t)]
    L.mhx_gunzip_buffer_mt.argtypes = [c.c_char_p, c.c_size_t, c.c_void_p, c.c_size_t, c.POINTER(c.c_size_t), c.c_int]
    _lib = L
    return L


def _check(rc: int) -> None:
    if rc == MHX_OK:
        return
    msg = load().mhx_last_error().decode("utf-8", "replace")
    if rc == MHX_E_NO_RECORDS:
        raise NoRecordsError(rc, msg)
    raise EngineError(rc, msg)


_initialised = False


def init(device: Optional[int] = None) -> None:
    """Select the GPU (default: LOCAL_RANK or 0). Raises EngineError when none is usable."""
    global _initialised
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if not _initialised else -1
    _check(load().mhx_init(device))
    _initialised = True


def device_name() -> str:
    init()
    buf = ctypes.create_string_buffer(256)
    _check(load().mhx_device_name(buf, len(buf)))
    return buf.value.decode()


def stream_handle() -> int:
    init()
    return load().mhx_stream() or 0


def _text_call(fn, *args, guess: int = 1 << 16) -> str:
    """The C ABI's two-call text pattern (size, then fill) with the first call already carrying a buffer: a text that
    fits it -- the 24 rows of an AuriClass `mash dist`, the bounds table -- costs ONE call (the library does the whole
    work in either call); a longer one reports its size (MHX_E_CAPACITY, `need`) and is fetched by a second call."""
    need = ctypes.c_size_t(0)
    buf = ctypes.create_string_buffer(guess)
    rc = fn(*args, buf, guess, ctypes.byref(need))
    if rc == MHX_E_CAPACITY and need.value > guess:
        buf = ctypes.create_string_buffer(need.value)
        rc = fn(*args, buf, need.value, ctypes.byref(need))
    _check(rc)
    return buf.value.decode("utf-8", "surrogateescape")   # names inside a .msh are arbitrary bytes


# --------------------------------------------------------------------------- file level
def sketch_files(paths: Sequence, k: int, s: int, out_msh, reads: bool = False, min_mult: int = 1) -> Tuple[str, float]:
    """`mash sketch [-r -m M] -o OUT -k K -s S paths...` -> (stderr text, estimated genome size)."""
    init()
    L = load()
    arr = (ctypes.c_char_p * len(paths))(*[os.fsencode(str(p)) for p in paths])
    need = ctypes.c_size_t(0)
    est = ctypes.c_double(0.0)
    cap = 4096 + sum(len(str(p)) for p in paths) * 2 + len(str(out_msh))
    buf = ctypes.create_string_buffer(cap)
    rc = L.mhx_sketch_files(arr, len(paths), k, s, int(reads), min_mult, os.fsencode(str(out_msh)), buf, cap,
                            ctypes.byref(need), ctypes.byref(est))
    if rc == MHX_E_NO_RECORDS:
        raise NoRecordsError(rc, L.mhx_last_error().decode())
    _check(rc)
    return buf.value.decode("utf-8", "surrogateescape"), est.value


def dist_files(ref_msh, qry_msh) -> str:
    """`mash dist REF QUERY` stdout."""
    init()
    return _text_call(load().mhx_dist_files, os.fsencode(str(ref_msh)), os.fsencode(str(qry_msh)))


def bounds(k: int, p: float) -> str:
    """`mash bounds -k K -p P` stdout (host arithmetic; needs no GPU)."""
    return _text_call(load().mhx_bounds, k, p)


def fasta_total_bases(path) -> int:
    v = ctypes.c_uint64(0)
    _check(load().mhx_fasta_total_bases(os.fsencode(str(path)), ctypes.byref(v)))
    return v.value


def fastq_tail_complete(tail: bytes) -> bool:
    """Is the last record of a 4-line FASTQ complete (its last bytes are enough)?  See mhx_fastq_tail_complete."""
    return bool(load().mhx_fastq_tail_complete(tail, len(tail)))


def sniff_fastq(path) -> bool:
    rc = load().mhx_sniff_fastq(os.fsencode(str(path)))
    if rc < 0:
        _check(rc)
    return bool(rc)


def sniff_fasta(path) -> bool:
    rc = load().mhx_sniff_fasta(os.fsencode(str(path)))
    if rc < 0:
        _check(rc)
    return bool(rc)


def p_value(common: int, len_ref: int, len_qry: int, k: int, denom: int) -> float:
    return load().mhx_p_value(common, len_ref, len_qry, k, denom)


def msh_write(path, k: int, s: int, names: Sequence[str], comments: Sequence[str], lengths: Sequence[int],
              hashes: Sequence[np.ndarray]) -> None:
    n = len(names)
    c = ctypes
    arrs = [np.ascontiguousarray(h, dtype=np.uint64) for h in hashes]
    u64p = c.POINTER(c.c_uint64)
    _check(load().mhx_msh_write(
        os.fsencode(str(path)), k, s, n,
        (c.c_char_p * n)(*[x.encode() for x in names]), (c.c_char_p * n)(*[x.encode() for x in comments]),
        (c.c_uint64 * n)(*lengths), (u64p * n)(*[a.ctypes.data_as(u64p) for a in arrs]),
        (c.c_uint32 * n)(*[len(a) for a in arrs])))


# --------------------------------------------------------------------------- buffer level
class Sketcher:
    """Device sketch accumulator (one reference)."""

    def __init__(self, k: int, s: int, min_mult: int = 1, expected_bytes: int = 0, budget_scale: int = 1):
        init()
        self.k, self.s, self.m = k, s, max(1, min_mult)
        h = ctypes.c_void_p()
        _check(load().mhx_sketcher_create_scaled(k, s, self.m, expected_bytes, budget_scale, ctypes.byref(h)))
        self._h = h
        self._keep: list = []   # owners of pushed device memory, released at the next settling call

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.mhx_sketcher_destroy(h)

    def __del__(self):
        try:                      # the module globals may already be gone at interpreter shutdown
            self.close()
        except Exception:
            pass

    def reset(self) -> None:
        _check(load().mhx_sketcher_reset(self._h))
        self._keep.clear()

    def push_device(self, ptr: int, nbytes: int, fmt: int, keep=None) -> None:
        """Feeds `nbytes` of device memory at `ptr` (asynchronous, on the engine's stream).
        LIFETIME: the bytes must stay valid AND unchanged until sync(), finish(), an export or reset() has returned --
        a synchronisation of the stream alone is not enough: FASTQ spans whose reads are longer than ~2.7 kb are read
        a second time by a repair pass that those calls start (include/mhx.h).  `keep`: any object (e.g. the torch
        tensor that owns the memory) to be referenced by this sketcher until then, so that dropping the caller's last
        reference cannot hand the memory to someone else in between."""
        _check(load().mhx_sketcher_push_device(self._h, ctypes.c_void_p(ptr), nbytes, fmt))
        if keep is not None:
            self._keep.append(keep)

    def push_host(self, data, fmt: int) -> None:
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        a = np.ascontiguousarray(a)
        _check(load().mhx_sketcher_push_host(self._h, a.ctypes.data, a.size, fmt))

    def sync(self) -> None:
        _check(load().mhx_sketcher_sync(self._h))
        self._keep.clear()

    def finish(self) -> Tuple[np.ndarray, np.ndarray]:
        hashes = np.zeros(self.s, dtype=np.uint64)
        counts = np.zeros(self.s, dtype=np.uint32)
        n = ctypes.c_uint32(0)
        _check(load().mhx_sketcher_finish(self._h, hashes.ctypes.data, counts.ctypes.data, ctypes.byref(n)))
        self._keep.clear()
        return hashes[:n.value].copy(), counts[:n.value].copy()

    def stats(self) -> dict:
        raw = np.zeros(8, dtype=np.uint64)
        _check(load().mhx_sketcher_stats(self._h, raw.ctypes.data))
        return {"kmers": int(raw[0]), "inserts": int(raw[1]), "lines": int(raw[2]), "flags": int(raw[3]),
                "occupied": int(raw[4]), "hash_ms": float(raw[5:6].view(np.float64)[0]), "launches": int(raw[6]),
                "threshold": int(raw[7])}

    def record_count(self) -> int:
        """FASTQ records pushed so far whose sequence line holds >= k bytes (mash's sequence count)."""
        n = ctypes.c_uint64(0)
        _check(load().mhx_sketcher_record_count(self._h, ctypes.byref(n)))
        return int(n.value)

    def threshold(self) -> int:
        v = ctypes.c_uint64(0)
        _check(load().mhx_sketcher_threshold(self._h, ctypes.byref(v)))
        return v.value

    def export_slab(self, device_ptr: int, cap: int) -> None:
        """Partial result as one device-resident int64 slab [n, T, flags, hashes[cap], counts (u32 pairs)];
        `device_ptr` must hold 3 + cap + cap // 2 int64 words (see include/mhx.h)."""
        _check(load().mhx_sketcher_export_slab(self._h, ctypes.c_void_p(device_ptr), cap))

    def export_begin(self) -> np.ndarray:
        """Sharded path, step 1: compacts this shard's partial result (every (hash, count) <= its threshold) on the
        device and returns the 8-word header [n, T, flags, #(2^64-1), occupied slots, 0, 0, 0] the ranks exchange."""
        hdr = np.zeros(8, dtype=np.uint64)
        _check(load().mhx_sketcher_export_begin(self._h, hdr.ctypes.data))
        return hdr

    def export_pack(self, dst_ptr: int, cap_entries: int) -> None:
        """Step 2: the partial result as one slab [hashes[cap_entries] | counts u32[cap_entries]] at `dst_ptr`
        (device or host memory; cap_entries + cap_entries // 2 eight-byte words)."""
        _check(load().mhx_sketcher_export_pack(self._h, ctypes.c_void_p(dst_ptr), cap_entries))

    def merge_slabs(self, slabs_ptr: int, on_device: bool, n_ranks: int, cap_entries: int, headers: np.ndarray, own_rank: int
                    ) -> Tuple[np.ndarray, np.ndarray]:
        """Step 3: adds the other ranks' gathered slabs to this sketcher's table ON THE DEVICE and extracts the sketch of
        the union (EngineError(MHX_E_CAPACITY) when the partials do not determine it).  The sketcher must be reset()
        before it is pushed to again."""
        headers = np.ascontiguousarray(headers, dtype=np.uint64).reshape(-1)
        assert headers.size == 8 * n_ranks
        hashes = np.zeros(self.s, dtype=np.uint64)
        counts = np.zeros(self.s, dtype=np.uint32)
        n = ctypes.c_uint32(0)
        _check(load().mhx_sketcher_merge_slabs(self._h, ctypes.c_void_p(slabs_ptr), int(on_device), n_ranks, cap_entries,
                                               headers.ctypes.data, own_rank, hashes.ctypes.data, counts.ctypes.data, ctypes.byref(n)))
        return hashes[:n.value].copy(), counts[:n.value].copy()

    def export_into(self, device_ptr: int, cap_entries: int) -> np.ndarray:
        """One-collective form of the exchange (device buffers): the partial result straight into the send slab
        [header8 | hashes[cap_entries] | counts u32[cap_entries]] at `device_ptr`; returns the header."""
        hdr = np.zeros(8, dtype=np.uint64)
        _check(load().mhx_sketcher_export_into(self._h, ctypes.c_void_p(device_ptr), cap_entries, hdr.ctypes.data))
        return hdr

    def merge_gathered(self, slabs_ptr: int, n_ranks: int, cap_entries: int, own_rank: int):
        """Merges the gathered header-carrying slabs on the device.  Returns (hashes, counts, 0), or (None, None, need)
        when some shard holds `need` > cap_entries entries: repeat export_into / all-gather with a larger capacity."""
        hashes = np.zeros(self.s, dtype=np.uint64)
        counts = np.zeros(self.s, dtype=np.uint32)
        n = ctypes.c_uint32(0)
        need = ctypes.c_uint64(0)
        rc = load().mhx_sketcher_merge_gathered(self._h, ctypes.c_void_p(slabs_ptr), n_ranks, cap_entries, own_rank,
                                                hashes.ctypes.data, counts.ctypes.data, ctypes.byref(n), ctypes.byref(need))
        if rc == MHX_E_CAPACITY and need.value:
            return None, None, int(need.value)
        _check(rc)
        return hashes[:n.value].copy(), counts[:n.value].copy(), 0

    def export(self, limit: int) -> Tuple[np.ndarray, np.ndarray]:
        cap = 1 << 16
        while True:
            hashes = np.zeros(cap, dtype=np.uint64)
            counts = np.zeros(cap, dtype=np.uint32)
            n = ctypes.c_uint32(0)
            rc = load().mhx_sketcher_export(self._h, limit, hashes.ctypes.data, counts.ctypes.data, cap, ctypes.byref(n))
            if rc == MHX_E_CAPACITY and n.value > cap:
                cap = n.value + 16
                continue
            _check(rc)
            return hashes[:n.value].copy(), counts[:n.value].copy()


def gunzip(data: bytes, threads: int = 1, size_hint: int = 0) -> bytes:
    """Inflate an in-memory .gz (all members) with the ingest's own DEFLATE decoder (host code, no GPU);
    threads > 1: the first member is decoded by that many threads (mhx_gunzip_buffer_mt)."""
    L = load()
    need = ctypes.c_size_t(0)

    def call(buf, cap):
        if threads > 1:
            return L.mhx_gunzip_buffer_mt(data, len(data), buf, cap, ctypes.byref(need), threads)
        return L.mhx_gunzip_buffer(data, len(data), buf, cap, ctypes.byref(need))

    if size_hint:
        out = ctypes.create_string_buffer(size_hint)
        rc = call(out, size_hint)
        if rc == MHX_OK:
            return out.raw[:need.value]
        if rc != MHX_E_CAPACITY:
            raise EngineError(rc, L.mhx_last_error().decode())
    else:
        rc = call(None, 0)
        if rc:
            raise EngineError(rc, L.mhx_last_error().decode())
    out = ctypes.create_string_buffer(max(1, need.value))
    rc = call(out, need.value)
    if rc:
        raise EngineError(rc, L.mhx_last_error().decode())
    return out.raw[:need.value]


def set_profiling(on: bool) -> None:
    load().mhx_set_profiling(int(on))


def merge_partials(hashes: np.ndarray, counts: np.ndarray, s: int, min_mult: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    oh = np.zeros(s, dtype=np.uint64)
    oc = np.zeros(s, dtype=np.uint32)
    n = ctypes.c_uint32(0)
    _check(load().mhx_merge_partials(hashes.ctypes.data, counts.ctypes.data, hashes.size, s, min_mult,
                                     oh.ctypes.data, oc.ctypes.data, ctypes.byref(n)))
    return oh[:n.value].copy(), oc[:n.value].copy()


def merge_shard_partials(hashes: Sequence[np.ndarray], counts: Sequence[np.ndarray], thresholds: Sequence[int], k: int, s: int,
                         min_mult: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    """Merge of the shards' exports (one array pair and one admission threshold per shard) with the exactness
    rule of the sharded path: EngineError(MHX_E_CAPACITY) when the partials do not determine the union's sketch."""
    assert len(hashes) == len(counts) == len(thresholds) and len(thresholds) > 0
    sizes = np.array([len(h) for h in hashes], dtype=np.uint64)
    allh = np.ascontiguousarray(np.concatenate([np.asarray(h, dtype=np.uint64) for h in hashes]) if len(hashes) else np.zeros(0, np.uint64))
    allc = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.uint32) for x in counts]) if len(counts) else np.zeros(0, np.uint32))
    thr = np.array([int(x) for x in thresholds], dtype=np.uint64)
    oh = np.zeros(s, dtype=np.uint64)
    oc = np.zeros(s, dtype=np.uint32)
    n = ctypes.c_uint32(0)
    _check(load().mhx_merge_shard_partials(allh.ctypes.data, allc.ctypes.data, sizes.ctypes.data, thr.ctypes.data, len(thr), k, s,
                                           min_mult, oh.ctypes.data, oc.ctypes.data, ctypes.byref(n)))
    return oh[:n.value].copy(), oc[:n.value].copy()


def dist_batch(q: np.ndarray, q_len: np.ndarray, r: np.ndarray, r_len: np.ndarray, k: int, s: int
               ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Host arrays in, host arrays out: common, denom, dist of shape [nq, nr]."""
    init()
    q = np.ascontiguousarray(q, dtype=np.uint64)
    r = np.ascontiguousarray(r, dtype=np.uint64)
    assert q.ndim == 2 and r.ndim == 2 and q.shape[1] == r.shape[1]
    q_len = np.ascontiguousarray(q_len, dtype=np.uint32)
    r_len = np.ascontiguousarray(r_len, dtype=np.uint32)
    nq, nr, stride = q.shape[0], r.shape[0], q.shape[1]
    common = np.zeros((nq, nr), dtype=np.uint32)
    denom = np.zeros((nq, nr), dtype=np.uint32)
    dist = np.zeros((nq, nr), dtype=np.float64)
    _check(load().mhx_dist_batch(q.ctypes.data, q_len.ctypes.data, nq, r.ctypes.data, r_len.ctypes.data, nr, stride,
                                 k, s, common.ctypes.data, denom.ctypes.data, dist.ctypes.data, 0))
    return common, denom, dist


def dist_batch_device(q_ptr: int, q_len_ptr: int, nq: int, r_ptr: int, r_len_ptr: int, nr: int, stride: int, k: int, s: int,
                      common_ptr: int, denom_ptr: int, dist_ptr: int) -> float:
    """Device pointers in and out; returns the kernel time in ms (HIP events on the engine stream)."""
    init()
    v = ctypes.c_void_p
    _check(load().mhx_dist_batch(v(q_ptr), v(q_len_ptr), nq, v(r_ptr), v(r_len_ptr), nr, stride, k, s,
                                 v(common_ptr), v(denom_ptr), v(dist_ptr), 1))
    return load().mhx_last_dist_kernel_ms()
