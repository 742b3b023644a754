"""C3 beyond the headline metric: the same 10 M x 150 bp reads with the host-to-device copy
inside the timed region, and through the file-level call (plain file in /dev/shm; a 3 M-read .gz sample for
the inflate-bound case)."""
import gzip, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auriclass_amd import engine, synth
engine.init(0)
g = synth.make_genome(12_000_000, 42)
n = 10_000_000
fq = synth.make_fastq(g, n, 150, 43, device="cuda")
torch.cuda.synchronize()
host = fq.cpu().numpy()
bases = n * 150
sk = engine.Sketcher(21, 1000, 1, expected_bytes=host.size)
for name, fn in (("device-resident", lambda: sk.push_device(fq.data_ptr(), fq.numel(), engine.FMT_FASTQ4)),
                 ("H2D-inclusive (pageable numpy buffer)", lambda: sk.push_host(host, engine.FMT_FASTQ4))):
    ts = []
    for _ in range(4):
        sk.reset(); t0 = time.perf_counter(); fn(); h, _ = sk.finish(); ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    print(f"{name:42s} {1e3*t:8.1f} ms  {bases/t/1e9:7.2f} Gbases/s")
sk.close()
d = tempfile.mkdtemp(dir="/dev/shm")
p = os.path.join(d, "reads.fq"); host.tofile(p)
for k, s, m in ((21, 1000, 1), (27, 50000, 3)):
    ts = []
    for _ in range(2):
        t0 = time.perf_counter(); engine.sketch_files([p], k, s, os.path.join(d, "o.msh"), reads=True, min_mult=m); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f"file-inclusive plain FASTQ k={k} s={s} m={m}      {1e3*t:8.1f} ms  {bases/t/1e9:7.2f} Gbases/s")
rb = synth.record_bytes(150)
pg = os.path.join(d, "r1.fq.gz")
with gzip.open(pg, "wb", compresslevel=1) as fh:
    fh.write(host[: 3_000_000 * rb].tobytes())
pg2 = os.path.join(d, "r2.fq.gz")
with gzip.open(pg2, "wb", compresslevel=1) as fh:
    fh.write(host[3_000_000 * rb: 6_000_000 * rb].tobytes())
def gz_cases(label, sets):
    """every case at the default thread budget and at explicit ones (MHX_INGEST_THREADS is read per call); best of 2"""
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count()
    for threads in (None, 16, 32, 64):
        if threads is None:
            os.environ.pop("MHX_INGEST_THREADS", None)
        else:
            os.environ["MHX_INGEST_THREADS"] = str(threads)
        for files in sets:
            ts = []
            for _ in range(2):
                t0 = time.perf_counter(); engine.sketch_files(files, 27, 50000, os.path.join(d, "o.msh"), reads=True, min_mult=3); ts.append(time.perf_counter() - t0)
            t = min(ts)
            nb = len(files) * 3_000_000 * 150
            tl = f"default budget ({usable} usable cores)" if threads is None else f"MHX_INGEST_THREADS={threads}"
            print(f"file-inclusive {len(files)} x 3 M-read {label} (k=27 s=50000 m=3), {tl}: {1e3*t:8.1f} ms  {nb/t/1e9:7.3f} Gbases/s", flush=True)
    os.environ.pop("MHX_INGEST_THREADS", None)


gz_cases(".fq.gz", ([pg], [pg, pg2]))
# the same two files as bgzip writes them (independent 64 KiB members)
import struct, zlib
def bgzf(data, level=1, block=0xFF00):
    out = bytearray()
    for i in range(0, len(data), block):
        chunk = data[i:i + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15); dd = c.compress(chunk) + c.flush()
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(dd) + 25) + dd + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
pb, pb2 = os.path.join(d, "b1.fq.gz"), os.path.join(d, "b2.fq.gz")
open(pb, "wb").write(bgzf(host[: 3_000_000 * rb].tobytes()))
open(pb2, "wb").write(bgzf(host[3_000_000 * rb: 6_000_000 * rb].tobytes()))
gz_cases("BGZF .fq.gz", ([pb], [pb, pb2]))
import shutil; shutil.rmtree(d)
