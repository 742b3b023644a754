"""Wall time of the file-level calls on BASELINE.json config 2 (12 Mb FASTA assembly in 20 contigs, sketched and compared
with a 24-reference set; /root/reference/auriclass/classes.py:696-713, 92-104): one assembly plain and gzipped, and the
24 x 12 Mb reference set in one `mash sketch`-shaped call.  Files live in /dev/shm (page cache), as after a first read."""
import gzip
import os
import statistics
import sys
import tempfile
import time

sys.path.insert(0, ".")
from auriclass_amd import engine, synth  # noqa: E402

engine.init(0)
g = synth.make_genome(12_000_000, 42)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
fa = os.path.join(d, "asm.fasta")
open(fa, "wb").write(synth.genome_fasta(g, 20))
fagz = fa + ".gz"
gzip.open(fagz, "wb", compresslevel=6).write(open(fa, "rb").read())
refs = []
for i in range(24):
    p = os.path.join(d, f"ref{i:02d}.fasta")
    open(p, "wb").write(synth.genome_fasta(synth.mutate(g, 0.0005 * (1 + i), 100 + i), 20, name=f"ref{i}"))
    refs.append(p)


def med(fn, n):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3, min(ts) * 1e3


for k, s in ((21, 1000), (27, 50000)):
    a_msh, r_msh = os.path.join(d, "a.msh"), os.path.join(d, "r.msh")
    for path in (fa, fagz):
        m, lo = med(lambda: engine.sketch_files([path], k, s, a_msh), 9)
        print(f"{os.path.basename(path):14s} k={k} s={s}: sketch {m:.2f} ms median, {lo:.2f} min "
              f"({12 / m:.2f} Gbases/s file-inclusive)", flush=True)
    m, lo = med(lambda: engine.sketch_files(refs, k, s, r_msh), 3)
    print(f"24 x 12 Mb refs k={k} s={s}: sketch {m:.1f} ms median, {lo:.1f} min ({m / 24:.2f} ms per file)", flush=True)
    engine.sketch_files([fa], k, s, a_msh)
    m, lo = med(lambda: engine.dist_files(r_msh, a_msh), 9)
    print(f"dist, 1 query x 24 refs k={k} s={s}: {m:.2f} ms median, {lo:.2f} min (kernels of the last call: {engine.load().mhx_last_dist_kernel_ms():.3f} ms)", flush=True)
for p in refs + [fa, fagz]:
    os.unlink(p)
