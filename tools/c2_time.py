"""Wall time of the file-level calls on BASELINE.json config 2 (12 Mb FASTA, 20 contigs)."""
import sys, time, tempfile, os
sys.path.insert(0, ".")
from auriclass_amd import engine, synth
engine.init(0)
g = synth.make_genome(12_000_000, 42)
d = tempfile.mkdtemp()
fa = os.path.join(d, "asm.fasta"); open(fa, "wb").write(synth.genome_fasta(g, 20))
import gzip
fagz = fa + ".gz"; gzip.open(fagz, "wb", compresslevel=6).write(open(fa, "rb").read())
for path in (fa, fagz):
    for k, s in ((21, 1000), (27, 50000)):
        engine.sketch_files([path], k, s, os.path.join(d, "a.msh"))
        t0 = time.perf_counter()
        for _ in range(3):
            engine.sketch_files([path], k, s, os.path.join(d, "a.msh"))
        t1 = time.perf_counter()
        engine.sketch_files([path, path, path, path], k, s, os.path.join(d, "r.msh"))
        t2 = time.perf_counter()
        for _ in range(3):
            engine.dist_files(os.path.join(d, "r.msh"), os.path.join(d, "a.msh"))
        t3 = time.perf_counter()
        print(f"{os.path.basename(path):14s} k={k} s={s}: sketch {1e3*(t1-t0)/3:.1f} ms ({12/( (t1-t0)/3)/1e3:.2f} Gbases/s file-inclusive)  dist(4 refs) {1e3*(t3-t2)/3:.2f} ms")
