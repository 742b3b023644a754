"""Where one `.fq.gz` file-level call spends its wall time (MHX_INGEST_DEBUG=1: the producer thread's own account;
the rest is set-up and tail): 3 M reads, k=27 s=50000 m=3, second call of two (buffers and pinned pool warm)."""
import gzip, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auriclass_amd import engine, synth
engine.init(0)
g = synth.make_genome(12_000_000, 42)
fq = synth.make_fastq(g, 3_000_000, 150, 43, device="cpu").numpy()
d = tempfile.mkdtemp(dir="/dev/shm")
p = os.path.join(d, "r1.fq.gz")
with gzip.open(p, "wb", compresslevel=1) as fh:
    fh.write(fq.tobytes())
print("compressed bytes", os.path.getsize(p), "inflated", fq.size, flush=True)
for rep in range(3):
    os.environ["MHX_INGEST_DEBUG"] = "1" if rep == 2 else ""
    if rep < 2:
        os.environ.pop("MHX_INGEST_DEBUG")
    t0 = time.perf_counter()
    engine.sketch_files([p], 27, 50000, os.path.join(d, "o.msh"), reads=True, min_mult=3)
    t = time.perf_counter() - t0
    print(f"call {rep}: {1e3 * t:.1f} ms  {3_000_000 * 150 / t / 1e9:.2f} Gbases/s", flush=True)
import shutil; shutil.rmtree(d)
