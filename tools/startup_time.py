"""Start-up cost of a one-sample process: library load, device initialisation, first (tiny) sketch, first distance call."""
import os, sys, time
t0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auriclass_amd import engine
t1 = time.perf_counter()
engine.load()
t2 = time.perf_counter()
engine.init(0)
t3 = time.perf_counter()
import numpy as np
rng = np.random.default_rng(1)
seq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=200_000))
sk = engine.Sketcher(27, 50000, 1, expected_bytes=len(seq) + 1)
sk.push_host(seq + b"\n", engine.FMT_SEQ)
h, _ = sk.finish()
t4 = time.perf_counter()
sk2 = engine.Sketcher(21, 1000, 1, expected_bytes=len(seq) + 1)
sk2.push_host(seq + b"\n", engine.FMT_SEQ)
sk2.finish()
t5 = time.perf_counter()
q = np.zeros((1, 1008), np.uint64); q[0, :1000] = np.sort(rng.integers(0, 1 << 60, 1000, dtype=np.uint64))
engine.dist_batch(q, np.array([1000], np.uint32), q, np.array([1000], np.uint32), 21, 1000)
t6 = time.perf_counter()
print(f"import engine {1e3*(t1-t0):.1f} ms | dlopen libmhx {1e3*(t2-t1):.1f} ms | mhx_init (HIP runtime, device, streams) {1e3*(t3-t2):.1f} ms | "
      f"first sketch (k=27: kernel code load + table) {1e3*(t4-t3):.1f} ms | first sketch at another k {1e3*(t5-t4):.1f} ms | first dist {1e3*(t6-t5):.1f} ms")
