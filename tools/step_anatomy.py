"""Where a bench step spends its time outside the tile kernel: reset / push / finish timed separately (each followed by
a stream sync), against the HIP-event time of the tile kernels alone.   python tools/step_anatomy.py [--k K --s S --m M]"""
import argparse, sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auriclass_amd import engine, synth

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10_000_000)
ap.add_argument("--k", type=int, default=21); ap.add_argument("--s", type=int, default=1000); ap.add_argument("--m", type=int, default=1)
a = ap.parse_args()
engine.init(0)
g = synth.make_genome(12_000_000, 42)
fq = synth.make_fastq(g, a.reads, 150, 43, device="cuda")
torch.cuda.synchronize()
sk = engine.Sketcher(a.k, a.s, a.m, expected_bytes=fq.numel())
def t(f):
    t0 = time.perf_counter(); f(); sk.sync(); return (time.perf_counter() - t0) * 1e3
rows = []
for it in range(8):
    r = t(sk.reset)
    p = t(lambda: sk.push_device(fq.data_ptr(), fq.numel(), engine.FMT_FASTQ4))
    f = t(sk.finish)
    rows.append((r, p, f))
rows = np.array(rows[3:])
engine.set_profiling(True)
sk.reset(); sk.push_device(fq.data_ptr(), fq.numel(), engine.FMT_FASTQ4); sk.finish()
st = sk.stats()
engine.set_profiling(False)
whole = []
for it in range(6):
    t0 = time.perf_counter(); sk.reset(); sk.push_device(fq.data_ptr(), fq.numel(), engine.FMT_FASTQ4); sk.finish(); whole.append((time.perf_counter() - t0) * 1e3)
print(f"k={a.k} s={a.s} m={a.m}: reset {np.median(rows[:,0]):.3f} ms  push {np.median(rows[:,1]):.3f} ms  finish {np.median(rows[:,2]):.3f} ms  "
      f"| tile kernels {st['hash_ms']:.3f} ms over {st['launches']} launches | whole step {np.median(whole[2:]):.3f} ms")
