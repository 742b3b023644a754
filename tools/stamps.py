"""Diagnostic: per-phase share of a workgroup's time in a -DMHX_STAMPS build (MHX_LIB=...exp_stamp.so)."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from auriclass_amd import engine, synth
engine.init(0)
g = synth.make_genome(12_000_000, 42)
fq = synth.make_fastq(g, 4_000_000, 150, 43, device="cuda")
torch.cuda.synchronize()
sk = engine.Sketcher(21, 1000, 1, expected_bytes=fq.numel())
for _ in range(2):
    sk.reset(); sk.push_device(fq.data_ptr(), fq.numel(), engine.FMT_FASTQ4); sk.finish()
L = engine.load(); L.mhx_sketcher_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
out = np.zeros(8, np.uint64); L.mhx_sketcher_debug_stamps(sk._h, out.ctypes.data)
names = ["stage", "classify", "nl-scan+lookback", "good map", "runs+worklist", "work loop"]
tot = float(out[:6].sum())
for n, v in zip(names, out[:6]):
    print(f"{n:18s} {int(v):>16d}  {100.0 * float(v) / tot:5.1f}%")
