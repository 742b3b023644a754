"""Diagnostic: the same reads as a FASTQ stream (315 B/read) and as a dense sequence stream
(151 B/read, MHX_FMT_SEQ): how much of the kernel time is hashing, how much is the FASTQ parse."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from auriclass_amd import engine, synth
engine.init(0)
g = synth.make_genome(12_000_000, 42)
n = 10_000_000
fq = synth.make_fastq(g, n, 150, 43, device="cuda")
rb = synth.record_bytes(150)
seq = fq.view(n, rb)[:, 11:11 + 151].contiguous().view(-1)     # bases + '\n'
torch.cuda.synchronize()
engine.set_profiling(True)
for name, buf, fmt in (("fastq", fq, engine.FMT_FASTQ4), ("seq", seq, engine.FMT_SEQ)):
    sk = engine.Sketcher(21, 1000, 1, expected_bytes=buf.numel())
    ms = []
    for _ in range(4):
        sk.reset(); sk.push_device(buf.data_ptr(), buf.numel(), fmt); h, _ = sk.finish(); st = sk.stats(); ms.append(st["hash_ms"])
    print(name, "bytes", buf.numel(), "kmers", st["kmers"], "kernel ms", [round(x, 3) for x in ms], "launches", st["launches"], "first hash", int(h[0]))
    sk.close()
