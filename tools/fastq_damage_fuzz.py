"""Differential fuzz of damaged FASTQ files: engine (file-level call) against the oracle; see DESIGN.md section 6."""
import os, sys, tempfile, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from auriclass_amd import engine, synth
from oracle import mash_oracle as mo
engine.init(0)
d = tempfile.mkdtemp()
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
genome = synth.make_genome(20000, seed=5)
base = synth.make_fastq(genome, 300, 80, seed=6, device="cpu").numpy().tobytes()
special = np.frombuffer(b"\n\n\n@+>\r ANacgt", np.uint8)
diffs = 0; both_err = 0; same = 0
for trial in range(int(os.environ.get("N", "300"))):
    b = bytearray(base)
    for _ in range(int(rng.integers(1, 5))):
        pos = int(rng.integers(0, len(b)))
        op = int(rng.integers(0, 3))
        if op == 0: b[pos] = int(rng.choice(special))
        elif op == 1: del b[pos:pos + int(rng.integers(1, 40))]
        else: b[pos:pos] = bytes(rng.choice(special, size=int(rng.integers(1, 6))))
    data = bytes(b)
    p = os.path.join(d, "f.fq"); open(p, "wb").write(data)
    res = []
    for impl in ("engine", "oracle"):
        try:
            if impl == "engine":
                engine.sketch_files([p], 21, 200, os.path.join(d, "e.msh"), reads=True, min_mult=1)
                r = mo.read_msh(os.path.join(d, "e.msh")).references[0]
            else:
                r = mo.sketch_files([p], 21, 200, reads=True)[0].references[0]
            res.append(("ok", r.hashes.tobytes(), r.comment))
        except Exception as e:
            res.append(("err",))
    if res[0][0] == "err" and res[1][0] == "err": both_err += 1
    elif res[0] == res[1]: same += 1
    else:
        diffs += 1
        if diffs <= 5:
            open(os.path.join(root := os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "fqfuzz_diff_%d.fq" % diffs), "wb").write(data)
            print("DIFF trial", trial, res[0][0], res[1][0], (len(res[0][1]) if res[0][0] == "ok" else None), (len(res[1][1]) if res[1][0] == "ok" else None), (res[0][2] if res[0][0]=="ok" else ""), "|", (res[1][2] if res[1][0]=="ok" else ""), flush=True)
print("same", same, "both refuse", both_err, "DIFFERENT", diffs, flush=True)
