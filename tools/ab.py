"""A/B timing of library variants on one GPU: every variant is a libmhx build (auriclass_amd/lib_variants/<name>.so,
made by tools/build_variants.sh); each round runs every variant once, in its own process (MHX_LIB selects the
library), on the same synthetic reads; medians over rounds are compared.  Every run also checks its sketch against
the oracle on a prefix, so a fast-but-wrong variant shows up as PARITY FAIL.

    python tools/ab.py [--reads N] [--rounds R] [--k K --s S --m M] name1 name2 ...   (name "base" = lib/libmhx.so)
"""
import argparse
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def child(args):
    import numpy as np
    import torch

    sys.path.insert(0, str(ROOT))
    from auriclass_amd import engine, synth

    engine.init(0)
    g = synth.make_genome(12_000_000, 42)
    fq = synth.make_fastq(g, args.reads, 150, 43, device="cuda")
    torch.cuda.synchronize()
    sk = engine.Sketcher(args.k, args.s, args.m, expected_bytes=fq.numel())
    engine.set_profiling(True)
    ms, wall = [], []
    import time
    for i in range(args.iters + 2):
        t0 = time.perf_counter()
        sk.reset(); sk.push_device(fq.data_ptr(), fq.numel(), engine.FMT_FASTQ4); h, _ = sk.finish()
        wall.append((time.perf_counter() - t0) * 1e3)
        ms.append(sk.stats()["hash_ms"])
    st = sk.stats()
    out = {"kernel_ms": float(np.median(ms[2:])), "step_ms": float(np.median(wall[2:])), "kmers": st["kmers"], "launches": st["launches"]}
    if args.parity:
        from oracle import mash_oracle as mo
        n = min(args.reads, 300_000)
        rb = synth.record_bytes(150)
        sk2 = engine.Sketcher(args.k, args.s, args.m, expected_bytes=n * rb)
        sk2.push_device(fq.data_ptr(), n * rb, engine.FMT_FASTQ4)
        got, _ = sk2.finish()
        ref = mo.Sketcher(args.k, args.s, args.m)
        ref.add_fastx(fq[: n * rb].cpu().numpy().tobytes())
        want, _ = ref.finish()
        out["parity"] = bool(np.array_equal(got, want))
    print("ABRESULT " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--s", type=int, default=1000)
    ap.add_argument("--m", type=int, default=1)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--parity", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    res = {n: [] for n in args.names}
    par = {}
    for rd in range(args.rounds):
        for n in args.names:
            lib = ROOT / "auriclass_amd" / "lib" / "libmhx.so" if n == "base" else ROOT / "auriclass_amd" / "lib_variants" / f"{n}.so"
            env = dict(os.environ, MHX_LIB=str(lib))
            cmd = [sys.executable, __file__, "--child", "--reads", str(args.reads), "--iters", str(args.iters), "--k", str(args.k),
                   "--s", str(args.s), "--m", str(args.m)] + (["--parity"] if rd == 0 else [])
            p = subprocess.run(cmd, env=env, capture_output=True, text=True)
            line = [x for x in p.stdout.splitlines() if x.startswith("ABRESULT ")]
            if p.returncode != 0 or not line:
                print(f"{n}: FAILED rc={p.returncode}\n{p.stdout[-500:]}\n{p.stderr[-1500:]}", flush=True)
                res[n].append(None)
                continue
            r = json.loads(line[0][9:])
            res[n].append(r)
            if "parity" in r:
                par[n] = r["parity"]
            print(f"round {rd} {n:24s} kernel {r['kernel_ms']:.3f} ms  step {r['step_ms']:.3f} ms  launches {r['launches']}" +
                  (f"  parity {'ok' if r['parity'] else 'FAIL'}" if "parity" in r else ""), flush=True)
    import statistics
    base = None
    print("---- medians over rounds ----")
    for n in args.names:
        ok = [r for r in res[n] if r]
        if not ok:
            print(f"{n:24s} no data")
            continue
        km = statistics.median(r["kernel_ms"] for r in ok)
        sm = statistics.median(r["step_ms"] for r in ok)
        if base is None:
            base = (km, sm)
        print(f"{n:24s} kernel {km:.3f} ms ({100 * (km / base[0] - 1):+.1f}%)  step {sm:.3f} ms ({100 * (sm / base[1] - 1):+.1f}%)  "
              f"parity {'ok' if par.get(n) else 'FAIL' if n in par else '?'}")


if __name__ == "__main__":
    main()
