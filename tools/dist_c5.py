#!/usr/bin/env python3
"""BASELINE.json config 5: batched all-vs-refs `mash dist` arithmetic, 1024 query sketches x 24
reference sketches, s = 50 000, on one GPU.  Synthetic sketches per SURVEY.md 8(d): refs are
sorted unique u64 lists, query i = ref (i mod 24) with a fraction f_i in [0, 0.6] of its hashes
replaced by fresh uniform values.  Prints one JSON line: pairs/s and GB/s of algorithmic bytes.  (The
parity of the same batch against the CPU oracle is a test: tests/test_gpu_full_size.py.)"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from auriclass_amd import engine  # noqa: E402


def make_c5(nq=1024, nr=24, s=50_000, seed=1000, device="cuda"):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    hi = (1 << 63) - 1

    def fresh(n):
        return torch.randint(0, hi, (n,), dtype=torch.int64, device=device, generator=g)

    base = fresh(s)
    refs = []
    for j in range(nr):                       # refs share most of a common base (clade-like)
        keep = torch.rand(s, device=device, generator=g) >= (0.002 * (j + 1) if j < 11 else 0.5 + 0.03 * j)
        v = torch.where(keep, base, fresh(s))
        refs.append(torch.sort(torch.unique(v))[0])
    R = torch.zeros((nr, s), dtype=torch.int64, device=device)
    r_len = torch.zeros(nr, dtype=torch.int32, device=device)
    for j, v in enumerate(refs):
        R[j, :len(v)] = v
        r_len[j] = len(v)
    Q = torch.zeros((nq, s), dtype=torch.int64, device=device)
    q_len = torch.zeros(nq, dtype=torch.int32, device=device)
    for i in range(nq):
        src = refs[i % nr]
        f = 0.6 * (i / max(1, nq - 1))
        keep = torch.rand(len(src), device=device, generator=g) >= f
        v = torch.sort(torch.unique(torch.where(keep, src, fresh(len(src)))))[0]
        Q[i, :len(v)] = v
        q_len[i] = len(v)
    return Q, q_len, R, r_len


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--nr", type=int, default=24)
    ap.add_argument("--s", type=int, default=50_000)
    ap.add_argument("--k", type=int, default=27)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    engine.init(0)
    Q, q_len, R, r_len = make_c5(args.nq, args.nr, args.s)
    pairs = args.nq * args.nr
    common = torch.zeros(pairs, dtype=torch.int32, device="cuda")
    denom = torch.zeros(pairs, dtype=torch.int32, device="cuda")
    dist = torch.zeros(pairs, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ms = []
    for _ in range(args.reps + 1):
        ms.append(engine.dist_batch_device(Q.data_ptr(), q_len.data_ptr(), args.nq, R.data_ptr(), r_len.data_ptr(), args.nr,
                                           args.s, args.k, args.s, common.data_ptr(), denom.data_ptr(), dist.data_ptr()))
    t0 = time.perf_counter()
    for _ in range(args.reps):
        engine.dist_batch_device(Q.data_ptr(), q_len.data_ptr(), args.nq, R.data_ptr(), r_len.data_ptr(), args.nr,
                                 args.s, args.k, args.s, common.data_ptr(), denom.data_ptr(), dist.data_ptr())
    wall = (time.perf_counter() - t0) / args.reps
    kernel_ms = float(np.median(ms[1:]))
    alg_bytes = 8 * int(q_len.sum().item() + r_len.sum().item())
    out = {"config": f"{args.nq} queries x {args.nr} refs, s={args.s}, k={args.k}", "pairs": pairs,
           "kernel_ms": round(kernel_ms, 4), "wall_ms_per_call": round(wall * 1e3, 3), "pairs_per_s": round(pairs / (kernel_ms / 1e3)),
           "algorithmic_bytes": alg_bytes, "achieved_GBps": round(alg_bytes / (kernel_ms / 1e3) / 1e9, 2), "hbm_peak_GBps": 8000.0,
           "roofline_frac": round(alg_bytes / (kernel_ms / 1e3) / 1e9 / 8000.0, 5)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
