#!/usr/bin/env python3
"""Headline benchmark: Gbases/s sketched (k=21, s=1000) on synthetic FASTQ resident in HBM.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N ...                      (starts N ranks by itself, one per GPU)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one whole sketch job over this rank's reads: reset the device table, run the tile kernels
over the FASTQ bytes, pull the final sorted sketch to host memory; with N > 1 also the cross-rank
exchange + merge with its exactness rule (auriclass_amd.multigpu.sharded_sketch).
  weak scaling (default):      every rank owns --reads reads (10 M x 150 bp, BASELINE.json configs[2]),
                               the answer is the sketch of the union;
  strong scaling (--total-reads T): T reads in all, rank r owns records [r*T/N, (r+1)*T/N)
                               (BASELINE.json configs[3] / SURVEY 8(d) C4: --total-reads 80000000).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from auriclass_amd import engine, multigpu, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline_run(sample: np.ndarray, rb: int, k: int, s: int, m: int, cores: int):
    """CPU baseline on the GPU box's host cores: the C oracle on `cores` record shards, one process
    each (the reference's own advice is one process per input, docs/running_analysis.md:47-59), the
    partial sketches merged.  Bottom-s partials only merge exactly for m = 1, so m > 1 runs as one
    process.  Returns (sketch of the whole sample, wall seconds incl. process start, cores used)."""
    import subprocess
    import tempfile

    from oracle import mash_oracle as mo

    n = sample.size // rb
    mo.lib()  # compile the oracle before anything is timed
    if m > 1 or cores <= 1 or n < cores:
        ref = mo.Sketcher(k, s, m)
        t0 = time.perf_counter()
        ref.add_fastx(sample.tobytes())
        want, _ = ref.finish()
        return want, time.perf_counter() - t0, 1
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=shm) as td:
        path = os.path.join(td, "sample.fq")
        sample.tofile(path)
        env = dict(os.environ, PYTHONPATH=str(ROOT))
        t0 = time.perf_counter()
        procs = []
        for r in range(cores):
            lo, hi = multigpu.shard_bounds(n, cores, r)
            procs.append(subprocess.Popen([sys.executable, "-m", "oracle.shard_worker", path, str(lo * rb), str(hi * rb),
                                           str(k), str(s), os.path.join(td, f"p{r}.npy")], env=env, cwd=str(ROOT)))
        for p in procs:
            if p.wait() != 0:
                raise RuntimeError("cpu baseline worker failed")
        parts = [np.load(os.path.join(td, f"p{r}.npy")) for r in range(cores)]
        want = np.unique(np.concatenate(parts))[:s]
        wall = time.perf_counter() - t0
    return want, wall, cores


def stock_mash_run(sample: np.ndarray, rb: int, k: int, s: int, m: int, cores: int):
    """if a stock `mash` binary is installed on this host, time it on the same
    sample, one process per record shard (`mash sketch -r` is single-threaded).  Returns None when there
    is no such binary (the usual case: only this repository travels to the GPU box)."""
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("mash")
    if exe is None or "auriclass_amd" in os.path.realpath(exe):   # never time our own shim as "stock mash"
        return None
    n = sample.size // rb
    cores = max(1, min(cores, n))
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=shm) as td:
        files = []
        for r in range(cores):
            lo, hi = multigpu.shard_bounds(n, cores, r)
            f = os.path.join(td, f"shard{r}.fq")
            sample[lo * rb:hi * rb].tofile(f)
            files.append(f)
        t0 = time.perf_counter()
        procs = [subprocess.Popen([exe, "sketch", "-r", "-m", str(m), "-k", str(k), "-s", str(s), "-o", f + ".out", f],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for f in files]
        ok = all(p.wait() == 0 for p in procs)
        wall = time.perf_counter() - t0
    return (wall, cores) if ok else None


def cpu_leg(args, fq, nproc, usable):
    """CPU baseline (the C oracle on the host cores: one process, and N record shards) + parity of the GPU sketch of the
    same sample against it.  Returns (cpu_baseline, parity, parity_kind)."""
    n_s = min(args.cpu_sample_reads, args.reads)
    rb = synth.record_bytes(args.read_len)
    sample = fq[: n_s * rb].cpu().numpy()
    # BASELINE.md section 3 / docs/running_analysis.md:47-59 of the reference: one process, and N processes (N stated)
    n_1 = min(n_s, args.cpu_single_reads)
    _, cpu_1, _ = cpu_baseline_run(sample[: n_1 * rb], rb, args.k, args.s, args.m, 1)
    want, cpu_s, cores = cpu_baseline_run(sample, rb, args.k, args.s, args.m, args.cpu_cores)
    cpu_baseline = {"value": round(n_s * args.read_len / cpu_s / 1e9, 5), "unit": "Gbases/s", "cores": cores,
                    "nproc": nproc, "usable_cores": usable,
                    "per_core": round(n_s * args.read_len / cpu_s / 1e9 / max(1, cores), 5),
                    "kind": "port",
                    "one_process": {"value": round(n_1 * args.read_len / cpu_1 / 1e9, 5), "unit": "Gbases/s", "cores": 1,
                                    "sample": f"first {n_1} reads, one process; wall {cpu_1:.1f} s"},
                    "sample": f"first {n_s} reads ({n_s * args.read_len / 1e6:.0f} Mbases) of rank 0's input in {cores} record "
                              f"shards, one process each (N = {cores}: the CPU share of a 1-GPU box of this pool; the host shows "
                              f"{nproc} cores): parse + sketch by the C oracle (oracle/mashcore.c), partial "
                              f"sketches merged; wall {cpu_s:.1f} s"}
    stock = stock_mash_run(sample, rb, args.k, args.s, args.m, args.cpu_cores)
    if stock is not None:   # a real mash on this host: that is the baseline to quote
        cpu_baseline = {"value": round(n_s * args.read_len / stock[0] / 1e9, 5), "unit": "Gbases/s", "cores": stock[1],
                        "nproc": nproc, "usable_cores": usable, "kind": "reference",
                        "sample": f"stock mash sketch -r -m {args.m} -k {args.k} -s {args.s} on the first {n_s} reads in "
                                  f"{stock[1]} record shards, one process each; wall {stock[0]:.1f} s; C-oracle port on the "
                                  f"same sample: {cpu_baseline['value']} Gbases/s on {cpu_baseline['cores']} cores"}
    sk2 = engine.Sketcher(args.k, args.s, args.m, expected_bytes=sample.size)
    sk2.push_device(fq.data_ptr(), n_s * rb, engine.FMT_FASTQ4)
    got, _ = sk2.finish()
    sk2.close()
    parity = bool(np.array_equal(got, want))
    parity_kind = f"GPU sketch of the first {n_s} reads == C oracle ({cores} record shards merged), bit for bit"

    return cpu_baseline, parity, parity_kind


def sharded_gate(args, world, rank, strong, fq, nbytes, genome, dev, result):
    """N > 1: rank 0 sketches EVERY rank's shard (regenerated from the seeds) through one sketcher on its own GPU -- the
    single-GPU path that is pinned against the CPU oracle at N = 1 -- and compares with the merged result."""
    skp = engine.Sketcher(args.k, args.s, args.m, expected_bytes=nbytes * world)
    for r in range(world):
        if r == rank:
            shard = fq
        elif strong:
            lo_r, hi_r = multigpu.shard_bounds(args.total_reads, world, r)
            shard = synth.make_fastq_range(genome, lo_r, hi_r, args.read_len, device=str(dev))
        else:
            shard = synth.make_fastq(genome, args.reads, args.read_len, seed=43 + r, device=str(dev), first_index=r * args.reads)
        torch.cuda.synchronize()
        skp.push_device(shard.data_ptr(), shard.numel(), engine.FMT_FASTQ4)
        skp.sync()   # the shard's buffer may go once the sketcher has settled
        del shard
    want_h, want_c = skp.finish()
    skp.close()
    return bool(np.array_equal(result[0], want_h) and np.array_equal(result[1], want_c))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (weak scaling)")
    ap.add_argument("--total-reads", type=int, default=0, help="reads in all, split over the GPUs (strong scaling); 0 = off")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--s", type=int, default=1000)
    ap.add_argument("--m", type=int, default=1)
    ap.add_argument("--genome", type=int, default=12_000_000)
    ap.add_argument("--cpu-sample-reads", type=int, default=10_000_000)
    ap.add_argument("--cpu-cores", type=int, default=0,
                    help="host cores of the CPU leg; 0 = min(cores this process may run on, 16 = the CPU share of a 1-GPU box)")
    ap.add_argument("--cpu-single-reads", type=int, default=500_000, help="reads of the one-process CPU figure")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-sketch", default="", help="rank 0 saves the final sketch (hashes, counts) to this .npz")
    ap.add_argument("--no-parity", action="store_true", help="N > 1: skip the sharded == unsharded check (rank 0 re-sketches all shards)")
    # ranks started by this script's own launcher get their arguments through the environment: torch.distributed.run's
    # parser claims abbreviations of its own options even behind the script name (`--s` -> "ambiguous option")
    forwarded = os.environ.get("MHX_BENCH_ARGV") if "WORLD_SIZE" in os.environ else None
    args = ap.parse_args(json.loads(forwarded)) if forwarded else ap.parse_args()

    nproc = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = nproc
    if args.cpu_cores <= 0:
        args.cpu_cores = max(1, min(usable, 16))

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Started as plain `python bench.py --gpus N`: this process becomes the launcher.  It has not touched
        # HIP (importing torch and counting devices does not), and it never will: the N ranks are fresh child
        # processes of torch.distributed.run, one per GPU, and their JSON line is passed through.
        import socket
        import subprocess

        if torch.cuda.device_count() < args.gpus and os.environ.get("MHX_DIST_BACKEND", "nccl") == "nccl":
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} visible "
                             "(MHX_DIST_BACKEND=gloo rehearses several ranks on one GPU)")
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MHX_BENCH_ARGV=json.dumps(sys.argv[1:]))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    strong = args.total_reads > 0
    if strong:
        lo, hi = multigpu.shard_bounds(args.total_reads, world, rank)
        args.reads = hi - lo
        first_read, seed = lo, None   # C4: seeds 43.., 10 M reads each (synth.make_fastq_range)
    else:
        first_read, seed = rank * args.reads, 43 + rank
    # one process per GPU; MHX_DIST_BACKEND=gloo lets several ranks rehearse on a 1-GPU box
    backend = os.environ.get("MHX_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # RCCL moves device-resident slabs (one collective per exchange); gloo carries host tensors (MHX_DIST_TENSORS=cuda:
    # device tensors under gloo too, to rehearse the RCCL form of the exchange on a one-GPU box)
    comm_dev = dev if backend == "nccl" or os.environ.get("MHX_DIST_TENSORS") == "cuda" else torch.device("cpu")
    use_dist = world > 1 or os.environ.get("MHX_FORCE_DIST") == "1"   # the latter rehearses the RCCL path on one GPU
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    engine.init(dev_index)

    # ---- synthetic input, resident in HBM before anything is timed -------------------------
    genome = synth.make_genome(args.genome, seed=42)
    if strong:   # the same 80 M-read stream whatever N is: rank r holds records [lo, hi) of it
        fq = synth.make_fastq_range(genome, first_read, first_read + args.reads, args.read_len, device=str(dev))
    else:
        fq = synth.make_fastq(genome, args.reads, args.read_len, seed=seed, device=str(dev), first_index=first_read)
    torch.cuda.synchronize()
    nbytes = fq.numel()
    bases = args.reads * args.read_len
    sk = engine.Sketcher(args.k, args.s, args.m, expected_bytes=nbytes)

    def push(target):
        target.push_device(fq.data_ptr(), nbytes, engine.FMT_FASTQ4)

    def step():
        sk.reset()
        push(sk)
        if not use_dist:
            return sk.finish()
        # RCCL: the partial results stay on the GPU until they have been gathered (gloo: host slabs); if the gathered
        # partials do not determine the union's sketch every rank re-sketches with a wider budget, inside the step
        return multigpu.sharded_sketch(push, args.k, args.s, args.m, nbytes, comm_dev, first=sk)

    def barrier():
        torch.cuda.synchronize()
        sk.sync()
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        result = step()
    barrier()
    exch = {}   # this rank's time inside the exchange (export, size all-gather, pack, slab all-gather, device merge), summed
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
        if use_dist:
            for key, v in multigpu.last_exchange.items():
                if key.endswith("_ms"):
                    exch[key] = exch.get(key, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    total_bases = bases
    if use_dist:   # ranks may hold different numbers of reads (strong scaling with a remainder)
        tb = torch.tensor([bases], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_bases = int(tb.item())
    value = total_bases / (elapsed / args.steps) / 1e9
    ranks_in_group = dist.get_world_size() if use_dist else 1

    # ---- dominant kernel: HIP events on the engine's stream around every tile-kernel launch --
    roofline = None
    cpu_baseline = None
    parity = None
    parity_kind = None
    if rank == 0:
        engine.set_profiling(True)
        ms = []
        for _ in range(3):
            sk.reset()
            sk.push_device(fq.data_ptr(), nbytes, engine.FMT_FASTQ4)
            sk.finish()
            st = sk.stats()
            ms.append(st["hash_ms"])
        engine.set_profiling(False)
        kernel_ms = float(np.median(ms))
        achieved = nbytes / (kernel_ms / 1e3) / 1e9
        # HBM bytes per step from the PMC passes (profiles/traffic.json), only quoted for the very
        # workload they were collected on
        traffic = None
        valu_instr = None
        clock_ghz = None
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists():
            try:
                tj = json.loads(tf.read_text())
                if tj.get("algorithmic_bytes_per_step") == nbytes and (args.k, args.s, args.m) == (21, 1000, 1):
                    traffic = tj.get("sketch_tile_kernel_hbm_bytes_per_step")
                    valu_instr = tj.get("sketch_tile_kernel_valu_wave_instructions_per_step")
                    clock_ghz = tj.get("shader_clock_ghz_from_pmc")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "kernel": "sketch_tile_kernel", "kernel_ms_per_step": round(kernel_ms, 4),
                    "launches_per_step": st["launches"], "algorithmic_bytes_per_step": nbytes,
                    "kmers_per_s": round(st["kmers"] / (kernel_ms / 1e3) / 1e9, 3), "kmers_unit": "G k-mers/s",
                    # achieved / frac / kernel_ms_per_step are measured in THIS run (HIP events on the engine stream);
                    # traffic is a counter reading of an earlier profiling run of the same workload, quoted from a file
                    "traffic_source": ("profiles/traffic.json: rocprofv3 --pmc passes (tools/profile.sh) of this exact workload, "
                                       "not of this run" if traffic else None)}
        if valu_instr and clock_ghz:
            # the operative limit: wave64 integer VALU instructions issue at ~4 cycles each per SIMD;
            # instruction count and shader clock from the PMC passes of the same workload, time measured live
            simds = 256 * 4
            roofline["valu_issue"] = {"wave_instructions_per_step": valu_instr,
                                      "cycles_per_instruction_per_simd": round(kernel_ms * 1e-3 * clock_ghz * 1e9 * simds / valu_instr, 3),
                                      "issue_cost_cycles": 4.0, "clock_ghz": clock_ghz,
                                      "source": "profiles/traffic.json (SQ_INSTS_VALU and GRBM_GUI_ACTIVE of the PMC passes), not this run"}

        # ---- N > 1: the gate of the sharded line (sharded_gate), outside the timed region
        if world > 1 and not args.no_parity:
            try:
                parity = sharded_gate(args, world, rank, strong, fq, nbytes, genome, dev, result)
                parity_kind = (f"sharded sketch over {world} ranks == one sketcher over all {world} shards on rank 0's GPU "
                               "(hashes and multiplicities); that single-GPU path is the one pinned against the CPU oracle at N=1")
            except Exception as e:   # the measured line must come out whatever happens to the check: it then says what did
                parity = None
                parity_kind = f"gate not evaluated: {type(e).__name__}: {e}"

        # ---- CPU baseline + parity on a bounded sample of the same workload -------------------
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is an N=1 figure; at N>1 every rank's host cores are busy
            try:
                cpu_baseline, parity, parity_kind = cpu_leg(args, fq, nproc, usable)
            except Exception as e:   # the measured line still comes out; it says why the baseline is missing
                cpu_baseline = {"value": None, "unit": "Gbases/s", "cores": 0, "kind": "port", "sample": f"not measured: {type(e).__name__}: {e}"}

    if rank == 0:
        line = {
            "metric": "Gbases/s sketched (k=21, s=1000)", "value": round(value, 3), "unit": "Gbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": (f"synthetic FASTQ {args.total_reads} x {args.read_len} bp reads in all, record-sharded over "
                                    f"{world} GPU(s) ({args.reads} on rank 0, {nbytes} bytes resident in its HBM)" if strong else
                                    f"synthetic FASTQ {args.reads} x {args.read_len} bp reads per GPU from a "
                                    f"{args.genome} bp genome, 0.5% substitutions, {nbytes} bytes resident in HBM"),
                       "k": args.k, "s": args.s, "min_multiplicity": args.m, "total_bases": total_bases,
                       "ranks": ranks_in_group, "collective": (backend if use_dist else None),
                       # rank 0's mean time per step inside the exchange: shard export, 64-byte size all-gather, slab pack,
                       # slab all-gather, merge of the other ranks' entries into its table + extraction (all on the device)
                       "exchange_ms": ({key: round(v / args.steps, 4) for key, v in sorted(exch.items())} if exch else None),
                       "exchange_entries_per_rank": (multigpu.last_exchange.get("entries_per_rank") if use_dist else None),
                       "parallelism": "1 process/GPU, record shards, all-gather of partial sketches" if world > 1 else "1 GPU"},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
            "parity_on_sample": parity, "parity_kind": parity_kind, "sketch_len": int(len(result[0])),
        }
    if rank == 0 and args.dump_sketch:
        np.savez(args.dump_sketch, hashes=result[0], counts=result[1])
    sk.close()
    if use_dist:
        dist.barrier()   # rank 0 may have spent a while in the parity gate: leave together
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(line), flush=True)   # the one JSON line, last thing on stdout


if __name__ == "__main__":
    main()
