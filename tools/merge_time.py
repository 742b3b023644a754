"""Cost of the sharded path's merge step on ONE rank with the GPU to itself (the rehearsals of tools/rehearse_ranks.sh
share one GPU among all ranks, so their exchange times are mostly queueing): N shards of the C4 read set are sketched one
after the other, each exported as the slab its rank would contribute; then rank 0's sketcher takes the N gathered slabs
(already in HBM, as after an RCCL all-gather) and merges them on the device (mhx_sketcher_merge_slabs).  Timed: shard
export (extract kernel + header), slab pack, device merge + extraction; beside it the round-2 host merge
(mhx_merge_shard_partials) on the same partials.  The merged sketch is compared with one sketcher over all N shards.

    python tools/merge_time.py [--ranks 8] [--reads 10000000]
"""
import argparse
import statistics
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from auriclass_amd import engine, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--reads", type=int, default=10_000_000)
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
engine.init(0)
genome = synth.make_genome(12_000_000, 42)
N = args.ranks
for k, s, m in ((21, 1000, 1), (21, 1000, 3), (27, 50000, 3)):
    whole = engine.Sketcher(k, s, m, expected_bytes=N * args.reads * synth.record_bytes(150))
    sk = engine.Sketcher(k, s, m, expected_bytes=args.reads * synth.record_bytes(150))
    headers, parts = [], []
    shard0 = None
    for r in range(N):
        fq = synth.make_fastq(genome, args.reads, 150, seed=43 + r, device="cuda", first_index=r * args.reads)
        torch.cuda.synchronize()
        whole.push_device(fq.data_ptr(), fq.numel(), engine.FMT_FASTQ4)
        whole.sync()
        sk.reset()
        sk.push_device(fq.data_ptr(), fq.numel(), engine.FMT_FASTQ4)
        sk.sync()
        hdr = sk.export_begin()
        cap = (int(hdr[0]) + 1023) // 1024 * 1024
        slab = torch.empty(cap + cap // 2, dtype=torch.int64, device="cuda")
        sk.export_pack(slab.data_ptr(), cap)
        headers.append(hdr)
        parts.append((slab, cap))
        if r == 0:
            shard0 = fq
        else:
            del fq
    want_h, want_c = whole.finish()
    whole.close()
    cap = max(c for _, c in parts)
    words = cap + cap // 2
    gathered = torch.zeros(N * words, dtype=torch.int64, device="cuda")
    for r, (slab, c) in enumerate(parts):   # re-lay every slab with the common capacity, as the ranks would have packed it
        n = int(headers[r][0])
        gathered[r * words:r * words + n] = slab[:n]
        cnt = slab[c:].view(torch.int32)[:n]
        gathered[r * words + cap:(r + 1) * words].view(torch.int32)[:n] = cnt
    torch.cuda.synchronize()
    hdrs = np.stack(headers)
    t_exp, t_pack, t_merge = [], [], []
    send = torch.empty(words, dtype=torch.int64, device="cuda")
    for _ in range(args.iters + 1):
        sk.reset()
        sk.push_device(shard0.data_ptr(), shard0.numel(), engine.FMT_FASTQ4)
        sk.sync()
        t0 = time.perf_counter()
        h0 = sk.export_begin()
        t1 = time.perf_counter()
        sk.export_pack(send.data_ptr(), cap)
        t2 = time.perf_counter()
        assert np.array_equal(h0[:2], hdrs[0][:2])
        got_h, got_c = sk.merge_slabs(gathered.data_ptr(), True, N, cap, hdrs, 0)
        t3 = time.perf_counter()
        t_exp.append((t1 - t0) * 1e3); t_pack.append((t2 - t1) * 1e3); t_merge.append((t3 - t2) * 1e3)
    ok = np.array_equal(got_h, want_h) and np.array_equal(got_c, want_c)
    # the one-collective form (device-resident slabs under RCCL): header-carrying slabs, export straight into the send buffer
    words2 = 8 + cap + cap // 2
    gathered2 = torch.zeros(N * words2, dtype=torch.int64, device="cuda")
    hdr_t = torch.from_numpy(hdrs.view(np.int64).copy()).to("cuda")
    for r in range(N):
        gathered2[r * words2:r * words2 + 8] = hdr_t[r]
        gathered2[r * words2 + 8:(r + 1) * words2] = gathered[r * words:(r + 1) * words]
    send2 = torch.empty(words2, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    t_exp2, t_merge2 = [], []
    for _ in range(args.iters + 1):
        sk.reset()
        sk.push_device(shard0.data_ptr(), shard0.numel(), engine.FMT_FASTQ4)
        sk.sync()
        t0 = time.perf_counter()
        h0 = sk.export_into(send2.data_ptr(), cap)
        t1 = time.perf_counter()
        gathered2[:words2] = send2          # (what the all-gather would do with this rank's slab)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        g2_h, g2_c, need = sk.merge_gathered(gathered2.data_ptr(), N, cap, 0)
        t3 = time.perf_counter()
        assert need == 0
        t_exp2.append((t1 - t0) * 1e3); t_merge2.append((t3 - t2) * 1e3)
    ok = ok and np.array_equal(g2_h, want_h) and np.array_equal(g2_c, want_c)
    # the host merge of round 2 on the same partials
    host = gathered.cpu().numpy()
    hh = [host[r * words:r * words + int(hdrs[r][0])].view(np.uint64) for r in range(N)]
    hc = [host[r * words + cap:(r + 1) * words].view(np.uint32)[:int(hdrs[r][0])] for r in range(N)]
    t_host = []
    for _ in range(3):
        t0 = time.perf_counter()
        ref_h, _ = engine.merge_shard_partials(hh, hc, [int(x[1]) for x in hdrs], k, s, m)
        t_host.append((time.perf_counter() - t0) * 1e3)
    med = statistics.median
    print(f"k={k} s={s} m={m} ranks={N} reads/rank={args.reads}: entries/rank {[int(x[0]) for x in hdrs][:3]}..., slab {words * 8 / 1e6:.2f} MB/rank | "
          f"export {med(t_exp[1:]):.3f} ms, pack {med(t_pack[1:]):.3f} ms, device merge+extract {med(t_merge[1:]):.3f} ms "
          f"(sum {med(t_exp[1:]) + med(t_pack[1:]) + med(t_merge[1:]):.3f} ms) | one-collective form: export into the send slab "
          f"{med(t_exp2[1:]):.3f} ms, merge {med(t_merge2[1:]):.3f} ms (sum {med(t_exp2[1:]) + med(t_merge2[1:]):.3f} ms) | "
          f"host merge (round 2) {min(t_host):.2f} ms | "
          f"merged == one sketcher over all shards: {ok and np.array_equal(ref_h, want_h)}", flush=True)
    sk.close()
    del gathered, gathered2, shard0, parts
