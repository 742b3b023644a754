#!/bin/bash
# Rehearsal of the sharded bench line with N ranks SHARING the box's one GPU (gloo carries the exchanged bytes; sketching,
# shard export and the merge of the gathered partials run on the GPU as under RCCL).  NOT a scaling number: the ranks
# queue for the same device.  What it shows: the N > 1 code path end to end, the gate (parity_on_sample) and the time a
# rank spends in the exchange (config.exchange_ms).  The pool allows at most 6 processes with the card open, and the launcher counts as one: N <= 5.
#   tools/rehearse_ranks.sh OUTDIR [N]
set -e
out=${1:-gpurun_out}; n=${2:-5}
cd "$(dirname "$0")/.."
export MHX_DIST_BACKEND=gloo
sleep 2; python bench.py --gpus $n --reads 1250000 --steps 10 --warmup 3 > $out/r03_rehearsal_${n}ranks_k21_s1000_m1.json
sleep 2; python bench.py --gpus $n --reads 1250000 --steps 10 --warmup 3 --m 3 > $out/r03_rehearsal_${n}ranks_k21_s1000_m3.json
sleep 2; python bench.py --gpus $n --reads 1250000 --steps 10 --warmup 3 --k 27 --s 50000 --m 3 > $out/r03_rehearsal_${n}ranks_k27_s50000_m3.json
sleep 2; python bench.py --gpus $n --total-reads $((n * 1250000)) --steps 10 --warmup 3 > $out/r03_rehearsal_${n}ranks_strong_k21_s1000_m1.json
# the RCCL form of the exchange (device-resident slabs, one collective) with gloo as the carrier
sleep 2; MHX_DIST_TENSORS=cuda python bench.py --gpus $n --reads 1250000 --steps 10 --warmup 3 --k 27 --s 50000 --m 3 > $out/r03_rehearsal_${n}ranks_k27_s50000_m3_device_slabs.json
