#!/bin/bash
# Second half of a round's profiles (see tools/round_profiles.sh).  profiles/traffic.json is made from the first half's
# PMC summary afterwards, wherever those files are:  python tools/make_traffic.py <pmc_summary> <trace.csv> <line.json> profiles/traffic.json r03
#   tools/round_profiles2.sh r03
set -o pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/$TAG
mkdir -p $O
# sustained clocks: the same step 1200 times (> 5 s of kernels back to back)
python bench.py --no-cpu-baseline --steps 1200 --warmup 5 2>&1 | tail -1 > $O/${TAG}_bench_line_sustained_1200_steps.json
# the sharded path: merge cost on one rank with the GPU to itself, then 5 ranks sharing the GPU (gloo) through bench.py
python tools/merge_time.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_sharded_merge_time.txt
tools/rehearse_ranks.sh $O 5 > $O/rehearse.log 2>&1
for f in $O/r03_rehearsal_*.json; do tail -1 $f > $f.tmp && mv $f.tmp $f; done
tools/profile_dist.sh > $O/profile_dist.log 2>&1
cp gpurun_out/prof_dist/summary_pmc_summary.json $O/${TAG}_dist_c5_pmc_summary.json
cp gpurun_out/prof_dist/summary_kernel_stats_c5trace.csv $O/${TAG}_dist_c5_kernel_stats.csv
ls -la $O
echo round profiles 2 done
