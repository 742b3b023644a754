#!/bin/bash
# Everything profiles/ quotes for one round, in two gpurun calls:  tools/round_profiles.sh r03 ; tools/round_profiles2.sh r03
# (bench lines, kernel trace + PMC passes of the headline command, inclusive timings, C2, C5, one AuriClass-sized sample)
set -o pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/$TAG
mkdir -p $O
python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/${TAG}_bench_line.json
python bench.py --no-cpu-baseline --m 3 > $O/bench_m3.log 2>&1; tail -1 $O/bench_m3.log > $O/${TAG}_bench_line_m3.json
python bench.py --no-cpu-baseline --k 27 --s 50000 --m 3 > $O/bench_k27.log 2>&1; tail -1 $O/bench_k27.log > $O/${TAG}_bench_line_k27_s50000_m3.json
tools/profile.sh > $O/profile.log 2>&1
cp gpurun_out/prof/summary_pmc_summary.json $O/${TAG}_pmc_summary.json
cp gpurun_out/prof/summary_kernel_stats_trace.csv $O/${TAG}_kernel_stats_trace.csv
cp gpurun_out/prof/bench_line_under_trace.json $O/${TAG}_bench_line_under_trace.json
python tools/step_anatomy.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_step_anatomy.txt
python tools/step_anatomy.py --m 3 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_step_anatomy.txt
python tools/step_anatomy.py --k 27 --s 50000 --m 3 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_step_anatomy.txt
python tools/c3_inclusive.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_c3_inclusive.txt
python tools/c2_time.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_c2_file_level.txt
python tools/dist_c5.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/${TAG}_dist_c5.json
python tools/sample_time.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_sample_end_to_end.txt
ls -la $O
echo round profiles done
