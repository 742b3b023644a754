#!/bin/bash
# Re-creates the profiles/ evidence on the GPU box (run through gpurun from the repo root):
#   one --kernel-trace --stats pass of the default bench command, then separate --pmc passes
#   (HBM read, HBM write, SQ instruction mix, SQ wave states), each its own process as the MI355X guide
#   prescribes, the program directly after `--`.
# usage: tools/profile.sh [extra bench.py flags]   -> gpurun_out/prof/<pass>/..., summarised by tools/summarize_prof.py
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
grep "^{\"metric\"" $OUT/trace.log | tail -1 > $OUT/bench_line_under_trace.json
STEPS=7   # passes over the input per PMC run: 2 warmup + 2 timed + 3 event-timed (bench.py)
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" \
            "sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH" \
            "sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
    set -- $pass
    name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/$name.log 2>&1
    echo "pass $name done"
done
cd $R
python3 tools/summarize_prof.py gpurun_out/prof gpurun_out/prof/summary $STEPS
find $OUT -name "*counter_collection.csv" -delete   # the summary has what is kept
echo profile done
