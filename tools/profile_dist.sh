#!/bin/bash
# Counter evidence for the batched distance kernels (BASELINE.json config 5), run through gpurun from the repo root:
# a --kernel-trace --stats pass and separate --pmc passes (HBM read, HBM write, SQ mix) over tools/dist_c5.py,
# the program directly after `--`.   -> gpurun_out/prof_dist/..., summary by tools/summarize_prof.py
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_dist
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPS=3; CALLS=$((2 * REPS + 1))   # dist_c5.py: REPS + 1 event-timed calls, then REPS wall-timed ones
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5trace -- python3 $R/tools/dist_c5.py --reps $REPS > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log > $OUT/dist_line_under_trace.json
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" \
            "sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM" \
            "sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    set -- $pass
    name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/dist_c5.py --reps $REPS > $OUT/$name.log 2>&1
    echo "pass $name done"
done
cd $R
python3 tools/summarize_prof.py gpurun_out/prof_dist gpurun_out/prof_dist/summary $CALLS
find $OUT -name "*counter_collection.csv" -delete   # tens of MiB per pass; the summary has what is kept
find $OUT -name "*kernel_trace.csv" -delete
echo profile done
