#!/bin/bash
# PMC passes over the sketch kernel of one library variant (run through gpurun from the repo root):
#   tools/pmc.sh <variant|base> <tag> [reads]   -> gpurun_out/pmc_<tag>/<pass>/..., summary in gpurun_out/pmc_<tag>/summary.txt
# Each pass is its own process; the program comes directly after `--` (no env/bash hop under the profiler).
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$1; TAG=$2; READS=${3:-4000000}
if [ "$V" = base ]; then export MHX_LIB=$R/auriclass_amd/lib/libmhx.so; else export MHX_LIB=$R/auriclass_amd/lib_variants/$V.so; fi
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_IFETCH SQ_LDS_BANK_CONFLICT" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_CYCLES"; do
    i=$((i+1))
    rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -- python3 $R/tools/ab.py --child --reads $READS --iters 3 > $OUT/p$i.log 2>&1
    echo "pass $i done"
done
cd $R
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); calls = 0
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sketch_tile_kernel" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as w:
    for k in sorted(tot): w.write(f"{k} {tot[k]:.0f}\n")
print(open(out + "/summary.txt").read())
PY
