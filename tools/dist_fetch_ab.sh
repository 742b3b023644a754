#!/bin/bash
# HBM-side fetch traffic (L2 -> fabric, FETCH_SIZE) and kernel time of the C5 distance batch for library variants:
#   tools/dist_fetch_ab.sh OUTDIR name1 name2 ...   ("base" = auriclass_amd/lib/libmhx.so; others from tools/build_variants.sh)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
    if [ "$v" = base ]; then unset MHX_LIB; else export MHX_LIB=$R/auriclass_amd/lib_variants/$v.so; fi
    python3 $R/tools/dist_c5.py --reps 5 2>/dev/null | tail -n 1 > $OUT/line_$v.json
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$v -- python3 $R/tools/dist_c5.py --reps 3 > $OUT/fetch_$v.log 2>&1
    python3 - $OUT/fetch_$v $v <<'PY' | tee -a $OUT/summary.txt
import csv, glob, sys, collections
tot = collections.Counter(); calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "dist_" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0]
            tot[k] += float(r["Counter_Value"]); calls[k] += 1
for k in sorted(tot):
    print(f"{sys.argv[2]:10s} {k:36s} calls {calls[k]:4d}  FETCH_SIZE per call {tot[k] / calls[k]:12.1f} KB")
PY
    rm -rf $OUT/fetch_$v
done
cat $OUT/line_*.json
