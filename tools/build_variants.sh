#!/bin/bash
# Experiment builds of libmhx: tools/build_variants.sh name "DEFS" [name "DEFS" ...]
#   -> auriclass_amd/lib_variants/<name>.so (git-ignored; travels to the GPU box), objects in csrc/_obj_<name>
# Only mhx_kernels.hip is rebuilt per variant when MHX_ONLY_K is part of DEFS it compiles one k only (seconds).
set -e
cd "$(dirname "$0")/../auriclass_amd/csrc"
mkdir -p ../lib_variants
while [ $# -ge 2 ]; do
    name=$1; defs=$2; shift 2
    make -s -j4 OBJDIR=_obj_$name OUT=../lib_variants/$name.so DEFS="$defs" &
done
wait
ls -la ../lib_variants
