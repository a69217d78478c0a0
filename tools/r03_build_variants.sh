#!/bin/bash
# Builds variants of the library for an A/B run on the GPU box (tools/r02_ab.sh takes .ab/lib<name>.so):
#   tools/r03_build_variants.sh name1:"-DFLAG ..." name2:"" ...      (run in the container, not on the box)
# Every variant is built in its own copy of csrc under /tmp, so the product library is not touched.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/.ab
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( d=/tmp/abbuild_$name; rm -rf $d; mkdir -p $d/mckpp_f90_amd $d/include
    cp -r $ROOT/mckpp_f90_amd/csrc $d/mckpp_f90_amd/csrc; cp $ROOT/include/*.h $d/include/; rm -f $d/mckpp_f90_amd/csrc/*.o $d/mckpp_f90_amd/csrc/*.s $d/mckpp_f90_amd/csrc/.kernel_checked
    mkdir -p $d/tools; cp $ROOT/tools/check_build.py $ROOT/tools/check_inflight.py $d/tools/   # the Makefile's build gate
    make -C $d/mckpp_f90_amd/csrc EXTRA="$flags" > $d/build.log 2>&1 || { echo "$name: build failed"; tail -5 $d/build.log; exit 1; }
    cp $d/mckpp_f90_amd/libmckpp_hip.so $ROOT/.ab/lib$name.so; echo "$name: built ($flags)" ) &
done
wait
