#!/bin/bash
# gpu_dense_ab.sh <tag> <variants...>: tools/bench_dense.py on the built library and on prebuilt variants, one box session
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ROWS=${ROWS:-1498398}
echo "== built library" | tee $OUT/ab.log
timeout 300 python tools/bench_dense.py $ROWS 2>&1 | grep -E "us |rror" | tee -a $OUT/ab.log
for v in "$@"; do
  echo "== variant $v" | tee -a $OUT/ab.log
  MDL_HIP_LIB=$GRAFT_REPO_ROOT/matdeeplearn_amd/lib/variants/$v.so timeout 300 python tools/bench_dense.py $ROWS 2>&1 | grep -E "us |rror" | tee -a $OUT/ab.log
done
true
