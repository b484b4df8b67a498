#!/bin/bash
# gpu_nnconv_ab.sh <tag> <variants...>: tools/bench_nnconv.py on the built library and on prebuilt variants, one box session
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== built library" | tee $OUT/ab.log
timeout 300 python tools/bench_nnconv.py 2>&1 | grep -E "us |rror|Trace" | tee -a $OUT/ab.log
for v in "$@"; do
  echo "== variant $v" | tee -a $OUT/ab.log
  MDL_HIP_LIB=$GRAFT_REPO_ROOT/matdeeplearn_amd/lib/variants/$v.so timeout 300 python tools/bench_nnconv.py 2>&1 | grep -E "us |rror|Trace" | tee -a $OUT/ab.log
done
true
