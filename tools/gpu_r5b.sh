#!/bin/bash
# round 5, second lease: K3 / K2 A/B of prebuilt variants on the bench batch AND at the reference's batch size, the small-batch replay
set -u
TAG=${1:-r5b}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { # label, lib (or -), bench_kernels args
  local lab=$1 lib=$2; shift 2
  echo "== $lab $*" | tee -a $OUT/ab.log
  if [ "$lib" = "-" ]; then timeout 300 python tools/bench_kernels.py "$@" 2>&1 | grep -E "^(N=|fwd|bwd|bwd_node):|rror" | tee -a $OUT/ab.log
  else MDL_HIP_LIB=$GRAFT_REPO_ROOT/matdeeplearn_amd/lib/variants/$lib.so timeout 300 python tools/bench_kernels.py "$@" 2>&1 | grep -E "^(N=|fwd|bwd|bwd_node):|rror" | tee -a $OUT/ab.log; fi
}
run base - --which fwd,bwd --iters 20
for v in slp prioR1 prioP1; do run $v $v --which fwd,bwd --iters 20; done
run base - --which fwd,bwd --iters 20
run base-small - --which fwd,bwd --iters 40 --graphs 100
for v in r128 r256 r512; do run $v-small $v --which fwd,bwd --iters 40 --graphs 100; done
run base-small - --which fwd,bwd --iters 40 --graphs 100
timeout 300 python tools/bench_small.py 2>&1 | tail -1 | tee -a $OUT/ab.log
timeout 600 python -m pytest tests -m gpu -q -x -k "fused_post_fc_head or padded_rows or replay" 2>&1 | tail -4 | tee -a $OUT/ab.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_small -o t -- python $GRAFT_REPO_ROOT/tools/bench_small.py > $OUT/small.log 2>&1
f=$(find $OUT/prof_small -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -70 "$f" > $OUT/kernel_stats_small.csv
rm -rf $OUT/prof_small
