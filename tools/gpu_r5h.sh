#!/bin/bash
# small-batch ablations of the per-wave K3 (what its 42 us are made of) + batch-100 replay + pack-multi / by-source checks
set -u
TAG=${1:-r5h}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "cgconv or model or replay or padded or batchnorm" 2>&1 | tail -20 > $OUT/pytest.log; tail -3 $OUT/pytest.log
run() { local lab=$1 lib=$2; shift 2
  echo "== $lab $*" | tee -a $OUT/ab.log
  if [ "$lib" = "-" ]; then timeout 300 python tools/bench_kernels.py "$@" 2>&1 | grep -E "^(N=|fwd|bwd|bwd_node):|rror" | tee -a $OUT/ab.log
  else MDL_HIP_LIB=$GRAFT_REPO_ROOT/matdeeplearn_amd/lib/variants/$lib.so timeout 300 python tools/bench_kernels.py "$@" 2>&1 | grep -E "^(N=|fwd|bwd|bwd_node):|rror" | tee -a $OUT/ab.log; fi
}
run base - --which fwd,bwd --iters 40 --graphs 100
for v in nodwe nowfl nodwe_nowfl nopre r128; do run $v $v --which fwd,bwd --iters 40 --graphs 100; done
run base - --which fwd,bwd --iters 40 --graphs 100
for i in 1 2; do timeout 300 python tools/bench_small.py --steps 600 2>&1 | tail -1 | tee -a $OUT/ab.log; done
