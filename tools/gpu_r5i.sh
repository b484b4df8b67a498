#!/bin/bash
set -u
TAG=${1:-r5i}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "cgconv or model or replay or padded or workloads" 2>&1 | tail -20 > $OUT/pytest.log; tail -3 $OUT/pytest.log
for i in 1 2; do timeout 300 python tools/bench_kernels.py --which fwd,bwd --iters 40 --graphs 100 2>&1 | grep -E "^(N=|fwd|bwd|bwd_node):|rror" | tee -a $OUT/ab.log; done
for i in 1 2; do timeout 300 python tools/bench_small.py --steps 600 2>&1 | tail -1 | tee -a $OUT/ab.log; done
