#!/bin/bash
# gpu_ab_small.sh <tag> <ENVVAR> [reps]: the batch-100 replay (600 steps) with ENVVAR=1 / 0 alternating in ONE box session
set -u
TAG=$1; VAR=$2; REPS=${3:-3}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ -n "${PRE_TESTS:-}" ]; then timeout 1200 python -m pytest tests -m gpu -q -x -k "$PRE_TESTS" 2>&1 | tail -25 > $OUT/pytest.log; tail -3 $OUT/pytest.log; fi
for rep in $(seq $REPS); do
  for v in 1 0; do
    echo "== $VAR=$v $(env $VAR=$v timeout 300 python tools/bench_small.py --steps 600 2>&1 | tail -1)" | tee -a $OUT/ab.log
  done
done
