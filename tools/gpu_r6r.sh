#!/bin/bash
set -u
TAG=${1:-r6r}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 1 2 3 4 5 6; do
  timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -k "bf16x3" 2>&1 | grep -E "passed|failed|Assertion|assert |^E " | head -8 | tee -a $OUT/log.txt
done
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -3 | tee -a $OUT/log.txt
