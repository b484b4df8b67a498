#!/bin/bash
# round 6: the full GPU suite three more times in fresh processes (no -x: every failure is listed)
set -u
TAG=${1:-r6u}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -E "^FAILED|passed|failed" | tee -a $OUT/log.txt
done
