#!/bin/bash
set -u
TAG=${1:-r6t}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 1 2 3 4 5 6 7 8 9 10; do
  timeout 900 python -m pytest tests/test_gpu_distributed.py -m gpu -q -k "rccl_backend" 2>&1 | grep -E "passed|failed" | tee -a $OUT/log.txt
  if ! grep -q RCCL_OK gpurun_out/rccl_world1.log; then cp gpurun_out/rccl_world1.log $OUT/rccl_fail_$r.log; fi
done
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -3 | tee -a $OUT/log.txt
