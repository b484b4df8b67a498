#!/bin/bash
set -u
TAG=${1:-r6s}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 1 2 3 4 5; do
  timeout 900 python -m pytest tests/test_gpu_distributed.py -m gpu -q -k "rccl_backend" 2>&1 | grep -E "passed|failed" | tee -a $OUT/log.txt
  if ! grep -q RCCL_OK gpurun_out/rccl_world1.log; then cp gpurun_out/rccl_world1.log $OUT/rccl_fail_$r.log; fi
done
for r in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -k "bf16x3" 2>&1 | grep -E "passed|failed" | tee -a $OUT/log.txt; done
