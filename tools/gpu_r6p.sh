#!/bin/bash
set -u
TAG=${1:-r6p}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_workloads.py -m gpu -q -k "bf16x3 or split" 2>&1 | tail -12 | tee -a $OUT/log.txt
