#!/bin/bash
# round 6: the split-product kernels at the reference's default width (dim1 = 100 -> 128-channel static kernels), new split tests
set -u
TAG=${1:-r6m}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_workloads.py -m gpu -q -x -k "split" 2>&1 | tail -12 | tee -a $OUT/log.txt
for cd in bf16 bf16x3 fp32; do
  timeout 900 python bench.py --dtype $cd --dim 100 --no-extras --no-cpu-baseline --no-other-models --steps 6 --warmup 2 --settle-s 0.3 --settle-cap-s 1.0 2>$OUT/err_$cd.txt | grep '^{"metric"' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('dim100 $cd step ms', j['ms_per_step'])" | tee -a $OUT/log.txt
  tail -2 $OUT/err_$cd.txt | cut -c1-300 >> $OUT/log.txt
done
