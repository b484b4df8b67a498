#!/bin/bash
# round 6: the split-bf16 ("bf16x3") CGConv kernels — parity tests, kernel timings against the exact-fp32 form, the step
set -u
TAG=${1:-r6e}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_workloads.py -m gpu -q -x -k "split_bf16 or (cgconv_matches_oracle and float32)" 2>&1 | tail -15 | tee -a $OUT/log.txt
for sp in 0 1; do
  echo -n "split=$sp: " | tee -a $OUT/log.txt
  MDL_BK_SPLIT=$sp timeout 300 python tools/bench_kernels.py --dtype fp32 --which fwd,bwd --iters 6 2>&1 | grep -E "^(fwd|bwd):|rror" | tr '\n' ' ' | tee -a $OUT/log.txt; echo | tee -a $OUT/log.txt
done
for cd in fp32 bf16x3; do
  timeout 600 python bench.py --dtype $cd --no-extras --no-cpu-baseline --no-other-models --steps 6 --warmup 2 --settle-s 0.3 --settle-cap-s 1.0 2>$OUT/err_$cd.txt | grep '^{"metric"' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$cd step ms', j['ms_per_step'])" | tee -a $OUT/log.txt
  tail -3 $OUT/err_$cd.txt | cut -c1-300 >> $OUT/log.txt
done
