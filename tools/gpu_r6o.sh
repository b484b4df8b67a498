#!/bin/bash
# round 6: bf16x3 on every model — the five-model test, and the legs' step times in fp32 / bf16x3
set -u
TAG=${1:-r6o}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_workloads.py -m gpu -q -x -k "bf16x3 or split" 2>&1 | tail -12 | tee -a $OUT/log.txt
for m in megnet schnet mpnn gcn; do
  timeout 900 python bench.py --model $m --steps 6 --warmup 2 --settle-s 0.3 --settle-cap-s 1.0 --no-extras --fp32-leg --no-cpu-baseline --no-other-models 2>$OUT/err_$m.txt | grep '^{"metric"' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$m bf16', j['ms_per_step'], 'fp32', j['fp32_mode']['ms_per_step'], 'bf16x3', j['bf16x3_mode'])" | tee -a $OUT/log.txt
  tail -2 $OUT/err_$m.txt | cut -c1-300 >> $OUT/log.txt
done
