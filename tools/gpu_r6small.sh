#!/bin/bash
# batch-100 regime (round 6, glue launches removed): the new kernel tests, the replay / padded-row tests, then the replayed step's time and its
# ordered kernel list.  gpu_r6small.sh <tag> [pytest -k expression]
set -u
TAG=${1:-r6small}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
K=${2:-"fp32_prediction or first_rows or padded_batch_assembly or post_fc_head or fused_loss or replay or padded_rows or batch_assembly"}
timeout 1500 python -m pytest tests -m gpu -q -x -k "$K" 2>&1 | grep -vE "^\s*$" | tail -${TAILN:-25} | tee $OUT/pytest.log
for rep in 1 2 3; do
  timeout 300 python tools/bench_small.py --steps 600 2>&1 | tail -1 | tee -a $OUT/small.log
done
bash tools/gpu_small_trace.sh $TAG > /dev/null 2>&1
head -70 $OUT/small_trace.txt
