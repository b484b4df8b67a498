#!/bin/bash
# round 5, third lease: full GPU suite after the struct entry points + BatchNorm statistics in the K2 epilogue; A/B of the fused statistics
set -u
TAG=${1:-r5c}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $OUT/pytest.log; tail -5 $OUT/pytest.log
for rep in 1 2; do
  for v in 1 0; do
    echo "== MDL_CG_BN_STATS=$v" | tee -a $OUT/ab.log
    MDL_CG_BN_STATS=$v timeout 300 python bench.py --no-extras --no-cpu-baseline --no-other-models 2>&1 | grep '^{"metric"' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['ms_per_step'], j['config']['ms_per_step_by_4'], 'K3', r['avg_launch_us'], r['parts_avg_launch_us'], 'K2', j['roofline_other']['avg_launch_us'])" | tee -a $OUT/ab.log
  done
done
for v in 1 0; do
  echo "== small MDL_CG_BN_STATS=$v" | tee -a $OUT/ab.log
  MDL_CG_BN_STATS=$v timeout 300 python tools/bench_small.py 2>&1 | tail -1 | tee -a $OUT/ab.log
done
