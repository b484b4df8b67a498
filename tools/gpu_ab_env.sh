#!/bin/bash
# gpu_ab_env.sh <tag> <ENVVAR> [reps]: headline bench (40 steps, no extras) and the batch-100 replay with ENVVAR=1 / 0 alternating in
# ONE box session (box-to-box spread is larger than most steps being measured); optional PRE_TESTS="<pytest -k expr>" first
set -u
TAG=$1; VAR=$2; REPS=${3:-2}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ -n "${PRE_TESTS:-}" ]; then timeout 1200 python -m pytest tests -m gpu -q -x -k "$PRE_TESTS" 2>&1 | tail -25 > $OUT/pytest.log; tail -3 $OUT/pytest.log; fi
for rep in $(seq $REPS); do
  for v in 1 0; do
    echo "== $VAR=$v" | tee -a $OUT/ab.log
    env $VAR=$v timeout 300 python bench.py --no-extras --no-cpu-baseline --no-other-models --steps 40 2>&1 | grep '^{"metric"' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['ms_per_step'], j['config']['ms_per_step_by_4'], 'K3', r['avg_launch_us'], r['parts_avg_launch_us'], 'K2', j['roofline_other']['avg_launch_us'])" | tee -a $OUT/ab.log
    env $VAR=$v timeout 300 python tools/bench_small.py --steps 600 2>&1 | tail -1 | tee -a $OUT/ab.log
  done
done
