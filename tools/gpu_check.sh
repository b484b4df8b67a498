#!/bin/bash
# Runs on the GPU box via gpurun: GPU tests, smoke, bench, rocprofv3 kernel trace of the same bench command.
# usage: tools/gpu_check.sh [tag] [bench args...]
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --maxfail=8 2>&1 | grep -vE "^\s*$" | tail -60 ) > $OUT/pytest.log
grep -E "passed|failed" $OUT/pytest.log | tail -3
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > $OUT/smoke.log; tail -2 $OUT/smoke.log
( timeout 900 python bench.py "$@" 2>&1 | tail -30 ) > $OUT/bench.log; tail -5 $OUT/bench.log
if [ "${PROFILE:-1}" = "1" ]; then
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1 )
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -40 "$f" > $OUT/kernel_stats_top.csv && head -14 $OUT/kernel_stats_top.csv | cut -c1-160
  find $OUT/prof -name "*kernel_trace.csv" -size +8M -delete
  find $OUT/prof -name "*.db" -delete
fi
