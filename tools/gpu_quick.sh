#!/bin/bash
# usage: tools/gpu_quick.sh <tag> "<pytest args>" [bench args...]   — selected GPU tests (full log kept) + bench without the profiler
set -u
TAG=${1:-q}; PYT=${2:-tests}; shift 2 || true
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1200 python -m pytest $PYT -m gpu -q --timeout 600 --maxfail=6 2>&1 | grep -vE "^\s*$" ) > $OUT/pytest.log
grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.log | tail -12
grep -E "^E  " $OUT/pytest.log | head -30
( timeout 900 python bench.py "$@" 2>&1 | tail -5 ) > $OUT/bench.log; tail -2 $OUT/bench.log
