#!/bin/bash
# gpu_suite.sh [tag] [bench args]: the full -m gpu suite (all failures in full, slowest tests listed), the experiments build's
# tests, smoke() and one bench line; everything under gpurun_out/<tag>/
set -u
TAG=${1:-suite}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --maxfail=25 --durations=15 2>&1 | grep -vE "^\s*$" ) > $OUT/pytest.log
grep -E "passed|failed" $OUT/pytest.log | tail -2
[ -f $GRAFT_REPO_ROOT/experiments/lib/libmdl_hip_exp.so ] && [ -z "${NO_EXP:-}" ] && ( timeout 600 python -m pytest experiments -m gpu -q --timeout 600 2>&1 | tail -15 ) > $OUT/pytest_experiments.log && tail -1 $OUT/pytest_experiments.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $OUT/smoke.log; tail -1 $OUT/smoke.log
[ -z "${NO_BENCH:-}" ] && timeout 900 python bench.py "$@" 2> $OUT/bench_err.log | tail -1 > $OUT/bench.json && cut -c1-400 $OUT/bench.json
cp $GRAFT_REPO_ROOT/gpurun_out/rccl_world1.log $OUT/ 2>/dev/null
true
