#!/bin/bash
# final check of round 5: the full suite twice (flakiness), smoke, one default bench line
set -u
TAG=${1:-r05f}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
git_head=$(cat .git/HEAD 2>/dev/null || echo "n/a")
for i in 1 2; do
  ( echo "# full pytest -m gpu, run $i"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --maxfail=25 --durations=8 2>&1 | grep -vE "^\s*$" | tail -30 ) > $OUT/pytest_$i.log
  grep -E "passed|failed" $OUT/pytest_$i.log | tail -1
done
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) | tee $OUT/smoke.log
timeout 900 python bench.py 2> $OUT/bench_err.log | tail -1 > $OUT/bench_line.json
python - $OUT/bench_line.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("headline", j["ms_per_step"], j["value"], j["config"]["ms_per_step_by_4"], "K3", j["roofline"]["frac"], j["roofline"]["avg_launch_us"], j["roofline"]["traffic_source"]["stale"], "K2", j["roofline_other"]["avg_launch_us"])
print("sustained", j["sustained"]["ms_per_step"], j["sustained"]["eager"]["ms_per_step"], "ref100", j["ref_batch_100"]["ms_per_step"])
print({k: v.get("ms_per_step") for k, v in j["other_models"].items()})
PY
