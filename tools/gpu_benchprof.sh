#!/bin/bash
# bench.py under rocprofv3 kernel trace: per-kernel totals of one profiled run (no tests)
TAG=${1:-bp}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 900 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 > $OUT/bench.json; cut -c1-300 $OUT/bench.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f" > $OUT/kernel_stats_top.csv && python - $OUT/kernel_stats_top.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:26]:
    print("%-70s calls %5s avg %8.1f us  total %8.2f ms  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
find $OUT/prof -name "*kernel_trace.csv" -size +8M -delete; find $OUT/prof -name "*.db" -delete
