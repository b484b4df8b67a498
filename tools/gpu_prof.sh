#!/bin/bash
# gpu_prof.sh <tag> [bench args]: rocprofv3 --kernel-trace --stats of bench.py --no-extras (headline workload only), the
# kernel_stats table (top 45) and the bench line printed under the profiler -> gpurun_out/<tag>/
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-other-models "$@" > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -46 "$f" > $OUT/kernel_stats.csv
grep -h '^{"metric"' $OUT/prof.log > $OUT/bench_under_rocprof.json
rm -rf $OUT/prof
python - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = max([int(r["Calls"]) for r in rows if "assemble_kernel" in r["Name"]] + [1])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step (%d steps): %.3f ms" % (steps, tot / steps / 1e6))
for r in rows[:32]:
    print("%6.2f%%  %5.1f/step  avg %8.1f us  %s" % (float(r["Percentage"]), int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
