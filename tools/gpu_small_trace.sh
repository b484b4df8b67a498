#!/bin/bash
# one replayed step of the batch-100 training graph as an ordered kernel list with gaps (rocprofv3 --kernel-trace) -> gpurun_out/<tag>/small_trace.txt
set -u
TAG=${1:-small_trace}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/p -o t -- python $GRAFT_REPO_ROOT/tools/bench_small.py --steps 40 ${SMALL_ARGS:-} > $OUT/run.log 2>&1
f=$(find $OUT/p -name "*kernel_trace.csv" | head -1)
python - "$f" > $OUT/small_trace.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete step: from the last assemble_kernel back to the previous one
idx = [k for k, r in enumerate(rows) if "assemble_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
print("step of %d launches, %.1f us from first start to next step's first start" % (b - a, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  +gap %6.1f  dur %6.1f  q%s  g%s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r["Kernel_Name"][:110]))
    prev_end = max(prev_end, e)
PY
rm -rf $OUT/p
head -80 $OUT/small_trace.txt
