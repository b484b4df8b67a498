#!/bin/bash
# kernel trace of a few bench steps, reduced to one line per launch (name, grid, duration) in launch order: gpurun_out/trace/step.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 "$@" > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python - "$f" > $OUT/step.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
for r in rows[-(n // 3 + 10):]:          # the last step (of 3 executed)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%8.1f us  grid %-10s wg %-5s %s" % (d, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r["Kernel_Name"][:110]))
PY
rm -rf $OUT/prof
wc -l $OUT/step.txt
