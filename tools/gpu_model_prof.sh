#!/bin/bash
export TMPDIR=/tmp
for m in "$@"; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/mprof_$m; mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o t -- python $GRAFT_REPO_ROOT/bench.py --model $m --no-cpu-baseline --no-extras --steps 5 --warmup 2 $EXTRA > $OUT/p.log 2>&1
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
steps=max([int(r['Calls']) for r in rows if 'assemble_kernel' in r['Name']] + [1])      # one batch assembly per step (settle + warm-up + timed)
print("total kernel time per step (%d steps): %.2f ms"%(steps, tot/steps/1e6))
for r in rows[:16]:
    print("%6.2f%%  calls %5s  avg %9.1f us  %s"%(float(r['Percentage']), r['Calls'], float(r['AverageNs'])/1e3, r['Name'][:110]))
PY
find $OUT/p -name "*kernel_trace.csv" -delete; find $OUT/p -name "*.db" -delete
done
