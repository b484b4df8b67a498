#!/bin/bash
# kernel traces of the headline with and without the BatchNorm statistics in the K2 epilogue + the rest of the GPU suite
set -u
TAG=${1:-r5d}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/pytest.log; tail -5 $OUT/pytest.log
for v in 1 0 1 0; do
  cd /tmp
  MDL_CG_BN_STATS=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$v -o t -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-other-models --steps 40 > $OUT/prof$v.log 2>&1
  f=$(find $OUT/prof$v -name "*kernel_stats.csv" | head -1)
  echo "== MDL_CG_BN_STATS=$v $(grep -h '^{"metric"' $OUT/prof$v.log | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])")" | tee -a $OUT/ab.log
  python - "$f" <<'PY' | tee -a $OUT/ab.log
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = max([int(r["Calls"]) for r in rows if "assemble_kernel" in r["Name"]] + [1])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step (%d steps): %.3f ms" % (steps, tot / steps / 1e6))
for r in rows[:14]:
    print("%6.2f%%  %5.2f/step  avg %8.1f us  %s" % (float(r["Percentage"]), int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
  rm -rf $OUT/prof$v
done
