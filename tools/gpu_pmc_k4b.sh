#!/bin/bash
# HBM bytes of the K4 kernel by store mode (FETCH_SIZE / WRITE_SIZE, separate passes): is a partial-line store a read-modify-write?
TAG=${1:-pmck4b}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/tools/bench_cfconv.py --iters 1 > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "cfconv_fwd_kernel" in r["Kernel_Name"]]
print(rows[0]["Counter_Name"], "KiB per launch, in launch order:", [round(float(r["Counter_Value"])) for r in rows])
PY
  rm -rf $OUT/p$i
done 2>&1 | tee $OUT/summary.txt
