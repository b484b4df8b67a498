#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes with the kernel trace only) of the streaming dense kernels over
# tools/bench_dense.py: what the one-pass dense backward reads and writes against the pair it replaces -> gpurun_out/<tag>/
TAG=${1:-pmc_dense}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ROWS=${ROWS:-1498398}
i=0
for grp in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/tools/bench_dense.py $ROWS > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $grp <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if not k.startswith(("void mdl::gemm_tn_stream_kernel", "void mdl::linear_act_kernel")):
        continue
    short = re.sub(r"\(.*", "", k.replace("void mdl::", ""))
    agg[(short, int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))].append(float(r["Counter_Value"]))
for (k, g), v in sorted(agg.items()):
    print("%s %-58s launches %3d  avg %10.1f KiB" % (sys.argv[2], k, len(v), sum(v) / len(v)))
PY
  find $OUT/p$i -name "*.csv" -size +5M -delete
done 2>&1 | tee $OUT/summary.txt
