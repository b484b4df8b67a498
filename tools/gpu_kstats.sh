#!/bin/bash
# per-kernel average durations of tools/bench_kernels.py (rocprofv3 kernel trace)
OUT=$GRAFT_REPO_ROOT/gpurun_out/ks; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o t -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --iters 5 "$@" > $OUT/log.txt 2>&1
python - <<'PY'
import csv, os
f = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/ks/p/t_kernel_stats.csv")
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
