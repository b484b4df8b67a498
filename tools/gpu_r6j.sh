#!/bin/bash
# round 6: per-phase cycle counters of the split-product (x3) forward and per-wave backward (-DMDL_CG_TIMING build), and the
# kernel stats of the bf16x3 step after the node kernel
set -u
TAG=${1:-r6j}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MDL_HIP_LIB=$GRAFT_REPO_ROOT/matdeeplearn_amd/lib/variants/x3t.so MDL_CG_EP=0 MDL_BK_SPLIT=1 timeout 300 python tools/bench_kernels.py --dtype fp32 --which fwd,bwd --iters 4 2>&1 | grep -E "^(fwd|bwd)|per-tile|rror" | cut -c1-900 | tee -a $OUT/log.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --dtype bf16x3 --no-extras --no-cpu-baseline --no-other-models --steps 6 --warmup 2 --settle-s 0.3 --settle-cap-s 1.0 > $OUT/x3.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" > $OUT/x3_kernel_stats.csv
rm -rf $OUT/prof
grep '^{"metric"' $OUT/x3.log | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('bf16x3 ms/step under rocprof', j['ms_per_step'])" | tee -a $OUT/log.txt
