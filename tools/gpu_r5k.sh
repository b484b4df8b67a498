#!/bin/bash
set -u
TAG=${1:-r5k}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "fused_post_fc_head or model or replay or padded or deterministic or rccl" 2>&1 | tail -20 > $OUT/pytest.log; tail -3 $OUT/pytest.log
for i in 1 2; do timeout 300 python tools/bench_small.py --steps 600 2>&1 | tail -1 | tee -a $OUT/ab.log; done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_small -o t -- python $GRAFT_REPO_ROOT/tools/bench_small.py > $OUT/small.log 2>&1
f=$(find $OUT/prof_small -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -70 "$f" > $OUT/kernel_stats_small.csv
rm -rf $OUT/prof_small
