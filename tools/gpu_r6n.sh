#!/bin/bash
# round 6: where the exact-fp32 steps of MEGNet / SchNet / MPNN spend their time (kernel stats, 4 timed steps each)
set -u
TAG=${1:-r6n}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for m in megnet schnet mpnn; do
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$m -o t -- python $GRAFT_REPO_ROOT/bench.py --model $m --dtype fp32 --no-extras --no-cpu-baseline --no-other-models --steps 4 --warmup 1 --settle-s 0 > $OUT/$m.log 2>&1
f=$(find $OUT/p_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" > $OUT/${m}_fp32_kernel_stats.csv
rm -rf $OUT/p_$m
grep '^{"metric"' $OUT/$m.log | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$m fp32 ms/step', j['ms_per_step'])" | tee -a $OUT/log.txt
done
