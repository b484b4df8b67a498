#!/bin/bash
# round 6, second lease: what allocates inside the timed region?  Not cyclic garbage (tools/dbg/gc_cycles.py: no tensor among the
# unreachable objects) — the sustained leg shows one hipMalloc per ~10 steps for ever, i.e. the caching allocator fragments on
# batch sizes that differ by a per cent.  Allocator configurations A/B; the collector modes; where the fp32 (parity) step spends 26 ms
set -u
TAG=${1:-r6b}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
line() { python -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['config']; s=j.get('sustained',{}).get('eager',{})
print('ms/step', j['ms_per_step'], 'by4', c['ms_per_step_by_4'], 'mallocs', c['device_mallocs'], 'settle', c['settle_ms_per_step_by_8'][-2:], 'k3', j['roofline']['avg_launch_us'], '| sustained', s.get('ms_per_step'), 'mallocs', s.get('device_mallocs'), 'steps', s.get('steps'))"; }
for conf in "roundup_power2_divisions:8" "roundup_power2_divisions:2" "roundup_power2_divisions:1" "expandable_segments:True" "roundup_power2_divisions:4"; do
  echo "== allocator $conf" | tee -a $OUT/log.txt
  PYTORCH_ALLOC_CONF=$conf PYTORCH_HIP_ALLOC_CONF=$conf PYTORCH_CUDA_ALLOC_CONF=$conf timeout 600 python bench.py --no-cpu-baseline --no-other-models 2>$OUT/err.txt | grep '^{"metric"' | line | tee -a $OUT/log.txt
  tail -2 $OUT/err.txt | cut -c1-300 >> $OUT/log.txt
done
for g in off freeze on; do
  echo "== bench headline only, --collector $g" | tee -a $OUT/log.txt
  timeout 600 python bench.py --no-extras --no-cpu-baseline --no-other-models --collector $g 2>/dev/null | grep '^{"metric"' | line | tee -a $OUT/log.txt
done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof32 -o t -- python $GRAFT_REPO_ROOT/bench.py --dtype fp32 --no-extras --no-cpu-baseline --no-other-models --steps 6 --warmup 2 --settle-s 0.3 --settle-cap-s 1.0 > $OUT/fp32.log 2>&1
f=$(find $OUT/prof32 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" > $OUT/fp32_kernel_stats.csv
rm -rf $OUT/prof32
grep '^{"metric"' $OUT/fp32.log | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('fp32 ms/step', j['ms_per_step'])" | tee -a $OUT/log.txt
