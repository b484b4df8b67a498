#!/bin/bash
# round 6: repeatability — the full GPU suite twice more (fresh processes) and the driver's bench command twice more
set -u
TAG=${1:-r6q}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  ( timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -3 ) | tee -a $OUT/log.txt
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench$r.json
  python - $OUT/bench$r.json <<'PY' | tee -a $OUT/log.txt
import json,sys
j=json.load(open(sys.argv[1])); c=j["config"]
print("bench: ms/step", j["ms_per_step"], "value", j["value"], "by4", c["ms_per_step_by_4"], "mallocs", c["device_mallocs"], "k3", j["roofline"]["avg_launch_us"], j["roofline"]["frac"], "stale", j["roofline"]["traffic_source"]["stale"], "| sustained", j["sustained"]["eager"]["ms_per_step"], "| x3", j["bf16x3_mode"]["ms_per_step"], j["cpu_baseline"]["val_mae_delta_bf16x3"])
PY
done
