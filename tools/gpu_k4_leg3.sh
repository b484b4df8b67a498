#!/bin/bash
# the fused SchNet leg three times (one-slow-group check)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
python bench.py --model schnet --steps 20 --warmup 3 --settle-s 0.5 --settle-cap-s 3.0 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config'].get('ms_per_step_by_4'), 'mallocs', j['config'].get('device_mallocs'))"
done
