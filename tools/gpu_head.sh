#!/bin/bash
# headline bench with and without the fused post-FC head (one box session): timed steps, 2-s sustained eager / replay legs
for b in ${@:-1 0}; do
  MDL_MLP_HEAD=$b python bench.py --no-cpu-baseline --no-other-models 2>/dev/null | tail -1 > /tmp/b.json
  python - $b <<'PY'
import json, sys
d = json.load(open("/tmp/b.json"))
print("mlp_head", sys.argv[1], d["ms_per_step"], "eager", d["sustained"]["eager"]["ms_per_step"], "replay", d["sustained"]["ms_per_step"],
      "ref100", d["ref_batch_100"]["ms_per_step"], "K3", d["roofline"]["avg_launch_us"])
PY
done
