#!/bin/bash
# headline bench with and without the cost-balanced node ranges of the backward (one box session)
for b in 1 0; do
  MDL_CG_BALANCE=$b python bench.py --no-cpu-baseline --no-other-models 2>/dev/null | tail -1 > /tmp/b.json
  python - $b <<'PY'
import json, sys
d = json.load(open("/tmp/b.json"))
print("balance", sys.argv[1], d["ms_per_step"], "K3", d["roofline"]["avg_launch_us"], "K2", d["roofline_other"]["avg_launch_us"],
      "replay", d["sustained"]["ms_per_step"], "eager", d["sustained"]["eager"]["ms_per_step"], "ref100", d["ref_batch_100"]["ms_per_step"])
PY
done
