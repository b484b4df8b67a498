#!/bin/bash
# full suite + the default bench line (what the driver runs)
set -u
TAG=${1:-r5j}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $OUT/pytest.log; tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench.log 2>&1
grep -h '^{"metric"' $OUT/bench.log > $OUT/bench_line.json
python - $OUT/bench_line.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("headline", j["ms_per_step"], "ms", j["value"], "by4", j["config"]["ms_per_step_by_4"], "settle", j["config"]["settle_steps"])
print("roofline", {k: j["roofline"][k] for k in ("frac", "avg_launch_us", "parts_avg_launch_us")}, "edge", j["roofline_edge_pass"]["frac"], "fwd", j["roofline_other"]["frac"], j["roofline_other"]["avg_launch_us"], "step", j["step_roofline"]["frac"])
print("sustained", j.get("sustained", {}).get("ms_per_step"), j.get("sustained", {}).get("eager", {}).get("ms_per_step"), "ref100", {k: v for k, v in j.get("ref_batch_100", {}).items() if k in ("ms_per_step", "eager")}, "fp32", j.get("fp32_mode", {}).get("ms_per_step"))
for k, v in (j.get("other_models") or {}).items():
    print(k, {q: v.get(q) for q in ("ms_per_step", "ms_per_step_by_4", "device_mallocs", "fp32_mode", "val_mae_delta", "val_mae_delta_bf16", "error")})
cb = j.get("cpu_baseline") or {}
print("cpu", {k: cb.get(k) for k in ("value", "cores", "val_mae_delta", "val_mae_delta_bf16", "pred_max_rel_delta_bf16")})
PY
