#!/bin/bash
# round 5, first lease: the new parity / reproducibility tests, a kernel trace of the small-batch replay, the full default bench line
set -u
TAG=${1:-r5a}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "megnet_matches_reference_goldens or nnconv_contraction or megnet_leg_reproduces or bench_distributed_path" -s 2>&1 | tail -25 > $OUT/new_tests.log
tail -8 $OUT/new_tests.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_small -o t -- python $GRAFT_REPO_ROOT/tools/bench_small.py > $OUT/small.log 2>&1
f=$(find $OUT/prof_small -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -70 "$f" > $OUT/kernel_stats_small.csv
rm -rf $OUT/prof_small
tail -2 $OUT/small.log
cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_small.py > $OUT/small_noprof.log 2>&1; tail -1 $OUT/small_noprof.log
timeout 900 python bench.py > $OUT/bench.log 2>&1
grep -h '^{"metric"' $OUT/bench.log > $OUT/bench_line.json
python - $OUT/bench_line.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("headline", j["ms_per_step"], "ms", j["value"], "by4", j["config"]["ms_per_step_by_4"], "settle", j["config"]["settle_steps"], j["config"]["settle_ms_per_step_by_8"])
print("roofline", {k: j["roofline"][k] for k in ("kernel", "frac", "avg_launch_us", "parts_avg_launch_us")})
print("edge", j["roofline_edge_pass"]["frac"], j["roofline_edge_pass"]["avg_launch_us"], "fwd", j["roofline_other"]["frac"], j["roofline_other"]["avg_launch_us"])
print("sustained", j.get("sustained", {}).get("ms_per_step"), j.get("sustained", {}).get("eager", {}).get("ms_per_step"), "ref100", j.get("ref_batch_100"))
for k, v in (j.get("other_models") or {}).items():
    print(k, {q: v.get(q) for q in ("ms_per_step", "ms_per_step_by_4", "settle_steps", "device_mallocs", "fp32_mode", "val_mae_delta", "val_mae_delta_bf16", "error")})
print("cpu", j.get("cpu_baseline"))
PY
