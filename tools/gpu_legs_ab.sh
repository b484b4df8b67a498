#!/bin/bash
# gpu_legs_ab.sh <tag> <ENV=VAL|-> <models...>: bench legs (ms/step) with and without an environment switch in ONE box session
# e.g. tools/gpu_legs_ab.sh l1 MDL_DENSE_BWD=0 megnet schnet mpnn
set -u
TAG=$1; SW=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for m in "$@"; do
  for mode in default "$SW"; do
    [ "$mode" = "-" ] && continue
    pre=""; [ "$mode" != "default" ] && pre="env $mode"
    $pre timeout 600 python bench.py --model $m --no-cpu-baseline --no-extras --no-other-models --steps 20 --warmup 5 2> $OUT/err_${m}.log | tail -1 > $OUT/bench_${m}_${mode%%=*}.json
    python - $OUT/bench_${m}_${mode%%=*}.json "$m" "$mode" <<'PY' | tee -a $OUT/legs.log
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-8s %-22s %8.3f ms/step  %.3e %s" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], d["unit"]))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done
true
