#!/bin/bash
# K4b (CFConv backward with the filter recomputed): its tests, the micro-benchmark, the SchNet tests, then the SchNet bench leg in both forms
set -u
TAG=${1:-r6w}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "cfconv" 2>&1 | grep -vE "^\s*$" | tail -${TAILN:-30} | tee $OUT/pytest_k4b.log
timeout 600 python tools/bench_cfconv_bwd.py 2>&1 | tail -8 | tee $OUT/bench_cfconv_bwd.log
[ -n "${ONLY_K4:-}" ] && exit 0
timeout 1200 python -m pytest tests -m gpu -q -k "schnet or SchNet or ensemble or wrappers or replay" 2>&1 | tail -8 | tee $OUT/pytest_schnet.log
for f in 1 0; do
  echo -n "cfconv_recompute=$f: " | tee -a $OUT/ab.log
  timeout 600 python bench.py --model schnet --steps 20 --warmup 3 --settle-s 0.5 --settle-cap-s 3.0 --no-extras --ops cfconv_recompute=$f 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['config'].get('ms_per_step_by_4'))" | tee -a $OUT/ab.log
done
