#!/bin/bash
# K4 (fused CFConv forward): its tests, the SchNet model / workload tests, then the SchNet bench leg fused against three-pass
set -u
TAG=${1:-k4}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "cfconv" 2>&1 | grep -vE "^\s*$" | tail -${TAILN:-25} | tee $OUT/pytest_k4.log
[ -n "${ONLY_K4:-}" ] && exit 0
[ -z "${SKIP_MODEL:-}" ] && timeout 900 python -m pytest tests -m gpu -q -k "schnet or SchNet or ensemble or wrappers or replay" 2>&1 | tail -6 | tee $OUT/pytest_schnet.log
for rep in $(seq ${REPS:-1}); do
  for f in 1 0; do
    echo -n "MDL_CFCONV_FUSED=$f: " | tee -a $OUT/ab.log
    MDL_CFCONV_FUSED=$f timeout 600 python bench.py --model schnet --steps 20 --warmup 3 --settle-s 0.5 --settle-cap-s 3.0 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['config'].get('ms_per_step_by_4'))" | tee -a $OUT/ab.log
  done
done
