#!/bin/bash
# weight gradients written in place (no assembly kernel): parity tests, bench, small batch; small-batch ablations of the per-wave K3
set -u
TAG=${1:-r5e}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "not distributed" 2>&1 | tail -30 > $OUT/pytest.log; tail -4 $OUT/pytest.log
run() { local lab=$1 lib=$2; shift 2
  echo "== $lab $*" | tee -a $OUT/ab.log
  if [ "$lib" = "-" ]; then timeout 300 python tools/bench_kernels.py "$@" 2>&1 | grep -E "^(N=|fwd|bwd|bwd_node):|rror" | tee -a $OUT/ab.log
  else MDL_HIP_LIB=$GRAFT_REPO_ROOT/matdeeplearn_amd/lib/variants/$lib.so timeout 300 python tools/bench_kernels.py "$@" 2>&1 | grep -E "^(N=|fwd|bwd|bwd_node):|rror" | tee -a $OUT/ab.log; fi
}
run base-small - --which fwd,bwd --iters 40 --graphs 100
timeout 300 python tools/bench_small.py 2>&1 | tail -1 | tee -a $OUT/ab.log
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-other-models 2>&1 | grep '^{"metric"' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['ms_per_step'], j['config']['ms_per_step_by_4'], 'K3', r['avg_launch_us'], r['parts_avg_launch_us'], 'K2', j['roofline_other']['avg_launch_us'])" | tee -a $OUT/ab.log
