#!/bin/bash
# round 6, fourth lease: the fp32 (parity-mode) forward at one wave per SIMD (no spills) against the round-5 register allocation
set -u
TAG=${1:-r6d}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "cgconv or cgcnn or golden" 2>&1 | tail -3 | tee -a $OUT/log.txt
for r in 1 2; do
  for v in base f32w2; do
    if [ "$v" = base ]; then unset MDL_HIP_LIB; else export MDL_HIP_LIB=$GRAFT_REPO_ROOT/matdeeplearn_amd/lib/variants/$v.so; fi
    echo -n "$v: " | tee -a $OUT/log.txt
    timeout 300 python tools/bench_kernels.py --dtype fp32 --which fwd,bwd --iters 6 2>&1 | grep -E "^(fwd|bwd):|rror" | tr '\n' ' ' | tee -a $OUT/log.txt; echo | tee -a $OUT/log.txt
  done
done
unset MDL_HIP_LIB
timeout 600 python bench.py --dtype fp32 --no-extras --no-cpu-baseline --no-other-models --steps 6 --warmup 2 --settle-s 0.3 --settle-cap-s 1.0 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('fp32 step ms', j['ms_per_step'])" | tee -a $OUT/log.txt
