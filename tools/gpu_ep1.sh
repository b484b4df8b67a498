#!/bin/bash
# check of the edge-per-lane backward: parity tests (subset unless FULL=1), A/B timing vs the per-wave kernel, phase timing
OUT=gpurun_out/ep1; mkdir -p $OUT
export TMPDIR=/tmp
if [ "${FULL:-0}" = "1" ]; then K="cgconv or bulk or cfg2"; else K="cgconv_matches_oracle or sum_aggr or c_abi or permutation"; fi
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_workloads.py -m gpu -q -x -k "$K" --timeout 600 2>&1 | tail -25 ) > $OUT/pytest.log; tail -8 $OUT/pytest.log
for ep in ${EPS:-2 1 0}; do echo "== MDL_CG_EP=$ep"; MDL_CG_EP=$ep timeout 300 python tools/bench_kernels.py --which fwd,bwd --iters 20 2>&1 | grep -E "N=|bwd|rror"; done 2>&1 | tee $OUT/ab.log
for v in "$@"; do echo "== variant $v"; MDL_HIP_LIB=$PWD/matdeeplearn_amd/lib/variants/$v.so timeout 300 python tools/bench_kernels.py --which fwd,bwd --iters 10 2>&1 | grep -E "bwd" | grep -v node; done | tee $OUT/variants.log
