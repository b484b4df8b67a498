#!/bin/bash
# edge-per-lane kernel 2: parity subset, then bwd timings of the built library (MDL_CG_EP = 2 / 0) and of prebuilt variants (MDL_CG_EP=2)
OUT=gpurun_out/ep2; mkdir -p $OUT
bash tools/gpu_t.sh "edge_per_lane" tests/test_gpu_kernels.py
for ep in 2 0; do echo "== MDL_CG_EP=$ep"; MDL_CG_EP=$ep timeout 300 python tools/bench_kernels.py --which bwd --iters 20 2>&1 | grep -E "^bwd:|rror"; done 2>&1 | tee $OUT/ab.log
for v in "$@"; do echo "== variant $v"; MDL_CG_EP=2 MDL_HIP_LIB=$PWD/matdeeplearn_amd/lib/variants/$v.so timeout 300 python tools/bench_kernels.py --which bwd --iters 20 2>&1 | grep -E "^bwd:|rror"; done | tee $OUT/variants.log
