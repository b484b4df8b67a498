#!/bin/bash
# timings of prebuilt library variants on the bench batch (bwd only)
for v in "$@"; do echo "== variant $v"; MDL_HIP_LIB=$PWD/matdeeplearn_amd/lib/variants/$v.so timeout 300 python tools/bench_kernels.py --which fwd,bwd --iters 10 2>&1 | grep -E "^bwd" | grep -v node; done | tee gpurun_out/variants.log
