#!/bin/bash
# fwd + bwd timings of prebuilt library variants on the bench batch
for v in "$@"; do echo "== variant $v"; MDL_HIP_LIB=$PWD/matdeeplearn_amd/lib/variants/$v.so timeout 300 python tools/bench_kernels.py --which fwd,bwd --iters 20 2>&1 | grep -E "^fwd|^bwd:" ; done | tee gpurun_out/variants2.log
