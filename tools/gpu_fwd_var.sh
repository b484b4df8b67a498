#!/bin/bash
# K2 timings of prebuilt library variants (one box session)
for v in "$@"; do echo "== variant $v"; MDL_HIP_LIB=$PWD/matdeeplearn_amd/lib/variants/$v.so timeout 300 python tools/bench_kernels.py --which fwd --iters 40 2>&1 | grep -E "^fwd:|rror"; done
