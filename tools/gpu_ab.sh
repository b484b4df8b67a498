#!/bin/bash
# A/B the kernel micro-benchmark over prebuilt library variants
OUT=gpurun_out/ab; mkdir -p $OUT
W=${WHICH:-fwd}
for v in "$@"; do
  echo "== $v"
  MDL_HIP_LIB=$PWD/matdeeplearn_amd/lib/variants/$v.so timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "cgconv" 2>&1 | tail -1
  MDL_HIP_LIB=$PWD/matdeeplearn_amd/lib/variants/$v.so timeout 300 python tools/bench_kernels.py --which $W 2>&1 | grep -E "fwd|bwd|rror"
done 2>&1 | tee $OUT/ab.log
