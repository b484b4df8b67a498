#!/bin/bash
# usage: tools/gpu_ab_time.sh <variant>...   — conv-kernel timings (tools/bench_kernels.py) of prebuilt library variants
# (matdeeplearn_amd/lib/variants/<name>.so from tools/build_variant.sh; "main" = the in-tree library)
OUT=gpurun_out/ab; mkdir -p $OUT
for v in "$@"; do
  echo "== $v"
  if [ "$v" = main ]; then unset MDL_HIP_LIB; else export MDL_HIP_LIB=$PWD/matdeeplearn_amd/lib/variants/$v.so; fi
  if [ -n "$AB_TEST" ]; then timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "$AB_TEST" 2>&1 | tail -2; fi
  timeout 300 python tools/bench_kernels.py --which fwd,bwd --iters ${AB_ITERS:-20} 2>&1 | grep -E "fwd|bwd|rror"
done 2>&1 | tee $OUT/ab.log
