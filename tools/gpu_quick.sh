#!/bin/bash
# quick GPU iteration: kernel parity tests + bench (no profile)
TAG=${1:-q}; shift || true
OUT=gpurun_out/$TAG; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gpu_kernels.py -q --timeout 180 -x 2>&1 | grep -vE "^\s*$" | tail -30 ) > $OUT/pytest.log; grep -E "passed|failed|Error" $OUT/pytest.log | tail -5
( timeout 900 python bench.py "$@" 2>&1 | tail -30 ) > $OUT/bench.log; tail -3 $OUT/bench.log
