#!/bin/bash
# usage: tools/gpu_t.sh "<pytest -k expression>" [files...]  — selected GPU tests, failures in full
OUT=gpurun_out/t; mkdir -p $OUT; K="$1"; shift
timeout 1500 python -m pytest ${@:-tests} -m gpu -q --timeout 900 -k "$K" 2>&1 | grep -vE "^\s*$" > $OUT/pytest.log
grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.log | tail -15; grep -E "^E  " $OUT/pytest.log | head -40
