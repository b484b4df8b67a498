#!/bin/bash
# samples sclk / power while bench.py runs its sustained legs
( for i in $(seq 1 200); do echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr -s ' \t' ' ' | tr '\n' ' ')"; sleep 0.25; done ) > $GRAFT_REPO_ROOT/gpurun_out/clocks.log 2>&1 &
W=$!
python bench.py --no-cpu-baseline --sustain-s 4 2>&1 | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/clock_bench.log
kill $W 2>/dev/null
awk '{print $1, $0}' $GRAFT_REPO_ROOT/gpurun_out/clocks.log | grep -oE "t=[0-9.]+|sclk clock level: [0-9]+: \([0-9]+Mhz\)|Power \(W\): [0-9.]+" | paste - - - | awk 'NR%2==0' | head -70
