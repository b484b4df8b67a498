#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/padprof; mkdir -p $OUT
python -m pytest tests/test_gpu_training.py -m gpu -q -k graph 2>&1 | grep -E "^E  " | head -12 | cut -c1-600
cd /tmp
MODE=padded timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o t -- python $GRAFT_REPO_ROOT/tools/dbg/padded_body.py > $OUT/p.log 2>&1
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1); head -25 "$f" | cut -d, -f1-5 | cut -c1-150
MODE=plain timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q -o t -- python $GRAFT_REPO_ROOT/tools/dbg/padded_body.py > $OUT/q.log 2>&1
f=$(find $OUT/q -name "*kernel_stats.csv" | head -1); head -25 "$f" | cut -d, -f1-5 | cut -c1-150
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
