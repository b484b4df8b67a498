#!/bin/bash
# round 5 closing check: the full suite once (the log is the one committed under profiles/), the two order-noise-sensitive tests
# five more times each, then the two HBM counter passes that stamp profiles/hbm_traffic.json with the final kernel sources
set -u
TAG=${1:-r05g}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
( echo "# full pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --maxfail=25 --durations=8 2>&1 | grep -vE "^\s*$" | tail -40 ) > $OUT/pytest_1.log
grep -E "passed|failed" $OUT/pytest_1.log | tail -1
for i in 1 2 3 4 5; do
  timeout 600 python -m pytest tests/test_gpu_kernels.py::test_weight_gradients_written_in_place_match_the_assembled_ones tests/test_gpu_workloads.py::test_cfg4_megnet_bf16_training_tracks_fp32 -q -m gpu 2>&1 | tail -1
done | tee $OUT/repeat.log
PMC_ONLY=traffic bash tools/gpu_pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1
tail -5 $OUT/pmc.log; cat $OUT/pmc/hbm_traffic.json | head -30
