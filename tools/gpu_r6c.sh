#!/bin/bash
# round 6, third lease: K3 reducer variants (tile loop of the reducers unrolled / e fragments requested at the top of the tile
# step), what the model-level golden gradient checks deliver in fp32, the SchNet leg under the profiler (K4's final kernel)
set -u
TAG=${1:-r6c}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in uj ebt ujebt; do
  echo "== parity of variant $v" | tee -a $OUT/log.txt
  MDL_HIP_LIB=$GRAFT_REPO_ROOT/matdeeplearn_amd/lib/variants/$v.so timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "variants_match_oracle or default_dispatch" 2>&1 | tail -1 | tee -a $OUT/log.txt
done
bash tools/gpu_k3rep.sh $TAG 3 uj ebt ujebt 2>&1 | tee -a $OUT/log.txt
timeout 600 python tools/dbg/golden_grad_errors.py 2>&1 | tail -6 | tee -a $OUT/log.txt
EXTRA="" bash tools/gpu_model_prof.sh schnet 2>&1 | tee -a $OUT/log.txt
cp $GRAFT_REPO_ROOT/gpurun_out/mprof_schnet/kernel_stats.csv $OUT/schnet_kernel_stats.csv 2>/dev/null
grep -h '^{"metric"' $GRAFT_REPO_ROOT/gpurun_out/mprof_schnet/p.log > $OUT/schnet_line_under_rocprof.json
rm -rf $GRAFT_REPO_ROOT/gpurun_out/mprof_schnet/p
