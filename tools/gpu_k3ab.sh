#!/bin/bash
# gpu_k3ab.sh <tag> <variants...>: selected GPU tests on the built library, then for every prebuilt variant
# (matdeeplearn_amd/lib/variants/<v>.so) the parity tests of the edge-per-lane backward and K2 / K3 timings on the bench batch
# (median of 20 launches), all in ONE box session (box-to-box spread is larger than most steps being measured).
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ -n "${PRE_TESTS:-}" ]; then
  ( timeout 900 python -m pytest $PRE_TESTS -m gpu -q --timeout 600 -s 2>&1 | grep -vE "^\s*$" | tail -120 ) > $OUT/pre_tests.log
  grep -E "passed|failed" $OUT/pre_tests.log | tail -2
fi
echo "== built library" | tee $OUT/ab.log
timeout 300 python tools/bench_kernels.py --which fwd,bwd --iters 20 2>&1 | grep -E "^(fwd|bwd|bwd_node):|rror" | tee -a $OUT/ab.log
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/matdeeplearn_amd/lib/variants/$v.so
  echo "== variant $v" | tee -a $OUT/ab.log
  case "${NO_PARITY:-}$v" in 1*|*_t) ;; *)
    MDL_HIP_LIB=$lib timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "variants_match_oracle or default_dispatch" 2>&1 | tail -1 | tee -a $OUT/ab.log ;;
  esac
  MDL_HIP_LIB=$lib timeout 300 python tools/bench_kernels.py --which fwd,bwd --iters 20 2>&1 | grep -E "^(fwd|bwd|bwd_node):|rror|ep2|per-tile" | cut -c1-900 | tee -a $OUT/ab.log
done
true
