#!/bin/bash
# The round's record run (via gpurun): GPU tests, smoke, the three bench legs, rocprofv3 kernel stats of each, HBM traffic
# counters of the conv kernels.  Everything lands under gpurun_out/<tag>/; copy the summaries to profiles/.
set -u
TAG=${1:-round}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --maxfail=8 2>&1 | grep -vE "^\s*$" | tail -40 ) > $OUT/pytest.log
grep -E "passed|failed" $OUT/pytest.log | tail -2
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $OUT/smoke.log; tail -1 $OUT/smoke.log
timeout 900 python bench.py 2>/dev/null | tail -1 > $OUT/bench_cgcnn.json; cut -c1-300 $OUT/bench_cgcnn.json
for m in schnet megnet; do timeout 900 python bench.py --model $m --cpu-steps 1 2>/dev/null | tail -1 > $OUT/bench_$m.json; cut -c1-200 $OUT/bench_$m.json; done
cd /tmp
for m in cgcnn schnet megnet; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$m -o t -- python $GRAFT_REPO_ROOT/bench.py --model $m --no-cpu-baseline --no-extras > $OUT/prof_$m.log 2>&1
  f=$(find $OUT/prof_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 "$f" > $OUT/kernel_stats_$m.csv
  grep -h '^{"metric"' $OUT/prof_$m.log > $OUT/bench_${m}_under_rocprof.json
  rm -rf $OUT/prof_$m
done
# the headline model at the reference's default width (config.yml:123): static 128-channel kernels on padded rows
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dim100 -o t -- python $GRAFT_REPO_ROOT/bench.py --dim 100 --no-cpu-baseline --no-extras --no-other-models > $OUT/prof_dim100.log 2>&1
f=$(find $OUT/prof_dim100 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 "$f" > $OUT/kernel_stats_cgcnn_dim100.csv
grep -h '^{"metric"' $OUT/prof_dim100.log > $OUT/bench_cgcnn_dim100_under_rocprof.json
rm -rf $OUT/prof_dim100
head -6 $OUT/kernel_stats_cgcnn.csv | cut -c1-140
cd $GRAFT_REPO_ROOT && PMC_ONLY=traffic bash tools/gpu_pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1; cat $OUT/pmc/hbm_traffic.json 2>/dev/null | head -30
