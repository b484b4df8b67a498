#!/bin/bash
# The round's record run (via gpurun): full -m gpu suite, experiments tests, smoke, the default bench line, rocprofv3 kernel stats of
# the headline workload and of the schnet / megnet / mpnn / dim-100 legs, SQ + HBM-traffic counter passes of the conv kernels.
# Everything lands under gpurun_out/<tag>/; copy the summaries to profiles/.
set -u
TAG=${1:-r04}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/gpu_suite.sh $TAG
bash tools/gpu_prof.sh $TAG/prof_cgcnn | tee $OUT/prof_cgcnn.txt | head -3
for m in schnet megnet mpnn; do
  EXTRA="" bash tools/gpu_model_prof.sh $m > $OUT/prof_$m.txt 2>&1; head -2 $OUT/prof_$m.txt
  cp gpurun_out/mprof_$m/kernel_stats.csv $OUT/kernel_stats_$m.csv 2>/dev/null
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dim100 -o t -- python $GRAFT_REPO_ROOT/bench.py --dim 100 --no-cpu-baseline --no-extras --no-other-models > $OUT/prof_dim100.log 2>&1
f=$(find $OUT/prof_dim100 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 "$f" > $OUT/kernel_stats_cgcnn_dim100.csv
grep -h '^{"metric"' $OUT/prof_dim100.log > $OUT/bench_cgcnn_dim100_under_rocprof.json
rm -rf $OUT/prof_dim100
cd $GRAFT_REPO_ROOT
GRPS_SEL=4 bash tools/gpu_pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1; cat $OUT/pmc/hbm_traffic.json 2>/dev/null | head -30
