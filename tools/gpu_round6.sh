#!/bin/bash
# The record run of round 6 (via gpurun): full -m gpu suite, smoke, the default bench line (the driver's command), rocprofv3 kernel stats of the headline
# workload and of the SchNet leg, SQ + HBM-traffic counter passes of the conv kernels, kernel stats of the batch-100 replay.  Everything under gpurun_out/<tag>/.
set -u
TAG=${1:-r06}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
NO_EXP=1 NO_BENCH=1 bash tools/gpu_suite.sh $TAG
timeout 900 python bench.py 2> $OUT/bench_err.log | tail -1 > $OUT/bench_line.json; cut -c1-300 $OUT/bench_line.json
bash tools/gpu_prof.sh $TAG/prof_cgcnn | tee $OUT/prof_cgcnn.txt | head -12
bash tools/gpu_prof.sh $TAG/prof_schnet --model schnet --steps 20 --warmup 3 --settle-s 0.5 --settle-cap-s 3.0 | tee $OUT/prof_schnet.txt | head -8
bash tools/gpu_prof.sh $TAG/prof_x3 --dtype bf16x3 --steps 10 --warmup 3 --settle-s 0.5 --settle-cap-s 2.0 | tee $OUT/prof_x3.txt | head -10
cd $GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_small -o t -- python $GRAFT_REPO_ROOT/tools/bench_small.py > $OUT/small_under_rocprof.log 2>&1
f=$(find $OUT/prof_small -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -60 "$f" > $OUT/kernel_stats_small.csv
rm -rf $OUT/prof_small
cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_small.py --steps 600 2>&1 | tail -1 | tee $OUT/small.log
GRPS_SEL=4 bash tools/gpu_pmc.sh $TAG/pmc > $OUT/pmc.log 2>&1; cat $OUT/pmc/hbm_traffic.json 2>/dev/null | head -30
# K4 / K4b (round 6, second half): micro-benchmark of the three forms of the CFConv block and the counters of the two kernels
timeout 600 python tools/bench_cfconv_bwd.py 2>&1 | tail -5 | tee $OUT/bench_cfconv_bwd.log
bash tools/gpu_pmc_k4b.sh $TAG/pmc_k4b > /dev/null 2>&1; tail -12 $OUT/pmc_k4b/summary.txt | cut -c1-400
