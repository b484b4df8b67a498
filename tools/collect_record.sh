#!/bin/bash
# collect_record.sh <gpurun tag> <profiles prefix>: copy the summaries of a tools/gpu_round6.sh record run from gpurun_out/<tag>/ into profiles/
# (tracked) under <prefix>_*; hbm_traffic.json (hash-stamped PMC traffic of the conv kernels) replaces the previous one
set -eu
T=gpurun_out/$1; P=profiles/$2
cp $T/bench_line.json ${P}_bench_line.json
cp $T/prof_cgcnn/kernel_stats.csv ${P}_bench_kernel_stats.csv
cp $T/prof_cgcnn/bench_under_rocprof.json ${P}_bench_line_under_rocprof.json
cp $T/prof_schnet/kernel_stats.csv ${P}_schnet_kernel_stats.csv
cp $T/prof_schnet/bench_under_rocprof.json ${P}_bench_line_schnet_under_rocprof.json
cp $T/prof_x3/kernel_stats.csv ${P}_bf16x3_kernel_stats.csv
cp $T/kernel_stats_small.csv ${P}_small_batch_kernel_stats.csv
cp $T/pmc/summary.txt ${P}_pmc_sq_summary.txt
cp $T/pmc_k4b/summary.txt ${P}_k4b_pmc_summary.txt
cp $T/pmc/hbm_traffic.json profiles/hbm_traffic.json
( tail -40 $T/pytest.log; cat $T/smoke.log; cat $T/small.log ) > ${P}_pytest.log
[ -f $T/small_trace.txt ] && cp $T/small_trace.txt ${P}_small_batch_trace.txt
ls -la ${P}_* | awk '{print $5, $9}'
