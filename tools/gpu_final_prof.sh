set -u
TAG=r04c; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/gpu_prof.sh $TAG/prof_cgcnn | tee $OUT/prof_cgcnn.txt | head -4
for m in schnet megnet mpnn; do
  EXTRA="" bash tools/gpu_model_prof.sh $m > $OUT/prof_$m.txt 2>&1; head -2 $OUT/prof_$m.txt
  cp gpurun_out/mprof_$m/kernel_stats.csv $OUT/kernel_stats_$m.csv 2>/dev/null
done
