#!/bin/bash
# round 6, first lease: does the driver's 20-step window see the steady state now?  (bench line with host enqueue times per step,
# device mallocs, by-4 device times; the same with two warm-up steps enqueued behind the barrier), the parity subset the bench /
# ops.configure changes touch, kernel baselines of the box
set -u
TAG=${1:-r6a}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/log.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "cgconv or dispatch" 2>&1 | tail -3 | tee -a $OUT/log.txt
for i in 1 2; do
  echo "== bench headline only, run $i" | tee -a $OUT/log.txt
  timeout 600 python bench.py --no-extras --no-cpu-baseline --no-other-models 2>/dev/null | grep '^{"metric"' | tee $OUT/head$i.json | python -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['config']
print('ms/step', j['ms_per_step'], 'dev', c['device_ms_per_step'], 'by4', c['ms_per_step_by_4'], 'mallocs', c['device_mallocs'], 'settle', c['settle_steps'], c['settle_ms_per_step_by_8'])
print('host enqueue', c['host_enqueue_ms_per_step'])
print('k3', j['roofline']['avg_launch_us'], j['roofline']['parts_avg_launch_us'], 'k2', j['roofline_other']['avg_launch_us'])" | tee -a $OUT/log.txt
done
echo "== bench headline only, --run-in 2" | tee -a $OUT/log.txt
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-other-models --run-in 2 2>/dev/null | grep '^{"metric"' | tee $OUT/head_runin.json | python -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['config']
print('ms/step', j['ms_per_step'], 'dev', c['device_ms_per_step'], 'by4', c['ms_per_step_by_4'], 'mallocs', c['device_mallocs'])
print('host enqueue', c['host_enqueue_ms_per_step'])" | tee -a $OUT/log.txt
echo "== kernels" | tee -a $OUT/log.txt
timeout 300 python tools/bench_kernels.py --which fwd,bwd --iters 20 2>&1 | grep -E "^(N=|fwd|bwd|bwd_node):|rror" | tee -a $OUT/log.txt
echo "== full default bench (driver command)" | tee -a $OUT/log.txt
timeout 1500 python bench.py > $OUT/bench_full.log 2>&1
grep '^{"metric"' $OUT/bench_full.log > $OUT/bench_line.json
python - <<'PY' | tee -a $OUT/log.txt
import json,os
j=json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out",os.environ.get("TAG","r6a"),"bench_line.json")))
c=j["config"]
print("FULL ms/step", j["ms_per_step"], "value", j["value"], "by4", c["ms_per_step_by_4"], "mallocs", c["device_mallocs"])
print("sustained eager", j["sustained"]["eager"]["ms_per_step"], "replay", j["sustained"].get("ms_per_step"), "ref100", j["ref_batch_100"].get("ms_per_step"), "fp32", j["fp32_mode"]["ms_per_step"])
for k,v in j["other_models"].items(): print(k, v.get("ms_per_step"), v.get("ms_per_step_by_4"), v.get("device_mallocs"), v.get("error"))
print(j["cpu_baseline"])
PY
timeout 600 python -m pytest tests/test_gpu_distributed.py -m gpu -q -x -k "bench_distributed or driver_launcher" 2>&1 | tail -3 | tee -a $OUT/log.txt
