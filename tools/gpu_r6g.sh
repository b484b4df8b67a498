#!/bin/bash
# round 6: where the bf16x3 step spends its time (kernel stats), and whether torch.mm(bf16, bf16, out_dtype=fp32) exists here
set -u
TAG=${1:-r6g}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | tail -8 | tee -a $OUT/log.txt
import torch, time
a = torch.randn(209409, 256, device="cuda").bfloat16(); b = torch.randn(256, 64, device="cuda").bfloat16()
try:
    c = torch.mm(a, b, out_dtype=torch.float32)
    print("torch.mm out_dtype ok:", c.dtype, c.shape)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): c = torch.mm(a, b, out_dtype=torch.float32)
    torch.cuda.synchronize(); print("mm [N,256]x[256,64] bf16->f32: %.1f us" % ((time.perf_counter()-t)/20*1e6))
    x = torch.randn(209409, 64, device="cuda").bfloat16()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): d = torch.mm(a.t(), x, out_dtype=torch.float32)
    torch.cuda.synchronize(); print("mm [256,N]x[N,64] bf16->f32: %.1f us" % ((time.perf_counter()-t)/20*1e6))
except Exception as e:
    print("torch.mm out_dtype FAILED:", repr(e)[:300])
af = torch.randn(209409, 256, device="cuda"); bf_ = torch.randn(256, 64, device="cuda"); g = torch.randn(209409, 64, device="cuda")
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): c = torch.addmm(g, af, bf_)
torch.cuda.synchronize(); print("addmm fp32 [N,256]x[256,64]: %.1f us" % ((time.perf_counter()-t)/20*1e6))
PY
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --dtype bf16x3 --no-extras --no-cpu-baseline --no-other-models --steps 6 --warmup 2 --settle-s 0.3 --settle-cap-s 1.0 > $OUT/x3.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" > $OUT/x3_kernel_stats.csv
rm -rf $OUT/prof
grep '^{"metric"' $OUT/x3.log | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('bf16x3 ms/step under rocprof', j['ms_per_step'])" | tee -a $OUT/log.txt
