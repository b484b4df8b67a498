#!/bin/bash
# K4 / K4b under the counters (separate passes, kernel-trace only): SQ instruction mix, LDS, HBM bytes (FETCH_SIZE / WRITE_SIZE, KiB) of
# cfconv_fwd_kernel (forward + the dh pass: the same kernel) and cfconv_bwd_w_kernel on the SchNet bench batch -> gpurun_out/<tag>/summary.txt
TAG=${1:-pmck4b}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/tools/bench_cfconv_bwd.py --iters 2 --modes recompute > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    short = "cfconv_fwd" if "cfconv_fwd_kernel" in k else "cfconv_bwd_w" if "cfconv_bwd_w_kernel" in k else "cfconv_bwd_w_reduce" if "bwd_w_reduce" in k else None
    if short:
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", {c: len(v) for c, v in d.items()})
PY
  rm -rf $OUT/p$i
done 2>&1 | tee $OUT/summary.txt
grep -h "N=" $OUT/p1.log | head -1 | tee -a $OUT/summary.txt
