#!/bin/bash
# PMC passes over tools/bench_kernels.py (each counter group in its own run; kernel-trace only)
TAG=${1:-pmc}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python $GRAFT_REPO_ROOT/tools/bench_kernels.py "$@" > $OUT/plain.log 2>&1; tail -3 $OUT/plain.log
i=0
# PMC_ONLY=traffic: just the two HBM passes (FETCH_SIZE, WRITE_SIZE) that feed profiles/hbm_traffic.json
GRPS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
      "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
      "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum")
[ "${PMC_ONLY:-}" = "traffic" ] && GRPS=("FETCH_SIZE" "WRITE_SIZE")
[ "${PMC_ONLY:-}" = "sq" ] && GRPS=("${GRPS[0]}" "${GRPS[1]}")
[ -n "${GRPS_SEL:-}" ] && GRPS=("${GRPS[@]:0:${GRPS_SEL}}")      # the first n groups (4 = SQ instruction mix + FETCH_SIZE + WRITE_SIZE)
for grp in "${GRPS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --iters 2 "$@" > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    short = "bwd" if ("cgconv_bwd_kernel" in k or "ep::bwd" in k or "bwd2_kernel" in k or "bwd_kernel" in k) else "fwd" if ("cgconv_fwd_kernel" in k or "cb::fwd_kernel" in k) else \
        "node" if "cgconv_node" in k else "rbf" if "rbf_" in k else None
    if short:
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
  fi
  find $OUT/p$i -name "*.csv" -size +5M -delete
done 2>&1 | tee $OUT/summary.txt
# HBM traffic per launch for bench.py's roofline.traffic (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports
# half of the bytes of a wide coalesced read stream -> doubled; WRITE_SIZE taken as is; both are in KiB)
python - $OUT/summary.txt $OUT/plain.log > $OUT/hbm_traffic.json <<'PY'
import ast, glob, hashlib, json, os, re, sys
vals = {}
for line in open(sys.argv[1]):
    m = re.match(r"^(fwd|bwd|rbf|node) (\{.*\})\s*$", line)
    if m:
        vals.setdefault(m.group(1), {}).update(ast.literal_eval(m.group(2)))
ne = re.search(r"N=(\d+) E=(\d+)", open(sys.argv[2]).read())
# fingerprint of the conv kernel sources the counters belong to: bench.py marks the traffic figure stale when it differs
root = os.environ.get("GRAFT_REPO_ROOT", ".")
hsh = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(root, "matdeeplearn_amd/csrc/cgconv*")) + glob.glob(os.path.join(root, "matdeeplearn_amd/csrc/rbf.hip"))):
    hsh.update(open(f, "rb").read())
out = {"N": int(ne.group(1)), "E": int(ne.group(2)), "kernel_sources_sha16": hsh.hexdigest()[:16],
       "note": "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024; separate --pmc passes"}
for k, d in vals.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        out[{"rbf": "mdl_rbf_expand", "node": "mdl_cgconv_bwd_node"}.get(k, "mdl_cgconv_" + k)] = {
            "fetch_kib": d["FETCH_SIZE"], "write_kib": d["WRITE_SIZE"], "bytes": int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)}
print(json.dumps(out, indent=1))
PY
cat $OUT/hbm_traffic.json
