#!/bin/bash
# gpu_k3rep.sh <tag> <reps> <variants...>: K3 timings of the built library ("base") and prebuilt variants, interleaved <reps> times in
# ONE box session (60 launches each; the run-to-run spread of a 20-launch median is +-10 us, more than most steps being measured)
set -u
TAG=$1; REPS=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/rep.log
for r in $(seq 1 $REPS); do
  for v in base "$@"; do
    if [ "$v" = base ]; then unset MDL_HIP_LIB; else export MDL_HIP_LIB=$GRAFT_REPO_ROOT/matdeeplearn_amd/lib/variants/$v.so; fi
    echo -n "$v: " >> $OUT/rep.log
    timeout 300 python tools/bench_kernels.py --which fwd,bwd --iters 60 2>&1 | grep -E "^bwd:|rror" >> $OUT/rep.log
  done
done
unset MDL_HIP_LIB
python - $OUT/rep.log <<'PY'
import sys, re, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.match(r"(\S+): bwd: avg ([\d.]+) us  median ([\d.]+) us  min ([\d.]+)", l)
    if m: d[m.group(1)].append((float(m.group(3)), float(m.group(4)), float(m.group(2))))
for k, v in d.items():
    print("%-10s medians %s  mins %s" % (k, [x[0] for x in v], [x[1] for x in v]))
PY
