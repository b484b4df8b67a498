"""Aggregate a gpurun_out/<tag>/small_trace.txt (tools/gpu_small_trace.sh) by kernel: launches, mean and total duration."""
import re, collections, sys
rows = open(sys.argv[1]).read().splitlines()
print(rows[0])
agg = collections.defaultdict(lambda: [0, 0.0])
for l in rows[1:]:
    m = re.match(r"\s*([\d.]+)\s+\+gap\s+(-?[\d.]+)\s+dur\s+([\d.]+)\s+q\S+\s+(?:g\S+\s+)?(.*)", l)
    if m:
        a = agg[m.group(4)[:78]]; a[0] += 1; a[1] += float(m.group(3))
print("launches", sum(v[0] for v in agg.values()), "sum of durations %.0f us" % sum(v[1] for v in agg.values()))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print("%5d x %7.1f us = %8.1f  %s" % (v[0], v[1] / v[0], v[1], k))
