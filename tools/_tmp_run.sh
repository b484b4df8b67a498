for m in gcn mpnn; do timeout 900 python bench.py --model $m --no-cpu-baseline --sustain-s 1.0 2>&1 | tail -1 > gpurun_out/bench_$m.json; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_$m.json"))
    print("$m", d["ms_per_step"], d["value"], d["roofline"], "sustained", {k: d["sustained"].get(k) for k in ("mode","ms_per_step","graph_error")}, "eager", d["sustained"]["eager"]["ms_per_step"])
    print("   ref100", {k: d["ref_batch_100"].get(k) for k in ("mode","ms_per_step","value","graph_error")}, "eager", d["ref_batch_100"]["eager"], "fp32", d.get("fp32_mode"))
except Exception as e:
    print("$m failed", e, open("gpurun_out/bench_$m.json").read()[-600:])
PY
done
