#!/bin/bash
# usage: bash tools/ab.sh "<label>|<env assignments>|<bench args>" ...   (GPU box)
for spec in "$@"; do
  IFS='|' read -r label envs args <<< "$spec"
  out=gpurun_out/ab_$label.json
  env $envs python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-variants --no-configs $args > $out 2> gpurun_out/ab_$label.err || { echo "$label FAILED"; tail -3 gpurun_out/ab_$label.err; continue; }
  python - <<PY
import json
d=json.load(open("$out")); r=d["roofline"]
print("%-28s kernel %.1f us (min %.1f)  %.0f Mpix/s  frac %.3f" % ("$label", r["kernel_ms_mean"]*1e3, r["kernel_ms_min"]*1e3, d["value"], r["frac"]))
PY
done
