#!/bin/bash
# GPU box: parity tests + short benches; prints compact results.
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -4 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1   # no further GPU work after a failed (possibly faulting) test run
for cfg in "u8 linear" "f32 linear" "u8 nearest"; do
  set -- $cfg
  python bench.py --steps 100 --warmup 10 --dtype $1 --interp $2 --no-cpu-baseline --no-variants --no-configs "${@:3}" > gpurun_out/bench_$1_$2.json 2> gpurun_out/bench_$1_$2.err || tail -5 gpurun_out/bench_$1_$2.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_$1_$2.json")); r=d["roofline"]
print("$1 $2: %.0f Mpix/s  kernel %.1f us (min %.1f)  achieved %.0f GB/s  frac %.3f" % (d["value"], r["kernel_ms_mean"]*1e3, r["kernel_ms_min"]*1e3, r["achieved"], r["frac"]))
PY
done
