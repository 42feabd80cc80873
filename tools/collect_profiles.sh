#!/bin/bash
# Build side: copy the newest profile set from gpurun_out/ into profiles/ and rebuild profiles/pmc_traffic.json.
#   bash tools/collect_profiles.sh [geom-tag]
for t in u8_linear f32_linear u8_nearest u8_brno f32_brno u8_nearest_brno; do
  cp gpurun_out/prof_$t/summary.txt profiles/r03_${t}_rocprofv3_summary.txt
  cp "$(ls -t gpurun_out/prof_$t/kt/runc/*_kernel_stats.csv | head -1)" profiles/r03_${t}_kernel_stats.csv
done
[ -n "$1" ] && cp gpurun_out/profgeom_$1/summary.txt profiles/r03_geom_rocprofv3_summary.txt
python3 tools/make_traffic.py r03 | head -2
