#!/bin/bash
# Build side: copy the newest profile set from gpurun_out/ into profiles/ and rebuild profiles/pmc_traffic.json.
#   bash tools/collect_profiles.sh <round, e.g. r04> [geom-tag]
R=${1:?round tag}
for t in u8_linear f32_linear u8_nearest u8_brno f32_brno u8_nearest_brno; do
  [ -f gpurun_out/prof_$t/summary.txt ] || continue
  cp gpurun_out/prof_$t/summary.txt profiles/${R}_${t}_rocprofv3_summary.txt
  cp "$(ls -t gpurun_out/prof_$t/kt/runc/*_kernel_stats.csv | head -1)" profiles/${R}_${t}_kernel_stats.csv
done
[ -n "$2" ] && cp gpurun_out/profgeom_$2/summary.txt profiles/${R}_geom_rocprofv3_summary.txt
python3 tools/make_traffic.py $R | head -2
