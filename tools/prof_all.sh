#!/bin/bash
# The round's profile set (GPU box): kernel-trace stats + SQ / FETCH / WRITE / TCC passes for the benchmarked workloads of configs[1];
# profiles/pmc_traffic.json is then rebuilt from them by tools/collect_profiles.sh <round> (build side).
#   bash tools/prof_all.sh [first|second]      (two halves, so that each fits one gpurun call)
FIRST=("u8_linear --dtype u8" "f32_linear --dtype f32" "u8_nearest --dtype u8 --interp nearest")
SECOND=("u8_brno --dtype u8 --homography brno" "f32_brno --dtype f32 --homography brno" "u8_nearest_brno --dtype u8 --interp nearest --homography brno")
case "$1" in first) SPECS=("${FIRST[@]}");; second) SPECS=("${SECOND[@]}");; *) SPECS=("${FIRST[@]}" "${SECOND[@]}");; esac
for spec in "${SPECS[@]}"; do
  set -- $spec
  tag=$1; shift
  bash tools/prof.sh $tag "$@" > gpurun_out/prof_$tag.log 2>&1
  echo "== $tag"; grep -E "warp_rows|FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU |SQ_ACTIVE_INST_VALU" gpurun_out/prof_$tag/summary.txt | cut -c1-140
done
