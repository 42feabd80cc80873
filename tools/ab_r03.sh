#!/bin/bash
# A/B of library builds on one box, interleaved (tools/abx.py), over the benchmarked formats and footprints.
#   bash tools/ab_r03.sh "base=bev_amd/csrc/variants/base.so new=bev_amd/csrc/libbevwarp.so" [rounds] [specs...]
LIBS=$1; R=${2:-30}; shift; shift
SPECS=("$@")
[ ${#SPECS[@]} -eq 0 ] && SPECS=("u8 linear keystone" "u8 linear brno" "u8 nearest keystone" "f32 linear keystone" "f32 linear brno" "u8 linear rot25")
for spec in "${SPECS[@]}"; do
  set -- $spec
  echo "== $1 $2 $3"
  python3 tools/abx.py --libs $LIBS --dtype $1 --interp $2 --homography $3 --rounds $R --check 2>&1 | grep -v amdgpu.ids || exit 1
done
