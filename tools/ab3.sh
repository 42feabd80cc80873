#!/bin/bash
# three-way A/B over the main variants (GPU box): bash tools/ab3.sh a.so b.so c.so
for cfg in "u8 linear keystone" "u8 linear brno" "u8 nearest keystone" "f32 linear keystone" "u8 linear rot25z1.4"; do
  set -- $cfg
  echo "== $1 $2 $3"
  python tools/abx.py --rounds 40 --check --dtype $1 --interp $2 --homography $3 --libs base=bev_amd/csrc/variants/base.so pair=bev_amd/csrc/variants/pair.so new=bev_amd/csrc/libbevwarp.so 2>/dev/null
done
