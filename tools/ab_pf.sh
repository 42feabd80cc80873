#!/bin/bash
# A/B of the tile prefetch (GPU box): product vs variants/pf4.so, pf6.so (tools/ablate.py pf4=pf4 pf6=pf6)
V=bev_amd/csrc/variants
for cfg in "u8 linear keystone" "u8 nearest keystone" "f32 linear keystone" "u8 linear brno" "f32 linear brno"; do
  set -- $cfg
  echo "== $cfg"
  timeout -k 10 150 python tools/abx.py --rounds 40 --check --dtype $1 --interp $2 --homography $3 --libs base=bev_amd/csrc/libbevwarp.so pf4=$V/pf4.so pf6=$V/pf6.so 2>/dev/null || exit 1
done
