#!/bin/bash
# A/B of workgroup sizes (GPU box): product (4 waves) vs variants/wg1.so, variants/wg2.so (tools/ablate.py wg1=wg1 wg2=wg2)
V=bev_amd/csrc/variants
for cfg in "u8 linear keystone" "u8 nearest keystone" "f32 linear keystone" "u8 linear brno" "f32 linear brno"; do
  set -- $cfg
  echo "== $cfg"
  timeout -k 10 150 python tools/abx.py --rounds 40 --check --dtype $1 --interp $2 --homography $3 --libs base=bev_amd/csrc/libbevwarp.so wg1=$V/wg1.so wg2=$V/wg2.so 2>/dev/null || exit 1
done
