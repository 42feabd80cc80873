#!/bin/bash
# A/B of two builds over the main variants (GPU box): bash tools/ab_quick.sh base.so new.so
A=$1; B=$2
for cfg in "u8 linear keystone" "u8 linear brno" "u8 nearest keystone" "f32 linear keystone" "f32 linear brno" "u8 linear rot25z1.4"; do
  set -- $cfg
  echo "== $cfg"
  python tools/abx.py --rounds 40 --check --dtype $1 --interp $2 --homography $3 --libs base=$A new=$B 2>/dev/null
done
