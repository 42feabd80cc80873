#!/bin/bash
# The ablation / ownership / footprint tables of DESIGN.md section 6 (GPU box; builds from `python tools/ablate.py ...`):
#   python tools/ablate.py nostore=nostore noload=noload noblend=noblend notie=notie nomem=nostore+noload ownrow=ownrow ownblk=ownblk \
#       ldsmall=ldsmall stsmall=stsmall noedge=noedge fillall=fillall edgefill=edgefill infill=infill
#   bash tools/r02_tables.sh > gpurun_out/r02_tables.txt
V=bev_amd/csrc/variants
L=bev_amd/csrc/libbevwarp.so
echo "== ablations, u8 bilinear keystone (us per 32 frames; all libraries interleaved in one process)"
python tools/abx.py --rounds 40 --libs cur=$L nostore=$V/nostore.so noload=$V/noload.so nomem=$V/nomem.so noblend=$V/noblend.so notie=$V/notie.so ldsmall=$V/ldsmall.so stsmall=$V/stsmall.so fillall=$V/fillall.so 2>/dev/null
echo "== ablations, u8 bilinear brno"
python tools/abx.py --rounds 40 --homography brno --libs cur=$L noedge=$V/noedge.so nomem=$V/nomem.so fillall=$V/fillall.so edgefill=$V/edgefill.so infill=$V/infill.so 2>/dev/null
echo "== ablations, f32 bilinear brno"
python tools/abx.py --rounds 40 --dtype f32 --homography brno --libs cur=$L fillall=$V/fillall.so edgefill=$V/edgefill.so infill=$V/infill.so 2>/dev/null
for h in brno keystone; do
  echo "== lane layout, u8 bilinear $h (1080p -> 1024^2)"
  python tools/abx.py --rounds 30 --homography $h --libs rule=$L rows=$V/ownrow.so patches=$V/ownblk.so 2>/dev/null
done
for h in rot0z1.4 rot4z1.4 rot6z1.4 rot10z1.4 rot15z1.4 rot25z1.4 rot45z1.4 rot90z1.4 rot8z1.0 rot12z1.0 rot16z1.0; do
  echo "== lane layout, u8 bilinear $h, all-interior footprint (32 x 3840x2160 -> 1024^2)"
  python tools/abx.py --rounds 30 --src 3840 2160 --homography $h --libs rule=$L rows=$V/ownrow.so patches=$V/ownblk.so 2>/dev/null
done
echo "== minification and rotation, all-interior footprints (u8 bilinear, 32 x 3840x2160 -> 1024^2)"
for h in rot0z0.5 rot0z1.0 rot0z1.5 rot0z2.0 rot20z1.0 rot20z1.5 rot20z2.0; do
  echo "-- $h"
  python tools/abx.py --rounds 30 --src 3840 2160 --homography $h --libs rule=$L rows=$V/ownrow.so patches=$V/ownblk.so 2>/dev/null
done
echo "== batch size, u8 bilinear keystone"
for b in 8 16 32 64 128; do echo "-- batch $b"; python tools/abx.py --libs cur=$L --batch $b --rounds 40 2>/dev/null | tail -1; done
