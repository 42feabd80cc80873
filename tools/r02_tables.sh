#!/bin/bash
# The ablation and ownership tables of DESIGN.md section 6 (GPU box; builds from `python tools/ablate.py ...` must exist):
#   bash tools/r02_tables.sh > gpurun_out/r02_tables.txt
V=bev_amd/csrc/variants
echo "== ablations, u8 bilinear keystone (us per 32 frames; interleaved in one process)"
python tools/abx.py --rounds 40 --libs cur=bev_amd/csrc/libbevwarp.so nostore=$V/nostore.so noload=$V/noload.so nomem=$V/nomem.so noblend=$V/noblend.so notie=$V/notie.so ldsmall=$V/ldsmall.so stsmall=$V/stsmall.so 2>/dev/null
echo "== ablations, u8 bilinear brno"
python tools/abx.py --rounds 40 --homography brno --libs cur=bev_amd/csrc/libbevwarp.so noedge=$V/noedge.so nomem=$V/nomem.so 2>/dev/null
for h in rot0z1.4 rot5z1.4 rot10z1.4 rot15z1.4 rot25z1.4 rot45z1.4 brno keystone; do
  echo "== ownership, u8 bilinear $h"
  python tools/abx.py --rounds 30 --homography $h --libs rule=bev_amd/csrc/libbevwarp.so rows=$V/ownrow.so blocks=$V/ownblk.so 2>/dev/null
done
