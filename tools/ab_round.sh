#!/bin/bash
# GPU box: parity tests, then interleaved A/B of the baseline build against the current one.
#   bash tools/ab_round.sh <base.so> [extra label=path ...]
BASE=$1; shift
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -4 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit 1   # no further GPU work after a failed (possibly faulting) test run
for cfg in "u8 linear keystone" "u8 linear brno" "f32 linear keystone" "f32 linear brno" "u8 nearest keystone"; do
  set -- $cfg
  echo "== $cfg"
  timeout -k 10 300 python tools/abx.py --libs base=$BASE new=bev_amd/csrc/libbevwarp.so "${EXTRA[@]}" --dtype $1 --interp $2 --homography $3 --rounds 30 --check 2>&1 | tail -6 || exit 1
done
