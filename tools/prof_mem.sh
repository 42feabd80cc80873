#!/bin/bash
# Memory-path counters for the warp kernel (GPU box): bash tools/prof_mem.sh <tag> [bench args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profmem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-configs --no-probe $@"
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i ($set) failed: $(grep -m1 -i "error code" $OUT/p$i.log)"
  echo "pass $i done"
done
cd $R
python3 - <<PY
import csv, glob, os
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "warp_" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print("  %-44s %16.1f" % (k, sum(acc[k]) / len(acc[k])))
PY
