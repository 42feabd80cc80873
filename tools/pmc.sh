#!/bin/bash
# One rocprofv3 --pmc pass per argument (a quoted list of counters) over `python bench.py --no-variants ...` (GPU box).
#   BEVWARP_LIB=<.so> BENCH_ARGS="--dtype u8" bash tools/pmc.sh <tag> "SQ_WAVES SQ_BUSY_CYCLES" "SQ_INST_LEVEL_VMEM ..." ...
# Counter passes never carry trace domains (--kernel-trace / --stats / --sys-trace), as the pool requires.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
[ -n "$BEVWARP_LIB" ] && [ "${BEVWARP_LIB:0:1}" != "/" ] && export BEVWARP_LIB=$R/$BEVWARP_LIB
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-configs --no-probe $BENCH_ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
cd $R
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "warp_" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print("  %-34s %16.1f" % (k, sum(acc[k]) / len(acc[k])))
PY
