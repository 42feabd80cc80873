#!/bin/bash
# SQ counters only (GPU box): bash tools/prof_sq.sh <tag> [bench args]   (BEVWARP_LIB etc. are inherited)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profsq_$TAG
mkdir -p $OUT
[ -n "$BEVWARP_LIB" ] && [ "${BEVWARP_LIB:0:1}" != "/" ] && export BEVWARP_LIB=$R/$BEVWARP_LIB
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-configs --no-probe $@"
timeout -k 5 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || echo "pmc_sq failed"
echo "pass 1 done"
timeout -k 5 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1 || echo "pmc_sq2 failed"
echo "pass 2 done"
timeout -k 5 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH --output-format csv -d $OUT/pmc_sq3 -- python3 $R/bench.py $ARGS > $OUT/pmc_sq3.log 2>&1 || echo "pmc_sq3 failed"
echo "pass 3 done"
cd $R
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("$OUT/pmc_sq*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "warp_" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print("  %-28s %16.1f" % (k, sum(acc[k]) / len(acc[k])))
PY
