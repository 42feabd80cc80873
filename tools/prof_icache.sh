#!/bin/bash
# Instruction-fetch counters of the warp kernel (GPU box):  bash tools/prof_icache.sh <tag> [bench args...]   -> gpurun_out/icache_<tag>.txt
# Two counter passes (never with trace domains): SQC instruction-cache requests / hits / misses, SQ_IFETCH and the derived fetch latency.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/icache_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-configs --no-probe $@"
timeout -k 5 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $R/bench.py $ARGS > $OUT/p1.log 2>&1 || echo "p1 failed"
timeout -k 5 300 rocprofv3 --pmc InstrFetchLatency --output-format csv -d $OUT/p2 -- python3 $R/bench.py $ARGS > $OUT/p2.log 2>&1 || echo "p2 failed"
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d $OUT/p3 -- python3 $R/bench.py $ARGS > $OUT/p3.log 2>&1 || echo "p3 failed"
cd $R
python3 - $OUT <<'PY' > $R/gpurun_out/icache_$TAG.txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for sub in ("p1", "p2", "p3"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "warp_rows" in row["Kernel_Name"]:
                acc[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        print("%s  %s" % (sub, k))
        for c, v in sorted(d.items()):
            print("     %-30s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat $R/gpurun_out/icache_$TAG.txt
