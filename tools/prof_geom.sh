#!/bin/bash
# Kernel-trace stats + one SQ pass + HBM byte passes for the kernels beside the warp (GPU box):  bash tools/prof_geom.sh <tag>
# Counter passes never carry trace domains, as the pool requires.  Output under gpurun_out/profgeom_<tag>/ .
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profgeom_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tools/prof_geom.py 30 > $OUT/kt.log 2>&1 || echo "kt failed"
timeout -k 5 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/prof_geom.py 10 > $OUT/pmc_sq.log 2>&1 || echo "pmc_sq failed"
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/prof_geom.py 10 > $OUT/pmc_fetch.log 2>&1 || echo "pmc_fetch failed"
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/prof_geom.py 10 > $OUT/pmc_write.log 2>&1 || echo "pmc_write failed"
cd $R
python3 - <<PY > $OUT/summary.txt
import csv, glob, os
from collections import defaultdict
out = "$OUT"
KEEP = ("project_points", "rbox_iou", "tracker_step", "warp_composite", "composite_kernel", "warp_rows", "rbox_transform")
for f in sorted(glob.glob(os.path.join(out, "kt", "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        if any(k in row.get("Name", "") for k in KEEP):
            print("  %-86s calls %5s  avg %10s ns  min %9s  max %9s" % (row["Name"][:86], row["Calls"], row["AverageNs"], row["MinNs"], row["MaxNs"]))
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:86]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("== counters (%s): mean per dispatch" % sub)
    for k, d in acc.items():
        if not any(x in k for x in KEEP):
            continue
        print("  " + k)
        for c, v in sorted(d.items()):
            print("     %-24s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat $OUT/summary.txt
