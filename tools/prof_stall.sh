#!/bin/bash
# Where the memory path stalls (GPU box): bash tools/prof_stall.sh <tag> [bench args]
# Separate --pmc passes (never with trace domains), two to four counters per block and pass: SQ issue-side FIFOs and
# instruction fetch, TA / TCP stall reasons, L2 (TCC) stalls, and the L2 <-> fabric (EA) request levels, from which the
# mean HBM read / write latency follows (LEVEL / REQ).  Output: gpurun_out/profstall_<tag>/summary.txt .
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profstall_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-configs --no-probe $@"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_LEVEL_WAVES" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_TC_STALL SQC_TC_INST_REQ SQC_DCACHE_REQ SQC_DCACHE_MISSES" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
           "TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum TCC_IB_REQ_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i ($set) failed: $(grep -m1 -i "error" $OUT/p$i.log | cut -c1-160)"
done
cd $R
python3 - <<PY > $OUT/summary.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "warp_rows" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== stall counters, mean per dispatch of warp_rows ($TAG: $@)")
m = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(m):
    print("  %-44s %18.1f  (n=%d)" % (k, m[k], len(acc[k])))
def ratio(name, a, b, scale=1.0):
    if a in m and b in m and m[b]:
        print("  %-60s %12.3f" % (name, scale * m[a] / m[b]))
print("== derived")
ratio("HBM/fabric read latency, cycles (EA RDREQ_LEVEL / RDREQ)", "TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum")
ratio("HBM/fabric write latency, cycles (EA WRREQ_LEVEL / WRREQ)", "TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_WRREQ_sum")
ratio("L1 -> L2 read latency, cycles", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum")
ratio("L1 -> L2 write latency, cycles", "TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_sum")
ratio("L2 busy share (TCC_BUSY / TCC_CYCLE)", "TCC_BUSY_sum", "TCC_CYCLE_sum")
ratio("L2 tag stall share (TCC_TAG_STALL / TCC_CYCLE)", "TCC_TAG_STALL_sum", "TCC_CYCLE_sum")
ratio("EA write-request stall share of L2 cycles", "TCC_EA0_WRREQ_STALL_sum", "TCC_CYCLE_sum")
ratio("too-many-EA-writes stall share of L2 cycles", "TCC_TOO_MANY_EA_WRREQS_STALL_sum", "TCC_CYCLE_sum")
ratio("DRAM read credit stall share of L2 cycles", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_CYCLE_sum")
ratio("DRAM write credit stall share of L2 cycles", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_CYCLE_sum")
ratio("I-cache hit rate", "SQC_ICACHE_HITS", "SQC_ICACHE_REQ")
ratio("VMEM instructions in flight per wave-slot sample (INST_LEVEL_VMEM / WAVE_CYCLES)", "SQ_INST_LEVEL_VMEM", "SQ_WAVE_CYCLES")
ratio("instruction fetches in flight (IFETCH_LEVEL / WAVE_CYCLES)", "SQ_IFETCH_LEVEL", "SQ_WAVE_CYCLES")
ratio("TCP pending-stall share of TCP busy", "TCP_PENDING_STALL_CYCLES_sum", "TCP_GATE_EN1_sum")
ratio("TCP <- TCR stall share of TCP busy", "TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_GATE_EN1_sum")
ratio("TA address stalled by TC, share of TA busy", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_TA_BUSY_sum")
ratio("TA data stalled by TC, share of TA busy", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_TA_BUSY_sum")
PY
cat $OUT/summary.txt
