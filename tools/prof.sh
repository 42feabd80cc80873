#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/prof.sh <tag> [bench args...]
# Writes rocprofv3 kernel-trace stats and PMC passes under gpurun_out/prof_<tag>/ .
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-configs --no-probe $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py $ARGS > $OUT/kt.log 2>&1 || echo "kt failed"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || echo "pmc_sq failed"
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1 || echo "pmc_sq2 failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || echo "pmc_fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1 || echo "pmc_write failed"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_tcc -- python3 $R/bench.py $ARGS > $OUT/pmc_tcc.log 2>&1 || echo "pmc_tcc failed"
cd $R
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
