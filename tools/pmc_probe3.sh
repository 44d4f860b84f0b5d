#!/bin/bash
# Run ON THE GPU BOX: scalar-data-cache (SQC) and instruction-level counters of one perf_probe run.
# usage: pmc_probe3.sh <tag> <perf_probe args...>
set -o pipefail
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/tools/perf_probe.py --reps 1 $*"
timeout -s KILL 200 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_BUSY_CYCLES SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_ICACHE_MISSES --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1 || exit 6
timeout -s KILL 200 rocprofv3 --pmc SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1 || exit 7
python3 $REPO/tools/pmc_summary.py $OUT/a/p_counter_collection.csv k_query_packet > $OUT/pmc.txt
python3 $REPO/tools/pmc_summary.py $OUT/b/p_counter_collection.csv k_query_packet >> $OUT/pmc.txt
cat $OUT/pmc.txt
