#!/bin/bash
# Run ON THE GPU BOX: SQ instruction-mix counters of one perf_probe run.  usage: pmc_probe.sh <tag> <perf_probe args...>
set -o pipefail
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/tools/perf_probe.py --reps 1 $*"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1 || exit 6
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1 || exit 7
python3 $REPO/tools/pmc_summary.py $OUT/a/p_counter_collection.csv k_query_packet > $OUT/pmc.txt
python3 $REPO/tools/pmc_summary.py $OUT/b/p_counter_collection.csv k_query_packet >> $OUT/pmc.txt
cat $OUT/pmc.txt
