#!/bin/bash
# Run ON THE GPU BOX: L2 (TCC) counters of one perf_probe run.  usage: pmc_probe2.sh <tag> <perf_probe args...>
set -o pipefail
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/tools/perf_probe.py --reps 1 $*"
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1 || exit 6
timeout -s KILL 200 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1 || echo "sqc pass failed"
python3 $REPO/tools/pmc_summary.py $OUT/a/p_counter_collection.csv k_query_packet > $OUT/pmc.txt
[ -f $OUT/b/p_counter_collection.csv ] && python3 $REPO/tools/pmc_summary.py $OUT/b/p_counter_collection.csv k_query_packet >> $OUT/pmc.txt
cat $OUT/pmc.txt
tail -3 $OUT/b.log
