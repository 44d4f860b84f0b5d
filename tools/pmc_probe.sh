#!/bin/bash
# Run ON THE GPU BOX: PMC counters of the packet kernel in one perf_probe run, one rocprofv3 --pmc pass per
# counter set (the hardware holds 8 SQ / 4 TCC counters at a time).
# usage: pmc_probe.sh <tag> <set> <perf_probe args...>     set: sq (instruction mix, waits) | sqc (scalar data
#        cache + SMEM/VMEM levels) | tcc (L2 hits / misses / EA requests)
set -o pipefail
TAG=$1; SET=$2; shift 2
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/tools/perf_probe.py --reps 1 $*"
pass() {  # <name> <counters...>
  local name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- $CMD > $OUT/$name.log 2>&1 || exit 6
  python3 $REPO/tools/pmc_summary.py $OUT/$name/p_counter_collection.csv k_query_packet >> $OUT/pmc.txt
}
: > $OUT/pmc.txt
case $SET in
  sq)  pass a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
       pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS ;;
  sqc) pass a SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_BUSY_CYCLES SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_ICACHE_MISSES
       pass b SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES ;;
  tcc) pass a TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum ;;
  *) echo "unknown set $SET"; exit 2 ;;
esac
cat $OUT/pmc.txt
