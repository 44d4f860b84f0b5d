#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root.  Produces, under gpurun_out/<tag>/:
#   bench.json                 the default bench.py line
#   kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command
#   pmc_pf / pw / pt / ps1 / ps2 / psc .csv   separate --pmc passes (fetch, write, TCC, SQ x 2, scalar data cache)
# Copy the summaries you want judged into profiles/ (tools/prof_digest.py does that).
set -o pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-grid --no-c5"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1 || exit 2
cp $OUT/trace/t_kernel_stats.csv $OUT/kernel_stats.csv
P1="python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-grid --no-c5"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pf -o p -- $P1 > $OUT/pf.log 2>&1 || exit 3
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pw -o p -- $P1 > $OUT/pw.log 2>&1 || exit 4
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pt -o p -- $P1 > $OUT/pt.log 2>&1 || exit 5
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/ps1 -o p -- $P1 > $OUT/ps1.log 2>&1 || exit 6
timeout -s KILL 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_BUSY_CYCLES --output-format csv -d $OUT/ps2 -o p -- $P1 > $OUT/ps2.log 2>&1 || exit 7
timeout -s KILL 200 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ --output-format csv -d $OUT/psc -o p -- $P1 > $OUT/psc.log 2>&1 || exit 8
for d in pf pw pt ps1 ps2 psc; do cp $OUT/$d/p_counter_collection.csv $OUT/pmc_$d.csv; done
echo collected $OUT
