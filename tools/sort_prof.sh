#!/bin/bash
# kernel durations of the query sort (and the rest) for one large device call; run on the GPU box
N=${1:-125000000}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sort_prof -o b -- python3 $GRAFT_REPO_ROOT/tools/perf_probe.py --n $N --only sdf --reps 2 > $GRAFT_REPO_ROOT/gpurun_out/sort_prof.log 2>&1 || exit 1
grep "^sdf" $GRAFT_REPO_ROOT/gpurun_out/sort_prof.log
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/sort_prof/b_kernel_stats.csv")))
for r in rows[:8]:
    print("%-60s %4s calls  avg %9.1f us  max %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
