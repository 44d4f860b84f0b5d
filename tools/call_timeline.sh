#!/bin/bash
# kernel timeline of ONE device call (last repetition) of perf_probe; run on the GPU box: bash tools/call_timeline.sh --n 4000000 --only nn
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/call_tl
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/call_tl -o t -- python3 $GRAFT_REPO_ROOT/tools/perf_probe.py --reps 2 "$@" > $GRAFT_REPO_ROOT/gpurun_out/call_tl.log 2>&1 || exit 1
grep -E "^(nn|sdf|contains)" $GRAFT_REPO_ROOT/gpurun_out/call_tl.log
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/call_tl/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: from the last k_bounds_init (or k_query_morton) on
start = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(("k_bounds_init", "k_query_morton")))
if rows[start]["Kernel_Name"].startswith("k_query_morton") and start and rows[start - 1]["Kernel_Name"].startswith("k_bounds"):
    start -= 2
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.3f ms  dur %7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, r["Kernel_Name"][:70]))
PY
