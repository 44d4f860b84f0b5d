#!/bin/bash
# kernel timeline of the last pipelined C5 frame (update + contains 4M + nn 4M); run on the GPU box
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/c5_tl
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5_tl -o t -- python3 $GRAFT_REPO_ROOT/tools/c5_probe.py --pipeline 1 --frames 6 > $GRAFT_REPO_ROOT/gpurun_out/c5_tl.log 2>&1 || exit 1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/c5_tl.log
python3 - <<'PY'
import csv, os
d = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/c5_tl/"
ev = []
for r in csv.DictReader(open(d + "t_kernel_trace.csv")):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
try:
    for r in csv.DictReader(open(d + "t_memory_copy_trace.csv")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", "")))
except Exception as e:
    print("no copy trace", e)
ev.sort()
# last frame: from the last k_face_precompute on
start = max(i for i, e in enumerate(ev) if e[2].startswith("k_face_precompute"))
prev = max(i for i, e in enumerate(ev[:start]) if e[2].startswith("k_face_precompute"))
t0 = ev[prev][0]
agg = {}
for s, e, n in ev[prev:start]:
    agg[n] = agg.get(n, 0) + (e - s)
print("one frame (between two k_face_precompute): %.3f ms" % ((ev[start][0] - t0) / 1e6))
for n, t in sorted(agg.items(), key=lambda x: -x[1])[:22]:
    print("  %8.3f ms  %s" % (t / 1e6, n))
busy = sum(e - s for s, e, n in ev[prev:start])
print("  sum of durations %.3f ms" % (busy / 1e6))
PY
