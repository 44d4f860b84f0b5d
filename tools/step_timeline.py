#!/usr/bin/env python3
"""Timeline of the last bench step from a rocprofv3 --kernel-trace csv: every kernel with its start offset,
duration and the idle gap before it.  Usage: step_timeline.py t_kernel_trace.csv"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last launch of the dominant kernel marks the last step; walk back to the k_bounds_init before it
idx = max(i for i, r in enumerate(rows) if "k_query_packet" in r["Kernel_Name"])
j = idx
while j > 0 and "k_bounds_init" not in rows[j]["Kernel_Name"]:
    j -= 1
k = idx
while (k + 1 < len(rows) and "k_bounds_init" not in rows[k + 1]["Kernel_Name"] and k - idx < 12
       and "k_face_precompute" not in rows[k + 1]["Kernel_Name"]):
    k += 1
t0 = int(rows[j]["Start_Timestamp"]); prev = t0; busy = 0
for r in rows[j:k + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:48]
    print(f"{(s - t0) / 1e6:9.3f} ms  dur {(e - s) / 1e6:8.3f}  gap {(s - prev) / 1e6:7.3f}  {name}")
    prev = e; busy += e - s
print(f"span {(prev - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms")
