#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv per (kernel, counter). Usage: pmc_summary.py file.csv [substr]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(float); cnt = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    if sub and sub not in k: continue
    acc[(k[:60], r["Counter_Name"])] += float(r["Counter_Value"]); cnt[k[:60]].add(r["Dispatch_Id"])
for (k, c), v in sorted(acc.items()):
    print(f"{k:60s} {c:28s} {v:16.0f}  per-dispatch {v/len(cnt[k]):16.0f}")
