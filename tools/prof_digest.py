#!/usr/bin/env python3
"""Digest gpurun_out/<tag>/ (made by tools/collect_profiles.sh) into profiles/<tag>_*:
   <tag>_bench.json, <tag>_kernel_stats.txt, <tag>_pmc.txt, <tag>_pmc.json (per-launch HBM
   traffic of the dominant kernel, with the MI355X_MICROARCH.md corrections spelled out)."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
DOM = "k_query_packet<2, true, true>"  # <Q_SDF, robust, approximate first pass>


def short(n):
    n = n.replace("void ", "")
    if "at::native" in n:
        return "at::native::<torch kernel>"
    return n.split("(")[0]


line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]
open(os.path.join(dst, f"{tag}_bench.json"), "w").write(line)
rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_kernel_stats.txt"), "w") as fh:
    fh.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-grid\n")
    fh.write(f"{'kernel':48s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>10s} {'pct':>7s}\n")
    for r in rows[:24]:
        fh.write(f"{short(r['Name'])[:48]:48s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.3f} "
                 f"{float(r['AverageNs'])/1e6:10.4f} {float(r['Percentage']):7.2f}\n")
acc = collections.defaultdict(float)
disp = collections.defaultdict(set)
for f in ("pmc_pf.csv", "pmc_pw.csv", "pmc_pt.csv", "pmc_ps1.csv", "pmc_ps2.csv", "pmc_psc.csv"):
    p = os.path.join(src, f)
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        k = short(r["Kernel_Name"])
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
with open(os.path.join(dst, f"{tag}_pmc.txt"), "w") as fh:
    fh.write("# rocprofv3 --pmc <one group per run> -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-grid\n")
    fh.write("# per-dispatch averages; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (uncorrected)\n")
    for (k, c), v in sorted(acc.items()):
        if k.startswith("at::"):
            continue
        fh.write(f"{k[:44]:44s} {c:24s} {v/len(disp[(k,c)]):18.0f}\n")


def per(counter):
    k = [kk for kk in acc if kk[1] == counter and kk[0].startswith("k_query_packet<2, true")]
    return acc[k[0]] / len(disp[k[0]]) if k else None


fetch_kb, write_kb = per("FETCH_SIZE"), per("WRITE_SIZE")
rd, wr = per("TCC_EA0_RDREQ_sum"), per("TCC_EA0_WRREQ_sum")
out = {"kernel": DOM, "fetch_size_kb_reported": fetch_kb, "write_size_kb_reported": write_kb,
       "tcc_ea0_rdreq": rd, "tcc_ea0_wrreq": wr, "tcc_hit": per("TCC_HIT_sum"), "tcc_miss": per("TCC_MISS_sum"),
       # MI355X_MICROARCH.md §HBM: FETCH_SIZE = RDREQ x 64 B and reads HALF the bytes of wide
       # (128 B request) coalesced streams; this kernel's reads are 12-B gathers and scalar/L2
       # misses whose request width is uncalibrated, so both bounds are recorded.
       "read_bytes_low": None if fetch_kb is None else fetch_kb * 1024.0,
       "read_bytes_high": None if fetch_kb is None else 2.0 * fetch_kb * 1024.0,
       "write_bytes": None if write_kb is None else write_kb * 1024.0}
# VALU utilisation against chip peak (north_star): a wave64 VALU instruction occupies its SIMD-32
# for 2 cycles (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs.
iv, gui, thr, act = per("SQ_INSTS_VALU"), per("GRBM_GUI_ACTIVE"), per("SQ_THREAD_CYCLES_VALU"), per("SQ_ACTIVE_INST_VALU")
if iv and gui:
    out["valu_insts_per_wave"] = iv / per("SQ_WAVES") if per("SQ_WAVES") else None
    out["valu_issue_util"] = 2.0 * iv / ((gui / 8.0) * 1024.0)
if thr and act:
    out["valu_lane_util"] = thr / (act * 64.0)
# scalar data cache (the chunk loads of the walk), per wave (round 1, 4-wide tree: 312 requests per wave)
rq, hit, mis = per("SQC_DCACHE_REQ"), per("SQC_DCACHE_HITS"), per("SQC_DCACHE_MISSES")
if rq and per("SQ_WAVES"):
    # (SQ_WAVES comes from another pass of the same command: the launch is identical)
    out["sqc_dcache_req_per_wave"] = rq / per("SQ_WAVES")
    out["sqc_dcache_misses_per_wave"] = None if mis is None else mis / per("SQ_WAVES")
    out["sqc_dcache_hit_rate"] = None if hit is None else hit / rq
    out["sqc_tc_data_read_req"] = per("SQC_TC_DATA_READ_REQ")
if out["read_bytes_low"] is not None and out["write_bytes"] is not None:
    out["traffic_bytes_per_launch"] = out["read_bytes_low"] + out["write_bytes"]
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps(out))
