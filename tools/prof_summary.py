#!/usr/bin/env python3
"""Condense a rocprofv3 rocpd database (…_results.db) into the per-kernel stats table that
`--stats` prints: name, calls, total/avg duration.  Usage: prof_summary.py results.db [out.txt]"""
import re
import sqlite3
import sys


def short(name, n=110):
    name = re.sub(r"at::native::.*", "at::native::<torch kernel>", name)
    return name if len(name) <= n else name[: n - 3] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    lines = [f"{'kernel':110s} {'calls':>7s} {'total_us':>14s} {'avg_us':>12s} {'pct':>7s}"]
    for name, calls, total, avg, pct in rows:
        lines.append(f"{short(name):110s} {calls:7d} {total / 1e0 / 1e3:14.1f} {avg / 1e3:12.2f} {pct:7.2f}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
