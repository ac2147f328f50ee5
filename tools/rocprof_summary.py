#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) as text for profiles/.

usage: tools/rocprof_summary.py <results.db> [title]  > profiles/<name>.txt
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
print("# rocprofv3 --kernel-trace --stats summary: %s" % title)
print("%-72s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-72s %8d %14.1f %12.3f %7.2f" % (name[:72], calls, total / 1e0, avg, pct))
print()
print("# per-kernel resources (first dispatch of each kernel)")
print("%-40s %6s %6s %6s %8s %10s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch"))
seen = set()
for name, v, a, s, l, sc in cur.execute(
        "select name,vgpr_count,accum_vgpr_count,sgpr_count,lds_size,scratch_size from kernels order by start"):
    if name in seen:
        continue
    seen.add(name)
    print("%-40s %6s %6s %6s %8s %10s" % (name.split("(")[0][:40], v, a, s, l, sc))
