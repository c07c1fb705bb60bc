#!/usr/bin/env python3
"""Summarise rocprofv3 result databases (the sqlite files ROCm 7.2 writes) as text:
kernel-trace statistics and per-kernel counter averages.
usage: rocprof_summary.py <dir with trace/ pmc*/ subdirs> > profiles/<name>.txt"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
for db in sorted(glob.glob(os.path.join(root, "*", "*.db"))):
    con = sqlite3.connect(db)
    tag = os.path.basename(os.path.dirname(db))
    print("== %s" % tag)
    if tag.startswith("trace"):
        print("kernel-trace --stats: name | calls | total_us | avg_us | pct")
        for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print("%-60s | %d | %.1f | %.1f | %.2f" % (name.split("(")[0][:60], calls, total, avg, pct))
    else:
        q = con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                        "where kernel_name like '%zkir%' group by kernel_name, counter_name")
        for k, c, n, v in q:
            print("%-40s %-24s dispatches=%d avg=%.6g" % (k.split("(")[0][:40], c, n, v))
