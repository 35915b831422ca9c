#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) PMC run: counters summed per kernel, plus per-leapfrog if --leap given."""
import sqlite3
import sys

db = sys.argv[1]
leap = float(sys.argv[2]) if len(sys.argv) > 2 else None
con = sqlite3.connect(db)
rows = con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                   "group by kernel_name, counter_name").fetchall()
for k, c, v, n in rows:
    if "run_kernel" in k:
        extra = "  per-leapfrog %.1f" % (v / leap) if leap else ""
        print("%-28s %.4e (%d dispatches)%s" % (c, v, n, extra))
