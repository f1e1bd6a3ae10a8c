#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 results database (kernel trace).  usage: prof_db_stats.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                  "from kernels group by name order by 3 desc")
for name, n, tot, avg, lo, hi in rows:
    print("%-84s n=%4d total %9.1f us  avg %8.1f  min %8.1f  max %8.1f" % (name[:84], n, tot, avg, lo, hi))
