#!/usr/bin/env python3
"""Kernels and memory copies of a rocprofv3 results database in time order (events of at least `min_us`).
usage: prof_db_timeline.py results.db [min_us] [last_n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 80
ev = []
for name, a, b, q, s in db.execute("select name,start,end,queue_id,stream_id from kernels"):
    ev.append((a, b, "K " + name.replace("void gtsc::(anonymous namespace)::", "").replace("gtsc::", "")[:44], q, s))
try:
    for name, a, b, size, s in db.execute("select name,start,end,size,stream_id from memory_copies"):
        ev.append((a, b, "C %s %d" % (name.replace("MEMORY_COPY_", ""), size), "-", s))
except sqlite3.OperationalError:
    pass
ev = sorted(e for e in ev if e[1] - e[0] >= min_us * 1e3)[-last_n:]
t0 = ev[0][0]
for a, b, what, q, s in ev:
    print("%-62s q%-3s s%-3s start %10.1f us  dur %9.1f" % (what, q, s, (a - t0) / 1e3, (b - a) / 1e3))
