#!/usr/bin/env python3
"""Timeline of the last gtsc_encode_host call in a rocprofv3 --kernel-trace --memory-copy-trace run (diagnostic).
usage: copy_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> [encode|decode]"""
import csv, glob, re, sys
d = sys.argv[1]; what = sys.argv[2] if len(sys.argv) > 2 else "encode"
mc = list(csv.DictReader(open(glob.glob(d + "/*/*_memory_copy_trace.csv")[0])))
kt = list(csv.DictReader(open(glob.glob(d + "/*/*_kernel_trace.csv")[0])))
ev = []
for r in mc:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"][-14:] + " s" + r["Stream_Id"]))
for r in kt:
    n = r["Kernel_Name"]
    m = re.search(r"(\w+_kernel)", n)
    if "gtsc" in n or "rocclr" in n:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + (m.group(1) if m else n[:40]) + " s" + r.get("Stream_Id", "")))
ev.sort()
idx = [i for i, e in enumerate(ev) if what + "_block" in e[2] or what + "_tiled" in e[2]]
last = idx[-1]
start = last
while start > 0 and ev[last][0] - ev[start - 1][0] < 60e6 and not ("decode" if what == "encode" else "encode") in ev[start - 1][2]:
    start -= 1
t0 = ev[start][0]
for e in ev[start:last + 16]:
    if e[1] - e[0] > 20000:
        print("%8.2f %8.2f %7.2f  %s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[2]))
