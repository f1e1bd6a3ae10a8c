#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by tools/profile.sh into one table:
per gtsc kernel, average duration and average counter values per launch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
rows = []
for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "gtsc" in r["Name"]:
            rows.append((r["Name"], int(r["Calls"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"])))
print("== kernel-trace stats (ns)")
for name, calls, avg, mn, mx in rows:
    print("%-60s calls=%d avg=%.0f min=%.0f max=%.0f" % (name[:60], calls, avg, mn, mx))
counters = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r.get("Kernel_Name", "")
        if "gtsc" in k:
            counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== PMC averages per launch")
for k, cs in counters.items():
    print(k[:70])
    for c in sorted(cs):
        v = cs[c]
        print("   %-28s %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
    if "FETCH_SIZE" in cs:
        # FETCH_SIZE is in KiB and counts wide coalesced reads at half their bytes on gfx950
        f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
        print("   fetched bytes (FETCH_SIZE KiB x1024 x2): %.4g" % (f * 1024 * 2))
    if "WRITE_SIZE" in cs:
        w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        print("   written bytes (WRITE_SIZE KiB x1024):    %.4g" % (w * 1024))

# HBM traffic per launch for bench.py's roofline.traffic (default workload only)
import json
traffic = {"workload": "gauge", "series": 1000000, "points": 120, "source": os.path.basename(root.rstrip("/"))}
for k, cs in counters.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024 * 2
        w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
        name = "encode" if "encode" in k else "decode" if "decode" in k else None
        if name and ("<0>" in k or "<0, " in k):  # the v2 kernels (the decoder also carries its ring size)
            traffic[name + "_bytes"] = round(f + w)
            traffic[name + "_fetch_bytes"] = round(f)
            traffic[name + "_write_bytes"] = round(w)
json.dump(traffic, open(os.path.join(root, "traffic.json"), "w"), indent=1)
